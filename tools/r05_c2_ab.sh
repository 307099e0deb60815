#!/bin/bash
# tools/r05_c2_ab.sh lib... -- A/B of library builds on config 3 and config 2 (bench.py without the CPU legs, multigrid, distorted mesh, config 5)
set -eo pipefail
mkdir -p gpurun_out/c2ab
for round in 1 2; do
  for v in "$@"; do
    cp fem_amd/lib/alt_$v.so fem_amd/lib/libfem_amd.so
    timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-mg --no-distorted --no-c5 > gpurun_out/c2ab/$v.json 2> gpurun_out/c2ab/$v.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/c2ab/$v.json').read().strip().splitlines()[-1])
c2=d['config2']
print('$v c3', round(d['value']/1e6,2), 'M DOF/s', round(d['config']['ms_per_cg_iteration'],4), 'ms/it | c2', round(c2['value']/1e6,3), 'M DOF/s', round(c2['ms_per_cg_iteration']*1e3,2), 'us/it', c2['cg_iterations'], 'its', flush=True)"
  done
done
