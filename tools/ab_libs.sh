#!/bin/bash
# tools/ab_libs.sh name1 name2 ... -- A/B of library builds on one box: fem_amd/lib/alt_<name>.so takes the place of
# libfem_amd.so, the headline bench runs (no CPU leg, no multigrid, no config 2; further flags in $AB_ARGS), twice round the list.
set -eo pipefail
mkdir -p gpurun_out/ab
for round in 1 2; do
  for v in "$@"; do
    cp fem_amd/lib/alt_$v.so fem_amd/lib/libfem_amd.so
    timeout -k 10 300 python3 bench.py --steps ${AB_STEPS:-5} --warmup 2 --no-cpu --no-mg --no-c2 $AB_ARGS > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab/$v.json').read().strip().splitlines()[-1])
print('$v', round(d['value']/1e6,2), 'M DOF/s', round(d['ms_per_step'],2), 'ms/step', round(d['config']['ms_per_cg_iteration'],4), 'ms/it, product', round(d['roofline']['avg_launch_ms'],4), flush=True)" | tee -a gpurun_out/ab/summary.txt
  done
done
