#!/bin/bash
# tools/r05_mg_ab.sh lib1 lib2 ... -- A/B of the -pc_type mg step at config 3 (tools/mg_timing.py, three solves each, twice round the list)
set -eo pipefail
mkdir -p gpurun_out/mgab
for round in 1 2; do
  for v in "$@"; do
    cp fem_amd/lib/alt_$v.so fem_amd/lib/libfem_amd.so
    timeout -k 10 300 python3 tools/mg_timing.py 149 0 "mg -pc_mg_type additive -mat_mult_type ebe" "mg -pc_mg_type additive -mat_mult_type ebe" "mg -pc_mg_type additive -mat_mult_type ebe" > gpurun_out/mgab/$v.$round.log 2>&1
    echo $v $(tail -1 gpurun_out/mgab/$v.$round.log)
  done
done
