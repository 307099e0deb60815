#!/bin/bash
# tools/mg_ab.sh name1 name2 ... -- A/B of library builds (fem_amd/lib/alt_<name>.so) on the multigrid-preconditioned solve of
# config 3 (tools/mg_timing.py: -pc_type mg -pc_mg_type additive -mat_mult_type ebe), twice round the list
set -eo pipefail
export FEM_AMD_ALLOW_OLDER_LIB=1
mkdir -p gpurun_out/ab
for round in 1 2; do
  for v in "$@"; do
    cp fem_amd/lib/alt_$v.so fem_amd/lib/libfem_amd.so
    echo "== $v" | tee -a gpurun_out/ab/mg_summary.txt
    timeout -k 10 300 python3 tools/mg_timing.py 149 0 "mg -pc_mg_type additive -mat_mult_type ebe" "mg -pc_mg_type additive -mat_mult_type ebe" "mg -pc_mg_type additive -mat_mult_type ebe" 2>&1 | tee -a gpurun_out/ab/mg_summary.txt
  done
done
