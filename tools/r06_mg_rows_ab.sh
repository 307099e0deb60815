#!/bin/bash
# tools/r06_mg_rows_ab.sh -- the middle levels' residual of the multigrid cycle: k_box_residual_rows with nine or three slots' loads in
# flight (alt_rows9 = the default build, alt_rows3) against the 32-lanes-per-box kernel (FEM_AMD_MG_ROWS_MIN=100000), one box
set -eo pipefail
for round in 1 2; do
  for v in rows9 rows3; do
    cp fem_amd/lib/alt_$v.so fem_amd/lib/libfem_amd.so
    echo "== $v"; python3 tools/mg_setup_steady.py 149 3 | tail -1
  done
  cp fem_amd/lib/alt_rows9.so fem_amd/lib/libfem_amd.so
  echo "== slots kernel"; FEM_AMD_MG_ROWS_MIN=100000 python3 tools/mg_setup_steady.py 149 3 | tail -1
done
