#!/bin/bash
# tools/r06_gen_ab.sh name1 name2 ... -- A/B of library builds (fem_amd/lib/alt_<name>.so) on the general tile kernel: 30 products
# back to back on the distorted config 3 (tools/ebe_products.py 149 30 0.15), twice round the list, and the checksum of the result
set -eo pipefail
mkdir -p gpurun_out/ab
for round in 1 2; do
  for v in "$@"; do
    cp fem_amd/lib/alt_$v.so fem_amd/lib/libfem_amd.so
    echo "== $v (round $round)" | tee -a gpurun_out/ab/gen_summary.txt
    timeout -k 10 300 python3 tools/ebe_products.py 149 30 0.15 2>&1 | grep "avg product" | tee -a gpurun_out/ab/gen_summary.txt
  done
done
