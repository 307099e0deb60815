#!/bin/bash
# tools/r06_asm_ab.sh name1 name2 ... -- A/B of library builds (fem_amd/lib/alt_<name>.so) on the element loop of config 3
# (tools/assembly_timing.py: k_assemble_rows + ApplyKBC), regular and distorted mesh, twice round the list; then the parity tests
# of the assembly with the last library in place.
set -eo pipefail
mkdir -p gpurun_out/ab
for round in 1 2; do
  for v in "$@"; do
    cp fem_amd/lib/alt_$v.so fem_amd/lib/libfem_amd.so
    echo "== $v (round $round)" | tee -a gpurun_out/ab/asm_summary.txt
    timeout -k 10 300 python3 tools/assembly_timing.py 149 0 10 | tee -a gpurun_out/ab/asm_summary.txt
    timeout -k 10 300 python3 tools/assembly_timing.py 149 0.15 10 | tee -a gpurun_out/ab/asm_summary.txt
    timeout -k 10 300 python3 tools/assembly_timing.py 40 0.15 3 | tee -a gpurun_out/ab/asm_summary.txt
  done
done
