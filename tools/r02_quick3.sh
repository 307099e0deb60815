#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02q
mkdir -p $O
for v in 0 1 2; do
  export FEM_AMD_ASM_VARIANT=$v
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "assembl" > $O/pytest_v$v.txt 2>&1 || { tail -40 $O/pytest_v$v.txt; exit 1; }
  tail -1 $O/pytest_v$v.txt
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$v -o c3 -- python3 $GRAFT_REPO_ROOT/tools/asm_only.py 149 5 > $GRAFT_REPO_ROOT/$O/asm_only_v$v.txt 2> $GRAFT_REPO_ROOT/$O/asm_only_v$v.err)
  cat $O/asm_only_v$v.txt
  f=$(find $O/prof$v -name "*kernel_stats.csv" | head -1); grep -E "k_assemble_rows" $f | awk -F'",' '{print substr($1,1,50), $2}'
  rm -rf $O/prof$v
done
