#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02q
mkdir -p $O
for v in 0 1 2 4 6 7; do
  export FEM_AMD_ASM_DEBUG=$v
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$v -o c3 -- python3 $GRAFT_REPO_ROOT/tools/asm_only.py 149 5 > $GRAFT_REPO_ROOT/$O/asm_only_v$v.txt 2> $GRAFT_REPO_ROOT/$O/asm_only_v$v.err)
  f=$(find $O/prof$v -name "*kernel_stats.csv" | head -1); echo "dbg=$v $(grep -E "k_assemble_rows" $f | awk -F'",' '{print $2}')"
  rm -rf $O/prof$v
done
