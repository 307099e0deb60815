#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02q
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c3 -- python3 $GRAFT_REPO_ROOT/tools/asm_only.py 149 5 > $GRAFT_REPO_ROOT/$O/asm_only.txt 2> $GRAFT_REPO_ROOT/$O/asm_only.err)
cat $O/asm_only.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_asm.csv; grep -E "k_assemble_rows|k_stress|k_sort|k_slots|k_node" $O/kernel_stats_asm.csv | awk -F'",' '{print substr($1,1,50), $2}'
rm -rf $O/prof
