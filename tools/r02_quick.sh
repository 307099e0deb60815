#!/bin/bash
# quick GPU check of the assembly / stress kernels: parity tests that touch them + per-kernel times at config 3
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02q
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_ebe.py tests/test_gpu_multidomain.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c3 -- python3 $GRAFT_REPO_ROOT/tools/profile_path.py c3 > $GRAFT_REPO_ROOT/$O/path_c3.txt 2> $GRAFT_REPO_ROOT/$O/path_c3.err)
cat $O/path_c3.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_c3.csv; head -25 $O/kernel_stats_c3.csv | cut -c1-150
rm -rf $O/prof
