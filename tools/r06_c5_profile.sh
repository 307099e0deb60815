#!/bin/bash
# tools/r06_c5_profile.sh -- kernel trace of config 5's connected strip (callbacks issued by the library, as bench.py times it)
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/config5_strip.py 1000 6 1 | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5prof -o c5 -- python3 $GRAFT_REPO_ROOT/tools/config5_strip.py 1000 5 1 > /tmp/c5prof.log 2>&1
tail -3 /tmp/c5prof.log
f=$(find /tmp/c5prof -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r06_c5b_kernel_stats.csv
t=$(find /tmp/c5prof -name "*kernel_trace.csv" | head -1)
cp "$t" $GRAFT_REPO_ROOT/gpurun_out/r06_c5b_kernel_trace.csv
