#!/bin/bash
# tools/r06_mg_profile.sh -- kernel statistics of steady-state -pc_type mg steps of config 3 (tools/mg_setup_steady.py 149 4)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mgprof -o mg -- python3 $GRAFT_REPO_ROOT/tools/mg_setup_steady.py 149 4 > /tmp/mgprof.log 2>&1
tail -2 /tmp/mgprof.log
cp "$(find /tmp/mgprof -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r06_mg_kernel_stats.csv
