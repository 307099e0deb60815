#!/bin/bash
# Second measurement call of round 6 (after the Newton-loop work): the headline bench and rocprofv3 --kernel-trace --stats of the SAME
# command, config 5 standalone.  (PMC passes of r06_final.sh: kernels unchanged since.)
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r06b
mkdir -p $O
echo "== bench"; timeout -k 10 700 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
echo "== driver-like"; timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverlike.json 2> $O/bench_driverlike.err; cut -c1-200 $O/bench_driverlike.json
echo "== rocprofv3 stats of the same command"; (cd /tmp && timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/bench_profiled.json 2> $GRAFT_REPO_ROOT/$O/bench_profiled.err)
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof_stats
head -8 $O/kernel_stats.csv | cut -c1-160
echo "== config 5"; timeout -k 10 300 python3 tools/config5_timing.py 1000 > $O/config5.txt 2>&1; cat $O/config5.txt
echo "== all done"
