#!/bin/bash
# Round-5 measurement call: headline bench (config3_distorted, config 2 with its roofline / CPU leg / multigrid, cold run, parity runs,
# multigrid, CPU leg), rocprofv3 --kernel-trace --stats of the SAME command, PMC FETCH / WRITE passes of the tile kernels, config 5
# (disjoint and connected strip), first-step timing.  (The GPU suite and the SQ / LDS / TD counters run in their own calls.)
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
echo "== bench"; timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
echo "== rocprofv3 stats of the same command"; (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/bench_profiled.json 2> $GRAFT_REPO_ROOT/$O/bench_profiled.err)
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof_stats
head -12 $O/kernel_stats.csv | cut -c1-160
echo "== pmc"; for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$c -o ebe -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py 149 10 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1); done
python3 tools/pmc_to_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "k_ebe_tile_x4<2, true>" $O/ebe_traffic.json 149 3 "k_assemble_rows<3, 8, 32>" "k_tile_stress_cart<2, true, true>" > /dev/null
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE FETCH_SIZE > $O/pmc_summary.txt; python3 tools/pmc_summary.py $O/pmc_WRITE_SIZE WRITE_SIZE >> $O/pmc_summary.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
echo "== config 5"; timeout -k 10 300 python3 tools/config5_timing.py 1000 > $O/config5.txt 2>&1; cat $O/config5.txt
echo "== first step"; FEM_AMD_EBE_VERBOSE=1 timeout -k 10 300 python3 tools/first_step_timing.py > $O/first_step.txt 2>&1
echo "== all done"
