#!/bin/bash
# Round-3 measurement call: PMC passes of the tile kernels (product, nodal stress, assembly), headline bench (with config 2,
# cold run, parity run, multigrid, CPU leg), the distorted mesh, config 2 alone, rocprofv3 stats of the bench, first-step
# timing, config 5.  (The GPU suite runs in its own call.)
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
echo "== pmc"; for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$c -o ebe -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py 149 10 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1); done
python3 tools/pmc_to_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "k_ebe_tile_x<2, true, true>" $O/ebe_traffic.json 149 3 "k_assemble_rows<3, 8, 32>" "k_tile_stress_cart<2>" > /dev/null
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE FETCH_SIZE > $O/pmc_summary.txt; python3 tools/pmc_summary.py $O/pmc_WRITE_SIZE WRITE_SIZE >> $O/pmc_summary.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cp $O/ebe_traffic.json profiles/ebe_traffic.json
echo "== bench"; timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
echo "== bench, distorted mesh"; timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mg --no-c2 --jitter 0.15 > $O/bench_c3_jitter.json 2> $O/bench_c3_jitter.err
echo "== bench config 2"; timeout -k 10 300 python3 bench.py --config 2 --steps 3 --warmup 1 --no-mg > $O/bench_c2.json 2> $O/bench_c2.err
echo "== rocprofv3 stats (bench with the mg leg)"; (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/$O/bench_profiled.json 2> $GRAFT_REPO_ROOT/$O/bench_profiled.err)
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof_stats
head -12 $O/kernel_stats.csv | cut -c1-160
echo "== first step"; FEM_AMD_EBE_VERBOSE=1 timeout -k 10 300 python3 tools/first_step_timing.py > $O/first_step.txt 2>&1
echo "== config 5"; timeout -k 10 300 python3 tools/config5_timing.py 1000 > $O/config5.txt 2>&1; cat $O/config5.txt
echo "== all done"
