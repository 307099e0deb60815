#!/bin/bash
# Round-2 GPU call A: test suite, headline bench, rocprofv3 stats of the same command, fresh PMC passes for the
# element-by-element product and the assembly kernels.  Run from the repo root on the GPU box.
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null || true; python3 -c "import os;print('affinity',len(os.sched_getaffinity(0)))"; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA"; free -g | head -2; } > $O/box_info.txt 2>&1
echo "== tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
echo "== bench"; timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
cat $O/bench.json
echo "== bench config 2"; timeout -k 10 300 python3 bench.py --config 2 --steps 3 --warmup 1 --no-mg > $O/bench_c2.json 2> $O/bench_c2.err
cat $O/bench_c2.json
echo "== rocprofv3 stats"; (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-mg > $GRAFT_REPO_ROOT/$O/bench_profiled.json 2> $GRAFT_REPO_ROOT/$O/bench_profiled.err)
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
echo "== pmc"; for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$c -o ebe -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py 149 10 > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1); done
python3 tools/pmc_to_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "k_ebe_apply<3, 8, 8>" $O/ebe_traffic.json 149 3 "k_ebe_gather<3>" "k_assemble_rows<3, 8, 32>" > /dev/null
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE FETCH_SIZE > $O/pmc_summary.txt; python3 tools/pmc_summary.py $O/pmc_WRITE_SIZE WRITE_SIZE >> $O/pmc_summary.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof_stats
echo "== done"
