#!/bin/bash
# Round-6 measurement call: headline bench (with config3_distorted, config 2 -- resident CG and launch path --, cold run, parity runs,
# multigrid, config 5, CPU legs), rocprofv3 --kernel-trace --stats of the SAME command, PMC FETCH / WRITE passes of the brick product,
# the assembly, the nodal stress and the assembled SpMV together with the kernel durations of the same programs on the same box,
# config 5, the multigrid set-up.  (GPU suite and SQ / LDS / TD counters: their own calls.)
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
echo "== bench"; timeout -k 10 700 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
echo "== rocprofv3 stats of the same command"; (cd /tmp && timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/$O/bench_profiled.json 2> $GRAFT_REPO_ROOT/$O/bench_profiled.err)
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof_stats
head -12 $O/kernel_stats.csv | cut -c1-160
pmc() { # tag program-args...
  local tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_${tag}_$c -o p -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py "$@" > $GRAFT_REPO_ROOT/$O/pmc_${tag}_$c.log 2>&1); done
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/dur_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py "$@" > /dev/null 2>&1)
  find $O/dur_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/dur_$tag.csv; rm -rf $O/dur_$tag
}
echo "== pmc ebe"; pmc ebe 149 10
PMC_KERNEL_STATS=$O/dur_ebe.csv python3 tools/pmc_to_traffic.py $O/pmc_ebe_FETCH_SIZE $O/pmc_ebe_WRITE_SIZE "k_ebe_tile_x4<2, true>" $O/ebe_traffic.json 149 3 "k_assemble_rows<3, 8, 32>" "k_tile_stress_cart<2, true, true>" > /dev/null
python3 tools/pmc_summary.py $O/pmc_ebe_FETCH_SIZE FETCH_SIZE > $O/pmc_summary.txt; python3 tools/pmc_summary.py $O/pmc_ebe_WRITE_SIZE WRITE_SIZE >> $O/pmc_summary.txt
echo "== pmc spmv"; pmc spmv 149 10 0 assembled
PMC_KERNEL_STATS=$O/dur_spmv.csv python3 tools/pmc_to_traffic.py $O/pmc_spmv_FETCH_SIZE $O/pmc_spmv_WRITE_SIZE "k_spmv_sell<3, true>" $O/spmv_traffic.json 149 3 > /dev/null
python3 tools/pmc_summary.py $O/pmc_spmv_FETCH_SIZE FETCH_SIZE >> $O/pmc_summary.txt; python3 tools/pmc_summary.py $O/pmc_spmv_WRITE_SIZE WRITE_SIZE >> $O/pmc_summary.txt
echo "== pmc general"; pmc gen 149 10 0.15
PMC_KERNEL_STATS=$O/dur_gen.csv python3 tools/pmc_to_traffic.py $O/pmc_gen_FETCH_SIZE $O/pmc_gen_WRITE_SIZE "k_ebe_tile<0, 0, true>" $O/general_traffic.json 149 3 > /dev/null
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
echo "== config 5"; timeout -k 10 300 python3 tools/config5_timing.py 1000 > $O/config5.txt 2>&1; cat $O/config5.txt
echo "== mg set-up"; timeout -k 10 300 python3 tools/mg_setup_steady.py 149 4 > $O/mg_setup.txt 2>&1; cat $O/mg_setup.txt
echo "== config 2"; timeout -k 10 300 python3 tools/r06_resident_check.py > $O/resident.txt 2>&1; tail -4 $O/resident.txt
echo "== first step"; FEM_AMD_EBE_VERBOSE=1 timeout -k 10 300 python3 tools/first_step_timing.py > $O/first_step.txt 2>&1
echo "== all done"
