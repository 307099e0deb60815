#!/bin/bash
# tools/cg_traffic.sh -- HBM bytes per kernel INSIDE the CG of config 3 (two rocprofv3 --pmc passes over one bench step): which of
# the iteration's ten vector passes and of the product's streams the caches serve.  FETCH_SIZE counts 1/2 of the bytes on gfx950
# (tools/pmc_calib.hip); both in KiB.
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/cgt
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_$c -o cg -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-mg --no-c2 $CGT_ARGS > $GRAFT_REPO_ROOT/$O/pmc_$c.log 2>&1)
  python3 tools/pmc_summary.py $O/pmc_$c $c > $O/summary_$c.txt
  rm -rf $O/pmc_$c
done
for c in FETCH_SIZE WRITE_SIZE; do sed -n 1,8p $O/summary_$c.txt | cut -c1-140; done
