#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
run() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$O/$name -o p -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py 149 5 > $GRAFT_REPO_ROOT/$O/$name.log 2>&1) || { echo "pass $name failed"; tail -3 $O/$name.log; return 0; }
  for c in "$@"; do python3 tools/pmc_summary.py $O/$name $c | grep -E "k_ebe_" >> $O/pmc2.txt || true; done
  rm -rf $O/$name
}
: > $O/pmc2.txt
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
run ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run td TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
cat $O/pmc2.txt | cut -c1-90
