#!/bin/bash
# SQ / LDS / TCC / TD counters of the element-by-element kernels at config 3 (separate rocprofv3 --pmc passes)
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_brick
mkdir -p $O
run() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$O/$name -o p -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py 149 5 > $GRAFT_REPO_ROOT/$O/$name.log 2>&1) || { echo "pass $name failed"; tail -3 $O/$name.log; return 0; }
  for c in "$@"; do python3 tools/pmc_summary.py $O/$name $c | grep -E "k_ebe_" >> $O/pmc_ebe_summary.txt || true; done
  rm -rf $O/$name
}
: > $O/pmc_ebe_summary.txt
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
run td TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run f FETCH_SIZE
run w WRITE_SIZE
cat $O/pmc_ebe_summary.txt | cut -c1-100
