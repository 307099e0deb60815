#!/bin/bash
# tools/r06_ebe_pmc.sh -- SQ / LDS / TCC / TCP / TD counters and HBM bytes of the tile kernels and the assembly (round 6) at config 3: the brick kernel
# k_ebe_tile_x<2, true, true> (regular mesh) and the general kernel k_ebe_tile<0, 0, true> (jitter 0.15); separate rocprofv3 --pmc passes
# with the program directly after `--`
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_r06
mkdir -p $O
run() { # tag jitter name counters...
  local tag=$1 jit=$2 name=$3; shift 3
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$O/$tag$name -o p -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py 149 5 $jit > $GRAFT_REPO_ROOT/$O/$tag$name.log 2>&1) || { echo "pass $tag$name failed"; tail -3 $O/$tag$name.log; return 0; }
  for c in "$@"; do python3 tools/pmc_summary.py $O/$tag$name $c | grep -E "k_ebe_|k_tile_stress|k_assemble_rows" >> $O/pmc_${tag}summary.txt || true; done
  rm -rf $O/$tag$name
}
for spec in "brick_ 0" "general_ 0.15"; do
  set -- $spec; tag=$1; jit=$2
  : > $O/pmc_${tag}summary.txt
  run $tag $jit sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  run $tag $jit sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
  run $tag $jit lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU
  run $tag $jit tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  run $tag $jit tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
  run $tag $jit td TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
  run $tag $jit f FETCH_SIZE
  run $tag $jit w WRITE_SIZE
  echo "== $tag"; cut -c1-110 $O/pmc_${tag}summary.txt
done
