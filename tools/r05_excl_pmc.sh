#!/bin/bash
# tools/r05_excl_pmc.sh -- (needs profiles/r05_experiment_exclusive_tiles.patch applied) SQ / LDS counters of the general tile kernel on the distorted config 3: exclusive tiles against node-complete tiles
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_r05
mkdir -p $O
run() { # tag excl name counters...
  local tag=$1 ex=$2 name=$3; shift 3
  export FEM_AMD_TILE_EXCLUSIVE=$ex
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$O/$tag$name -o p -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py 149 5 0.15 > $GRAFT_REPO_ROOT/$O/$tag$name.log 2>&1) || { echo "pass $tag$name failed"; tail -3 $O/$tag$name.log; return 0; }
  for c in "$@"; do python3 tools/pmc_summary.py $O/$tag$name $c | grep -E "k_ebe_tile<0, 0|k_rim_sum<0" >> $O/pmc_${tag}summary.txt || true; done
  rm -rf $O/$tag$name
}
for spec in "excl_ 1" "classic_ 0"; do
  set -- $spec; tag=$1; ex=$2
  : > $O/pmc_${tag}summary.txt
  run $tag $ex sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  run $tag $ex sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
  run $tag $ex lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU
  run $tag $ex td TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
  echo "== $tag"; cut -c1-120 $O/pmc_${tag}summary.txt
done
