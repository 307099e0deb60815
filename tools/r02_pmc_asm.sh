#!/bin/bash
# SQ / TCC counters of the assembly kernel at config 3 (separate rocprofv3 --pmc passes, kernel-trace only)
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02p
mkdir -p $O
run() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$O/$name -o p -- python3 $GRAFT_REPO_ROOT/tools/asm_only.py 149 3 > $GRAFT_REPO_ROOT/$O/$name.log 2>&1)
  for c in "$@"; do python3 tools/pmc_summary.py $O/$name $c | grep -E "k_assemble_rows|k_stress" >> $O/pmc_asm_summary.txt || true; done
  rm -rf $O/$name
}
: > $O/pmc_asm_summary.txt
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE
run f FETCH_SIZE
run w WRITE_SIZE
cat $O/pmc_asm_summary.txt | cut -c1-110
