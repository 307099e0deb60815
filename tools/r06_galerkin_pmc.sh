#!/bin/bash
# tools/r06_galerkin_pmc.sh lib... -- HBM bytes and SQ / TCC / TCP / TA counters of the set-up kernels of -pc_type mg at config 3
# (k_galerkin_fine*, k_assemble_rows), one rocprofv3 --pmc pass per group, the program directly after `--`
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/pmc_gal
mkdir -p $O
run() { # lib name counters...
  local lib=$1 name=$2; shift 2
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$O/$lib$name -o p -- python3 $GRAFT_REPO_ROOT/tools/mg_timing.py 149 0 "mg -pc_mg_type additive -mat_mult_type ebe" > $GRAFT_REPO_ROOT/$O/$lib$name.log 2>&1) || { echo "pass $lib$name failed"; tail -3 $O/$lib$name.log; return 0; }
  for c in "$@"; do python3 tools/pmc_summary.py $O/$lib$name $c | grep -E "k_galerkin_fine|k_assemble_rows" >> $O/pmc_${lib}_summary.txt || true; done
  rm -rf $O/$lib$name
}
for lib in "$@"; do
  cp fem_amd/lib/alt_$lib.so fem_amd/lib/libfem_amd.so
  : > $O/pmc_${lib}_summary.txt
  run $lib f FETCH_SIZE
  run $lib w WRITE_SIZE
  run $lib tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  run $lib td TD_TD_BUSY_sum TD_TC_STALL_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
  run $lib sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  run $lib sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
  echo "== $lib"; cut -c1-170 $O/pmc_${lib}_summary.txt
done
