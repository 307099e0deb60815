#!/bin/bash
# fp64 issue-rate microbenchmark + SQ counters of k_ebe_apply (monomial-basis hex8 vs the Gauss-loop form)
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
true
true
run() { # name mono counters...
  local name=$1; local mono=$2; shift; shift
  export FEM_AMD_EBE_MONO=$mono
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/$O/$name -o p -- python3 $GRAFT_REPO_ROOT/tools/ebe_products.py 149 5 > $GRAFT_REPO_ROOT/$O/$name.log 2>&1) || { echo "pass $name mono=$mono failed"; tail -3 $O/$name.log; return 0; }
  for c in "$@"; do python3 tools/pmc_summary.py $O/$name $c | grep -E "k_ebe_apply" | sed "s/^/mono=$mono /" >> $O/pmc_ebe_summary.txt || true; done
  rm -rf $O/$name
}
: > $O/pmc_ebe_summary.txt
for mono in 1; do
run sq1 $mono SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 $mono SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
run sq3 $mono SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAVE_DEP_WAIT SQ_IFETCH SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU
done
cat $O/pmc_ebe_summary.txt | cut -c1-120
