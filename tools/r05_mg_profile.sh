#!/bin/bash
# tools/r05_mg_profile.sh -- kernel times of the -pc_type mg step at config 3 (rocprofv3 --kernel-trace --stats), csv under gpurun_out/r05_mg_prof
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_mg_prof -o mg -- python3 $GRAFT_REPO_ROOT/tools/mg_timing.py 149 0 "mg -pc_mg_type additive -mat_mult_type ebe" "mg -pc_mg_type additive -mat_mult_type ebe" "mg -pc_mg_type additive -mat_mult_type ebe" > $GRAFT_REPO_ROOT/gpurun_out/r05_mg_prof.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r05_mg_prof.log | tail -5
