#!/bin/bash
# tile shape sweep of the hex8 tile product (FEM_AMD_TILE_NB), n^3 cube
set -eo pipefail
O=gpurun_out/nb; mkdir -p $O
N=${1:-149}
for nb in 15,15,7 15,15,8 15,14,8 14,15,8 15,15,6 15,15,5 15,13,8 15,12,9 15,11,10 15,15,7; do
  TAG="nb=$nb" FEM_AMD_EBE_VERBOSE=1 FEM_AMD_TILE_NB=$nb timeout -k 10 120 python3 tools/ebe_apply_time.py $N 40 2>&1 | grep -i "tile\|tag=" | cut -c1-220 | tee -a $O/log_$N.txt
done
