#!/bin/bash
# tools/r05_excl_sweep.sh -- (needs profiles/r05_experiment_exclusive_tiles.patch applied) exclusive tiles on the distorted config 3: box sizes against the node-complete tiles (product back to back)
cd "$(dirname "$0")/.."
out=gpurun_out/r05_excl_sweep.txt
: > $out
for nb in "" 16,16,8 16,16,4 16,8,8 16,16,6 16,16,7 16,12,8 32,8,8 16,16,5 15,16,8 15,15,8 14,16,8 15,15,7; do
  for ex in 1 0; do
    [ "$ex" = 0 ] && [ -n "$nb" ] && [ "$nb" != "15,15,7" ] && continue
    echo "== NB=${nb:-default} EXCLUSIVE=$ex" >> $out
    if [ -n "$nb" ]; then export FEM_AMD_TILE_NB=$nb; else unset FEM_AMD_TILE_NB; fi
    FEM_AMD_TILE_EXCLUSIVE=$ex FEM_AMD_EBE_VERBOSE=1 python3 tools/ebe_products.py 149 30 0.15 2>&1 | grep -E "ebe tiles:|avg" >> $out
  done
done
cat $out
