#!/bin/bash
# tools/r05_general_nb_sweep.sh -- tile shapes (owned nodes per box, FEM_AMD_TILE_NB) for the general tile kernel on the distorted config 3:
# products back to back (tools/ebe_products.py 149 30 0.15)
cd "$(dirname "$0")/.."
out=gpurun_out/r05_general_nb_sweep.txt
: > $out
for nb in "" 15,15,7 15,15,6 15,15,5 15,15,8 15,13,7 15,11,7 15,9,7 15,7,7 15,15,4 15,11,6 15,10,10 15,12,8; do
  echo "== NB=${nb:-default}" >> $out
  if [ -n "$nb" ]; then export FEM_AMD_TILE_NB=$nb; else unset FEM_AMD_TILE_NB; fi
  FEM_AMD_EBE_VERBOSE=1 timeout -k 10 120 python3 tools/ebe_products.py 149 30 0.15 2>&1 | grep -E "ebe tiles:|avg" | cut -c1-200 >> $out
done
cat $out
