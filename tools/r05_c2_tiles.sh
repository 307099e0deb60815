#!/bin/bash
# tools/r05_c2_tiles.sh -- config 2 (707^2 quad4): box sizes of the quad4 tile plan against the CG iteration time
cd "$(dirname "$0")/.."
out=gpurun_out/r05_c2_tiles.txt
: > $out
for nb in "" 15,15,1 23,23,1 31,15,1 15,31,1 31,23,1 47,15,1 63,15,1; do
  if [ -n "$nb" ]; then export FEM_AMD_TILE_NB=$nb; else unset FEM_AMD_TILE_NB; fi
  echo "== NB=${nb:-default}" >> $out
  python3 bench.py --config 2 --steps 3 --warmup 1 --no-cpu --no-mg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,3),'M DOF/s', d['config']['cg_iterations'],'its', round(d['config']['ms_per_cg_iteration']*1e3,2),'us/it', 'product', round(d['roofline']['avg_launch_ms']*1e3,2),'us', d['roofline'].get('tiles'))" >> $out 2>&1
done
cat $out
