#!/bin/bash
# tools/r06_bcr_ab.sh -- config 5's connected strip with the block cyclic reduction as one resident launch per solve (default) and as a
# launch per level (FEM_AMD_LU_BCR_RESIDENT=0), alternating on one box
for i in 1 2; do
  for r in 1 0; do
    echo "== FEM_AMD_LU_BCR_RESIDENT=$r"
    FEM_AMD_LU_BCR_RESIDENT=$r python tools/config5_strip.py 1000 6 | tail -3
  done
done
