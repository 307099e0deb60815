#!/bin/bash
# tools/r05_stress_ab.sh -- A/B of the nodal stress over the tiles: the tile tests, then tools/ebe_products.py with two library builds.
set -eo pipefail
mkdir -p gpurun_out/str
timeout -k 10 900 python3 -m pytest tests/test_gpu_ebe.py tests/test_gpu_parity.py tests/test_gpu_multidomain.py -m gpu -x -q > gpurun_out/str/pytest.txt 2>&1 || { tail -30 gpurun_out/str/pytest.txt; exit 1; }
tail -2 gpurun_out/str/pytest.txt
for round in 1 2; do
  for v in ${AB_LIBS:-r05all4 r05str}; do
    cp fem_amd/lib/alt_$v.so fem_amd/lib/libfem_amd.so
    timeout -k 10 300 python3 tools/ebe_products.py 149 30 > gpurun_out/str/$v.$round.log 2>&1
    echo $v $(grep "avg" gpurun_out/str/$v.$round.log | tr '\n' ' ')
  done
done
cp fem_amd/lib/alt_${AB_KEEP:-r05str}.so fem_amd/lib/libfem_amd.so
