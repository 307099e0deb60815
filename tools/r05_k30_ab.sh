#!/bin/bash
# tools/r05_k30_ab.sh -- the tile tests with the current library, then tools/ab_libs.sh over the named builds
set -eo pipefail
mkdir -p gpurun_out/k30
timeout -k 10 900 python3 -m pytest tests/test_gpu_ebe.py tests/test_gpu_parity.py tests/test_gpu_multidomain.py -m gpu -x -q > gpurun_out/k30/pytest.txt 2>&1 || { tail -40 gpurun_out/k30/pytest.txt; exit 1; }
tail -2 gpurun_out/k30/pytest.txt
tools/ab_libs.sh "$@"
