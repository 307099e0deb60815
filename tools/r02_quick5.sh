#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02q
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_snes.py tests/test_gpu_cohesive.py -x -q > $O/pytest_snes.txt 2>&1 || { tail -60 $O/pytest_snes.txt; exit 1; }
tail -3 $O/pytest_snes.txt
