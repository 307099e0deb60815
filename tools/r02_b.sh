#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fortran.py tests/test_gpu_snes.py tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1 || { tail -60 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
timeout -k 10 300 ./tools/spmv_bench 150 20 > $O/spmv_format_microbench.txt 2>&1
cat $O/spmv_format_microbench.txt
