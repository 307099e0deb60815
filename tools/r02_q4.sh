#!/bin/bash
set -eo pipefail
O=gpurun_out/r02q4; mkdir -p $O
FEM_AMD_EBE_VERBOSE=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_ebe.py tests/test_gpu_fullsize.py tests/test_gpu_multigrid.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
FEM_AMD_EBE_VERBOSE=1 timeout -k 10 300 python3 bench.py --config 2 --steps 5 --warmup 2 --no-cpu --no-mg > $O/bench_c2.json 2> $O/bench_c2.err
grep "ebe tiles:" $O/bench_c2.err | head -2
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r02q4/bench_c2.json").read().strip().splitlines()[-1])
print("c2 tiles", d["value"], d["ms_per_step"], d["config"]["cg_iterations"], d["config"]["ms_per_cg_iteration"])
PY
FEM_AMD_EBE_TILE=0 timeout -k 10 300 python3 bench.py --config 2 --steps 5 --warmup 2 --no-cpu --no-mg > $O/bench_c2_gather.json 2> $O/bench_c2.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r02q4/bench_c2_gather.json").read().strip().splitlines()[-1])
print("c2 two-kernel", d["value"], d["ms_per_step"], d["config"]["cg_iterations"], d["config"]["ms_per_cg_iteration"])
PY
