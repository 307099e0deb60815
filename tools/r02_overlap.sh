#!/bin/bash
set -eo pipefail
O=gpurun_out/r02o; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_multidomain.py tests/test_gpu_ebe.py tests/test_gpu_multigrid.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
FEM_AMD_OVERLAP=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_multidomain.py -x -q > $O/pytest_overlap.txt 2>&1 || { tail -40 $O/pytest_overlap.txt; exit 1; }
tail -1 $O/pytest_overlap.txt
FEM_AMD_OVERLAP=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-mg --force-comm > $O/bench_forcecomm.json 2> $O/bench_forcecomm.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r02o/bench_forcecomm.json").read().strip().splitlines()[-1])
print("force-comm", d["value"], d["ms_per_step"], d["config"]["cg_iterations"], d["config"]["ms_per_cg_iteration"])
PY
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu --no-mg --force-comm > $O/bench_forcecomm_serial.json 2> $O/bench_forcecomm.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r02o/bench_forcecomm_serial.json").read().strip().splitlines()[-1])
print("force-comm, one stream", d["value"], d["ms_per_step"], d["config"]["cg_iterations"], d["config"]["ms_per_cg_iteration"])
PY
