#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
O=gpurun_out/r02s; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_ebe.py tests/test_gpu_multidomain.py tests/test_gpu_multigrid.py tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
FEM_AMD_EBE_VERBOSE=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mg > $O/bench_c3.json 2> $O/bench_c3.err
python3 - <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r02s/bench_c3.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["cg_iterations"], d["config"]["ms_per_cg_iteration"])
print({k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "matmult_ms", "gather_ms")})
PY
FEM_AMD_EBE_VERBOSE=1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu --no-mg --jitter 0.15 > $O/bench_c3_jit.json 2> $O/bench_c3.err
python3 - <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r02s/bench_c3_jit.json").read().strip().splitlines()[-1])
print("jitter", d["value"], d["ms_per_step"], d["config"]["cg_iterations"], d["config"]["ms_per_cg_iteration"])
print({k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "matmult_ms", "gather_ms")})
PY
grep "ebe tiles" $O/bench_c3.err | head -3
