#!/bin/bash
# tools/r06_c5d_profile.sh -- kernel trace of config 5 as 1,000 disjoint copies (callbacks issued by the library)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/c5d.py <<'PY'
import os, sys, time
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
from fem_amd import capi, driver
from fem_amd.inp import read_inp
from fem_amd.mesh import tile_deck
m = tile_deck(read_inp(os.path.join(ROOT, "tests", "golden", "decks", "cohesive_test_2.inp")), 1000)
ctx = capi.Context(0)
for rep in range(5):
    t0 = time.perf_counter()
    r = driver.run_nonlinear(m, "-ksp_type preonly -pc_type lu", ctx=ctx, rels_mode=0, recover_stress=True, builtin_callbacks=True)
    print(f"run {rep}: {(time.perf_counter() - t0) * 1e3:.2f} ms", list(r.snes_its), flush=True)
PY
python3 /tmp/c5d.py | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5dprof -o c5d -- python3 /tmp/c5d.py > /tmp/c5dprof.log 2>&1
cp "$(find /tmp/c5dprof -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r06_c5d_kernel_stats.csv
cp "$(find /tmp/c5dprof -name '*kernel_trace.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/r06_c5d_kernel_trace.csv
