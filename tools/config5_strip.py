#!/usr/bin/env python3
"""tools/config5_strip.py [copies] [reps] [builtin] -- BASELINE config 5 as one connected strip (examples/cohesive_test_2.inp tiled along x with
shared edge nodes): the Newton increment loop with the reference's inner solver (KSPPREONLY + PCLU), rels_mode 1;
wall time of the last of `reps` runs; `builtin` = 1: the two SNES callbacks issued by the library (what bench.py times), 0 (default): by
this interpreter.  Meant to sit behind `rocprofv3 --kernel-trace --stats` as well."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd import capi, driver  # noqa: E402
from fem_amd.inp import read_inp  # noqa: E402
from fem_amd.mesh import tile_deck_connected  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
builtin = len(sys.argv) > 3 and sys.argv[3] == "1"
mc = tile_deck_connected(read_inp(os.path.join(ROOT, "tests", "golden", "decks", "cohesive_test_2.inp")), copies)
ctx = capi.Context(0)
print(f"config 5, connected strip: {mc.nels} elements, {mc.nnds} nodes, {mc.ndof} dofs", flush=True)
for rep in range(reps):
    t0 = time.perf_counter()
    r = driver.run_nonlinear(mc, "-ksp_type preonly -pc_type lu", ctx=ctx, rels_mode=1, recover_stress=True, builtin_callbacks=builtin)
    dt = time.perf_counter() - t0
    print(f"  run {rep}: {dt * 1e3:.1f} ms for {len(r.snes_its)} increments; SNES iterations {list(r.snes_its)}, reasons {list(r.snes_reasons)}, "
          f"linear solves {list(r.its)}", flush=True)
ctx.destroy()
