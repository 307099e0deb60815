#!/usr/bin/env python3
"""tools/config5_timing.py [copies] -- BASELINE config 5: examples/cohesive_test_2.inp tiled `copies` times (default
1000: 3,000 elements, 8,000 nodes, 16,000 dofs), the Newton increment loop of src/main.F90:216-296 with the reference's
own inner solver choice (KSPPREONLY + PCLU, served by the band LU) and with BiCGStab + Jacobi; wall time per run,
Newton and linear iteration counts.  The CPU oracle (dense LU) is timed on 40 copies for scale."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd import capi, driver  # noqa: E402
from fem_amd.inp import read_inp  # noqa: E402
from fem_amd.mesh import tile_deck, tile_deck_connected  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m1 = read_inp(os.path.join(ROOT, "tests", "golden", "decks", "cohesive_test_2.inp"))
m = tile_deck(m1, copies)
ctx = capi.Context(0)
print(f"config 5: {copies} copies, {m.nels} elements, {m.nnds} nodes, {m.ndof} dofs, {m.nsteps} increments")
for opts in ("-ksp_type preonly -pc_type lu", "-ksp_type bcgs -pc_type jacobi -ksp_rtol 1e-12 -ksp_atol 1e-16"):
    for rep in range(2):
        t0 = time.perf_counter()
        r = driver.run_nonlinear(m, opts, ctx=ctx, recover_stress=True)
        dt = time.perf_counter() - t0
    print(f"  [{opts}] {dt * 1e3:.1f} ms for {len(r.snes_its)} increments (second run); SNES iterations {list(r.snes_its)}, "
          f"reasons {list(r.snes_reasons)}, linear iterations {list(r.its)}, {m.ndof * len(r.snes_its) / dt:.3e} DOF-increments/s")
# the same deck as ONE connected strip (shared edge nodes: `copies` cohesive elements on one interface)
mc = tile_deck_connected(m1, copies)
print(f"config 5, connected strip: {mc.nels} elements, {mc.nnds} nodes, {mc.ndof} dofs")
for opts in ("-ksp_type preonly -pc_type lu",):
    for builtin in (False, True):
        for rep in range(3):
            t0 = time.perf_counter()
            r = driver.run_nonlinear(mc, opts, ctx=ctx, rels_mode=1, recover_stress=True, builtin_callbacks=builtin)
            dt = time.perf_counter() - t0
        print(f"  [{opts}, rels_mode 1, callbacks {'issued by the library (fem_snes_set_defmod_callbacks)' if builtin else 'in the host language (Python)'}] "
              f"{dt * 1e3:.1f} ms for {len(r.snes_its)} increments (third run); SNES iterations {list(r.snes_its)}, "
              f"reasons {list(r.snes_reasons)}, linear iterations {list(r.its)}")
# ... and the disjoint copies with the library's callbacks
for rep in range(3):
    t0 = time.perf_counter()
    r = driver.run_nonlinear(m, "-ksp_type preonly -pc_type lu", ctx=ctx, recover_stress=True, builtin_callbacks=True)
    dt = time.perf_counter() - t0
print(f"config 5, disjoint copies, callbacks issued by the library: {dt * 1e3:.1f} ms for {len(r.snes_its)} increments; SNES iterations {list(r.snes_its)}")
try:
    from oracle.oracle import Oracle
    o = Oracle()
    m40 = tile_deck(m1, 40)
    t0 = time.perf_counter()
    ro = o.snes_run(m40)
    print(f"  CPU oracle (dense LU stand-in for MUMPS), 40 copies = {m40.ndof} dofs: {(time.perf_counter() - t0) * 1e3:.1f} ms, "
          f"SNES iterations {list(ro['its'])}")
except Exception as e:   # noqa: BLE001
    print("  oracle leg skipped:", e)
ctx.destroy()
