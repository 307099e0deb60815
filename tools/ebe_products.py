#!/usr/bin/env python3
"""tools/ebe_products.py [n] [reps] [jitter] [mult] -- assembles config 3 once (jitter > 0: its distorted variant, the general tile
kernel) and runs `reps` element-by-element products and three nodal stress recoveries (for rocprofv3 --pmc passes over the tile
kernels)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
jitter = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
mult = sys.argv[4] if len(sys.argv) > 4 else "ebe"          # ebe | assembled (the sliced-ELL SpMV)
ctx = capi.Context(0)
m = structured_hex(n, jitter=jitter)
prob = driver.Problem(ctx, m, f"-mat_mult_type {mult}")
prob.assemble(); prob.ksp.set_up()
X, Y = capi.Vec(ctx, m.ndof), capi.Vec(ctx, m.ndof)
X.set_array(np.sin(0.001 * np.arange(m.ndof)))
print("avg product ms", prob.K.mult_timed(X, Y, reps), flush=True)
S = capi.Vec(ctx, m.nnds * 6)
import time  # noqa: E402
for _ in range(3):                      # nodal stress over the tiles (k_tile_stress_cart on this mesh of bricks)
    prob.dmesh.recover_nodal_stress_vec(X, S)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    prob.dmesh.recover_nodal_stress_vec(X, S)
ctx.synchronize()
print("avg nodal stress ms", (time.perf_counter() - t0) * 1e3 / 20, flush=True)
S.destroy()
prob.destroy(); ctx.destroy()
