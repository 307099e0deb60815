#!/usr/bin/env python3
"""tools/ebe_products.py [n] [reps] -- assembles config 3 once and runs `reps` element-by-element
products (for rocprofv3 --pmc passes over k_ebe_apply / k_ebe_gather)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = capi.Context(0)
m = structured_hex(n)
prob = driver.Problem(ctx, m, "-mat_mult_type ebe")
prob.assemble(); prob.ksp.set_up()
X, Y = capi.Vec(ctx, m.ndof), capi.Vec(ctx, m.ndof)
X.set_array(np.sin(0.001 * np.arange(m.ndof)))
print("avg product ms", prob.K.mult_timed(X, Y, reps), flush=True)
prob.destroy(); ctx.destroy()
