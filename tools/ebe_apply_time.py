#!/usr/bin/env python3
"""tools/ebe_apply_time.py [n] [reps] [sheared|shuffled] -- per-kernel time of the element-by-element product (event pairs around
the element kernel and the gather) on the n^3 hex8 cube; `sheared`: the cube mapped to a parallelepiped (stored adj(J))."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = capi.Context(0)
m = structured_hex(n)
if len(sys.argv) > 3 and sys.argv[3] == "shuffled":     # random element order and node numbering
    rng = np.random.default_rng(3)
    pe = rng.permutation(m.nels)
    pn = rng.permutation(m.nnds).astype(np.int32)
    inv = np.empty_like(pn); inv[pn] = np.arange(m.nnds, dtype=np.int32)
    m.conn = np.ascontiguousarray(pn[m.conn.reshape(m.nels, 8)[pe]].reshape(-1))
    m.mat_id = np.ascontiguousarray(m.mat_id[pe])
    m.coords = np.ascontiguousarray(m.coords[inv])
    m.bcnode = m.bcnode.copy(); m.bcnode[:, 0] = pn[m.bcnode[:, 0]]
    if len(m.fnode):
        m.fnode = pn[m.fnode]
if len(sys.argv) > 3 and sys.argv[3] == "sheared":
    M = np.array([[1.1, 0.3, -0.2], [0.05, 0.9, 0.4], [-0.15, 0.2, 1.3]])
    m.coords = np.ascontiguousarray(m.coords @ M.T + np.array([0.3, -1.7, 2.9]))
prob = driver.Problem(ctx, m, "-mat_mult_type ebe")
prob.assemble(); prob.ksp.set_up()
X, Y = capi.Vec(ctx, m.ndof), capi.Vec(ctx, m.ndof)
X.set_array(np.sin(0.001 * np.arange(m.ndof)))
for _ in range(5):
    prob.K.mult(X, Y)
prob.K.set_profiling(1)
for _ in range(reps):
    prob.K.mult(X, Y)
ctx.synchronize()
k, el, ga, pe = prob.K.get_profile(reps)
print(f"tag={os.environ.get('TAG', '')} products={k} apply_ms={el / k:.4f} gather_ms={ga / k:.4f}", flush=True)
prob.destroy(); ctx.destroy()
