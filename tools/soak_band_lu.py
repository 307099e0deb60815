#!/usr/bin/env python3
"""tools/soak_band_lu.py [first] [last] -- seeded soak of the unsymmetric direct solve on narrow bands: strips of quad4 / hex8 elements
(1-4 elements across, 60-2500 along), Mat_K with its penalties plus unsymmetric terms of random size on random off-diagonal entries
(MatSetValues), `-ksp_type preonly -pc_type lu` -- the band LU by one wavefront, block cyclic reduction where its two tests let it
through, the band LU behind it where they do not -- against scipy's sparse LU of the row-scaled system, judged by the componentwise
backward error (the strips are slender: cond ~ (L / h)^4)."""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex, structured_quad  # noqa: E402

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 100)
ctx = capi.Context(0)
bad, paths = [], {"bcr": 0, "band": 0}
if os.environ.get("SOAK_VERBOSE") != "1":
    os.environ.pop("FEM_AMD_DIRECT_VERBOSE", None)
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 5)
    nx = int(rng.integers(60, 2500))
    if kind < 4:
        m = structured_quad(nx, int(kind) + 1, jitter=float(rng.uniform(0.0, 0.2)))
    else:
        m = structured_hex(min(nx, 900), 1, 1, jitter=float(rng.uniform(0.0, 0.2)))
    prob = driver.Problem(ctx, m, "-ksp_type preonly -pc_type lu")
    prob.assemble()
    rp, ci, v = prob.K.get_csr()
    scale = np.abs(v[v < 1e29]).max()
    rows = rng.choice(np.arange(2, m.ndof - 2), size=max(1, m.ndof // int(rng.integers(5, 60))), replace=False)
    mag = float(10.0 ** rng.uniform(-3.0, -0.5))
    for r in rows:
        cols = ci[rp[r]:rp[r + 1]]
        cols = cols[cols != r]
        prob.K.set_values([int(r)], [int(cols[rng.integers(0, len(cols))])], [mag * scale * rng.uniform(-1, 1)], "add")
    rp, ci, v = prob.K.get_csr()
    prob.ksp.set_operators(prob.K); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    b = prob.F.get_array()
    prob.ksp.solve(prob.F, prob.U)
    x = prob.U.get_array()
    A = sp.csr_matrix((v, ci, rp))
    bwd = (np.abs(b - A @ x) / (abs(A) @ np.abs(x) + np.abs(b) + 1e-300)).max()
    ok = prob.ksp.get_converged_reason() == 4 and np.isfinite(x).all() and bwd <= 1e-9
    if not ok:
        bad.append((seed, m.ndof, float(bwd)))
    prob.destroy()
    if seed % 10 == 0:
        print(f"seed {seed}: {m.ndof} dofs, unsymmetric terms {mag:.1e} on {len(rows)} rows, backward error {bwd:.1e} {'ok' if ok else 'FAIL'}", flush=True)
print(f"band LU soak {lo} .. {hi - 1}: failures {bad}")
ctx.destroy()
