#!/usr/bin/env python3
"""tools/soak_tile_products.py [first] [last] -- seeded random hex8 bricks (sizes, materials, straight / sheared / turned /
jittered / half-jittered / shuffled numbering) through the one-pass tile product: y against the assembled product per row,
reproducibility, and the nodal stress over the tiles against the element-buffer path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd import capi  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
ctx = capi.Context(0)
bad = []
for seed in range(lo, hi):
    rng = np.random.default_rng(1000 + seed)
    dims = tuple(int(v) for v in rng.integers(2, [48, 30, 22]))
    kind = ["straight", "sheared", "turned", "jitter", "half_jitter", "shuffled"][seed % 6]
    m = structured_hex(*dims, jitter=0.15 if kind == "jitter" else 0.0)
    if kind == "straight":   # bricks of three (dyadic) edge lengths: one geometry record per tile, the coefficients of hex_forces_cart_k
        m.coords = np.ascontiguousarray(m.coords * rng.choice([0.5, 0.75, 1.0, 1.25, 1.5, 2.0], 3))
    if kind == "sheared":
        M = np.eye(3) + 0.4 * rng.standard_normal((3, 3))
        if abs(np.linalg.det(M)) < 0.3:
            M = np.eye(3) + np.diag([0.2, -0.1, 0.3])
        m.coords = np.ascontiguousarray(m.coords @ M.T + rng.standard_normal(3))
    elif kind == "turned":
        Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        m.coords = np.ascontiguousarray((m.coords * np.array([1.0, 0.7, 1.3])) @ Q.T)
    elif kind == "half_jitter":
        j = structured_hex(*dims, jitter=0.2)
        keep = m.coords[:, 0] < 0.5 * dims[0]
        m.coords = np.where(keep[:, None], m.coords, j.coords)
    elif kind == "shuffled":
        pe = rng.permutation(m.nels)
        pn = rng.permutation(m.nnds).astype(np.int32)
        inv = np.empty_like(pn); inv[pn] = np.arange(m.nnds, dtype=np.int32)
        m.conn = np.ascontiguousarray(pn[m.conn.reshape(m.nels, 8)[pe]].reshape(-1))
        m.mat_id = np.ascontiguousarray(m.mat_id[pe])
        m.coords = np.ascontiguousarray(m.coords[inv])
        m.bcnode = m.bcnode.copy(); m.bcnode[:, 0] = pn[m.bcnode[:, 0]]
        if len(m.fnode):
            m.fnode = pn[m.fnode]
    dm = capi.DeviceMesh(ctx, m)
    K = capi.Mat(dm)
    K.assemble_elements(); K.apply_kbc()
    x = rng.standard_normal(m.ndof)
    X, Ya, Ye, U = (capi.Vec(ctx, m.ndof) for _ in range(4))
    X.set_array(x)
    K.mult(X, Ya)
    ya = Ya.get_array()
    U.set_array(1e-3 * x)
    s_buf = dm.recover_nodal_stress(U)
    K.set_mult_type("ebe")
    info = K.get_mult_info()
    K.mult(X, Ye); ye = Ye.get_array()
    K.mult(X, Ye)
    ok = info["kind"] == 2 and np.array_equal(ye, Ye.get_array())
    # per-row scale: |K| |x| is not at hand without the oracle; the penalty rows carry 1e30 x, the others the stiffness scale
    pen = np.abs(ya) > 1e20 * np.abs(x).max()
    scale = np.where(pen, np.abs(ya), np.abs(ya[~pen]).max())
    err = (np.abs(ye - ya) / scale).max()
    s_tile = dm.recover_nodal_stress(U)
    es = np.abs(s_tile - s_buf).max() / np.abs(s_buf).max()
    ok = ok and err <= 1e-12 and es <= 1e-11
    print(f"seed {seed} {kind} {dims}: tiles {info['tiles']} (stored geometry {info['affine_tiles']}) visits/elements "
          f"{info['element_visits'] / m.nels:.2f} err {err:.1e} stress {es:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    if not ok:
        bad.append(seed)
    for o in (X, Ya, Ye, U, K, dm):
        o.destroy()
print("tile products", lo, "..", hi - 1, "failures:", bad)
