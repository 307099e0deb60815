#!/usr/bin/env python3
"""tools/check_8_domains.py [n] -- 2 x 2 x 2 straight bricks of n^3 hex8 each as eight in-process domains on one GPU (small
tiles with x in LDS, shared nodes, halo sums) against the same global problem solved as one domain."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402
from tests.test_gpu_multidomain import _run_domains  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
opts = "-mat_mult_type ebe -ksp_rtol 1e-10"
grid, out = _run_domains(8, (n, n, n), 0.0, opts)
g = structured_hex(*(grid[d] * n for d in range(3)))
ctx = capi.Context(0)
one = driver.run(g, opts, ctx=ctx)
ug = one.aggregate_u.reshape(-1, 3)
sg = one.aggregate_stress
worst_u = worst_s = 0.0
for (u, s, its, reason, y, gid) in out:
    assert reason > 0
    worst_u = max(worst_u, np.abs(u.reshape(-1, 3) - ug[gid]).max() / np.abs(ug).max())
    worst_s = max(worst_s, np.abs(s - sg[gid]).max() / np.abs(sg).max())
print(f"8 domains of {n}^3: iterations {[o[2] for o in out][:2]}.. vs one domain {one.its[-1]}; max rel diff u {worst_u:.2e} stress {worst_s:.2e}")
assert worst_u <= 1e-7 and worst_s <= 1e-6
print("ok")
