#!/usr/bin/env python3
"""tools/mesh_create_timing.py [n] -- what handing a mesh over costs (the boundary's host arrays -> device SoA): the Python side
(array conversions) and fem_mesh_create itself, config 3 by default."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
ctx = capi.Context(0)
m = structured_hex(n)
for rep in range(3):
    t0 = time.perf_counter()
    dm = capi.DeviceMesh(ctx, m)
    ctx.synchronize()
    print(f"DeviceMesh [{rep}]: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    dm.destroy()
pr = cProfile.Profile()
pr.enable()
dm = capi.DeviceMesh(ctx, m)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
