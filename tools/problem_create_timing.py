#!/usr/bin/env python3
"""tools/problem_create_timing.py -- what creating the objects of a run costs beside fem_mesh_create (config 3)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402
ctx = capi.Context(0)
m = structured_hex(149)
for rep in range(3):
    t0 = time.perf_counter(); p = driver.Problem(ctx, m, "-mat_mult_type ebe"); ctx.synchronize()
    print(f"Problem [{rep}]: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    p.destroy()
pr = cProfile.Profile(); pr.enable()
p = driver.Problem(ctx, m, "-mat_mult_type ebe"); ctx.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
