#!/usr/bin/env python3
"""tools/mg_timing.py -- assemble+solve of config 3 with CG/Jacobi and with -pc_type mg."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
jit = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
m = structured_hex(n, jitter=jit)
ctx = capi.Context(0)
pcs = sys.argv[3:] if len(sys.argv) > 3 else ["jacobi", "mg", "mg", "mg -pc_mg_type additive", "mg -pc_mg_type additive"]
for pc in pcs:
    prob = driver.Problem(ctx, m, f"-pc_type {pc}")
    t0 = time.perf_counter()
    prob.assemble(); ctx.synchronize()
    t1 = time.perf_counter()
    prob.ksp.set_up(); ctx.synchronize()
    t2 = time.perf_counter()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U); ctx.synchronize()
    t3 = time.perf_counter()
    print(f"n={n} jitter={jit} pc={pc}: assemble {1e3*(t1-t0):.1f} ms, pc setup {1e3*(t2-t1):.1f} ms, "
          f"solve {1e3*(t3-t2):.1f} ms, its {prob.ksp.get_iteration_number()}, reason {prob.ksp.get_converged_reason()}, "
          f"DOF/s {m.ndof/(t3-t0):.3e}", flush=True)
    prob.destroy()
ctx.destroy()
