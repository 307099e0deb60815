#!/usr/bin/env python3
"""tools/first_step_timing.py [n] -- what the first step on a new mesh pays beyond a later one (config 3 by default): symbolic
phase, element loop, and the set-up of the element-by-element product (tile plan), each timed with a device synchronisation."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
ctx = capi.Context(0)
m = structured_hex(n)
t = [time.perf_counter()]


def lap(label):
    ctx.synchronize()
    t.append(time.perf_counter())
    print(f"{label}: {(t[-1] - t[-2]) * 1e3:.1f} ms", flush=True)


prob = driver.Problem(ctx, m, "")
lap("mesh to the device, Vec / KSP objects")
for rep in range(2):
    prob.create_matrix(); lap(f"[{rep}] fem_mat_create_aij")
    prob.assemble(); lap(f"[{rep}] element loop + ApplyKBC")
    prob.K.set_mult_type("ebe"); lap(f"[{rep}] fem_mat_set_mult_type(ebe)")
    prob.ksp.set_operators(prob.K); prob.ksp.set_up(); lap(f"[{rep}] fem_ksp_set_up")
    X, Y = capi.Vec(ctx, m.ndof), capi.Vec(ctx, m.ndof)
    prob.K.mult(X, Y); lap(f"[{rep}] first MatMult")
    prob.K.mult(X, Y); lap(f"[{rep}] second MatMult")
    X.destroy(); Y.destroy()
    if rep == 0:
        prob.K.destroy(); prob.K = None; lap("Mat destroyed")
print(prob.K.get_mult_info())
prob.destroy(); ctx.destroy()
