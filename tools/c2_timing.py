#!/usr/bin/env python3
"""tools/c2_timing.py -- config 2 (707^2 quad4, 1.0 M DOF) through the solver options of the KSP surface."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_quad  # noqa: E402

m = structured_quad(707)
ctx = capi.Context(0)
for opts in ("-pc_type jacobi", "-pc_type jacobi -mat_mult_type ebe", "-pc_type mg -pc_mg_type additive",
             "-pc_type mg -pc_mg_type additive -mat_mult_type ebe", "-ksp_type preonly -pc_type lu"):
    for rep in range(2):
        prob = driver.Problem(ctx, m, f"{opts} -ksp_rtol 1e-9")
        ctx.synchronize(); t0 = time.perf_counter()
        prob.assemble(); prob.ksp.set_up(); ctx.synchronize()
        t1 = time.perf_counter()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        prob.ksp.solve(prob.F, prob.U); ctx.synchronize()
        t2 = time.perf_counter()
        if rep:
            print(f"c2 [{opts}]: assemble+set-up {1e3*(t1-t0):.1f} ms, solve {1e3*(t2-t1):.1f} ms, its "
                  f"{prob.ksp.get_iteration_number()}, reason {prob.ksp.get_converged_reason()}, DOF/s {m.ndof/(t2-t0):.3e}", flush=True)
        prob.destroy()
ctx.destroy()
