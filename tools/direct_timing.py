#!/usr/bin/env python3
"""tools/direct_timing.py -- factor + solve times of `-ksp_type preonly -pc_type lu` next to CG/Jacobi."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex, structured_quad  # noqa: E402

cases = {"quad200": lambda: structured_quad(200), "quad707": lambda: structured_quad(707),
         "hex20": lambda: structured_hex(20), "hex40": lambda: structured_hex(40)}
names = sys.argv[1:] or list(cases)
ctx = capi.Context(0)
for name in names:
    m = cases[name]()
    sol = {}
    for opts in ("-ksp_type preonly -pc_type lu", "-ksp_type cg -pc_type jacobi -ksp_rtol 1e-12"):
        prob = driver.Problem(ctx, m, opts)
        prob.assemble(); ctx.synchronize()
        t0 = time.perf_counter()
        prob.ksp.set_up(); ctx.synchronize()
        t1 = time.perf_counter()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        prob.ksp.solve(prob.F, prob.U); ctx.synchronize()
        t2 = time.perf_counter()
        sol[opts] = prob.U.get_array()
        print(f"{name} ndof={m.ndof} [{opts}]: set-up {1e3*(t1-t0):.1f} ms, solve {1e3*(t2-t1):.1f} ms, "
              f"its {prob.ksp.get_iteration_number()}, reason {prob.ksp.get_converged_reason()}", flush=True)
        prob.destroy()
    a, b = sol.values()
    print(f"{name}: max |u_lu - u_cg| / max|u| = {np.abs(a-b).max()/np.abs(b).max():.2e}", flush=True)
ctx.destroy()
