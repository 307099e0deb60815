#!/usr/bin/env python3
"""tools/c4_single_gpu.py -- config 4's global problem (298^3 hex8, 80.2 M DOF) on ONE MI355X:
the 288 GB of HBM hold the whole matrix (~57 GB), so the iteration growth of CG/Jacobi from
config 3 to config 4 can be measured without a second GPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

shapes = {"c3": (149, 149, 149), "c4": (298, 298, 298), "slab2": (298, 149, 149), "slab4": (298, 298, 149)}
names = sys.argv[1:] or ["c3", "slab2", "slab4", "c4"]
ctx = capi.Context(0)
for name in names:
    t0 = time.perf_counter()
    m = structured_hex(*shapes[name])
    print(f"{name}: mesh {shapes[name]} built on the host in {time.perf_counter()-t0:.1f} s, ndof {m.ndof}", flush=True)
    for pc in ("jacobi", "jacobi -mat_mult_type ebe", "mg -pc_mg_type additive", "mg -pc_mg_type additive -mat_mult_type ebe"):
        prob = driver.Problem(ctx, m, f"-pc_type {pc} -ksp_rtol 1e-9")
        ctx.synchronize(); t0 = time.perf_counter()
        prob.assemble(); prob.ksp.set_up(); ctx.synchronize()
        t1 = time.perf_counter()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        prob.ksp.solve(prob.F, prob.U); ctx.synchronize()
        t2 = time.perf_counter()
        print(f"{name} pc={pc}: assemble+setup {1e3*(t1-t0):.0f} ms, solve {1e3*(t2-t1):.0f} ms, its "
              f"{prob.ksp.get_iteration_number()}, reason {prob.ksp.get_converged_reason()}, "
              f"DOF/s {m.ndof/(t2-t0):.3e}", flush=True)
        prob.destroy()
    del m
ctx.destroy()
