#!/usr/bin/env python3
"""tools/mg_setup_steady.py [n] [reps] -- steady-state steps of config 3 with -pc_type mg on ONE mesh (as bench.py's pc_mg leg: Mat_K
destroyed and re-created, element loop, ApplyKBC, multigrid set-up, FormRHS, solve), times of the parts; for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver
from fem_amd.mesh import structured_hex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = structured_hex(n)
ctx = capi.Context(0)
prob = driver.Problem(ctx, m, "-pc_type mg -pc_mg_type additive -mat_mult_type ebe -ksp_reuse_solution 0")
for rep in range(reps + 1):
    if prob.K is not None:
        prob.K.destroy(); prob.K = None
    ctx.synchronize(); t0 = time.perf_counter()
    prob.assemble(); ctx.synchronize(); t1 = time.perf_counter()
    prob.ksp.set_up(); ctx.synchronize(); t2 = time.perf_counter()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U); ctx.synchronize(); t3 = time.perf_counter()
    if rep:
        print(f"step {rep}: assemble {1e3*(t1-t0):.2f} ms, set-up {1e3*(t2-t1):.2f} ms, solve {1e3*(t3-t2):.2f} ms, its {prob.ksp.get_iteration_number()}, "
              f"DOF/s {m.ndof/(t3-t0):.3e}", flush=True)
prob.destroy(); ctx.destroy()
