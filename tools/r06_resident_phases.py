#!/usr/bin/env python3
"""tools/r06_resident_phases.py -- phase times of k_cg_resident_q4 at config 2 from a library built with -DFEM_RES_PROFILE
(tools/build_alt.sh resprof "-DFEM_RES_PROFILE" ebe_tile.hip): s_memrealtime (100 MHz) at the phase boundaries of four workgroups,
parked in the tail of the residual history (so -ksp_max_it is set just above the iteration count)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver
from fem_amd.mesh import structured_quad
ctx = capi.Context(0)
m = structured_quad(707)
prob = driver.Problem(ctx, m, f"-ksp_rtol {float(np.float32(1e-9))!r} -mat_mult_type ebe -ksp_cg_resident 1 -ksp_reuse_solution 0 -ksp_max_it 3930")
prob.assemble(); prob.ksp.set_up()
prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
prob.ksp.solve(prob.F, prob.U); prob.ksp.solve(prob.F, prob.U); ctx.synchronize()
its = prob.ksp.get_iteration_number()
h = prob.ksp.get_residual_history(10001)
print("its", its, "resident solves", prob.ksp.get_resident_solves(), "hist len", len(h))
names = ["partials z.z z.r + loads", "logic + x, p update", "tile steps", "p.w + partial", "barrier B", "partials p.w", "r update + partials", "barrier C"]
t = h[3931 - 40:3931 - 40 + 32].reshape(4, 8) * 10.0 / its   # ns per iteration
for q in range(8):
    print(f"  {names[q]:28s}" + "".join(f"{t[wg, q] / 1e3:8.2f}" for wg in range(4)) + "  us per iteration (workgroups 0, 100, 300, 528)")
print("  sum".ljust(30) + "".join(f"{t[wg].sum() / 1e3:8.2f}" for wg in range(4)))
