#!/usr/bin/env python3
"""tools/r06_resident_check.py -- KSPCG as one resident launch (-ksp_cg_resident) against the launch-per-kernel path: small quad4
meshes against each other and the oracle, then config 2 (707^2 quad4) timed both ways."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_quad  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

ctx = capi.Context(0)
for n, jit in (((9, 7), 0.2), ((40, 33), 0.0), ((70, 45), 0.15), ((200, 150), 0.1)):
    m = structured_quad(*n, jitter=jit)
    uo, so, ito = Oracle().run(m, "cg", rtol=1e-13)
    out = {}
    for res in (0, 1):
        prob = driver.Problem(ctx, m, f"-ksp_rtol 1e-13 -mat_mult_type ebe -ksp_cg_resident {res}")
        prob.assemble(); prob.ksp.set_up()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        prob.ksp.solve(prob.F, prob.U)
        out[res] = (prob.U.get_array(), prob.ksp.get_iteration_number(), prob.ksp.get_converged_reason(), prob.ksp.get_resident_solves(),
                    prob.ksp.get_residual_norm())
        prob.destroy()
    e01 = np.abs(out[0][0] - out[1][0]).max() / np.abs(uo).max()
    eo = np.abs(out[1][0] - uo).max() / np.abs(uo).max()
    print(f"quad {n} jitter {jit}: its launch {out[0][1]} resident {out[1][1]} oracle {ito[-1]}, reasons {out[0][2]} {out[1][2]}, resident solves "
          f"{out[1][3]}, |u_res - u_launch| {e01:.2e}, |u_res - u_oracle| {eo:.2e}, rnorm {out[0][4]:.3e} {out[1][4]:.3e}", flush=True)

m = structured_quad(707)
for res in (0, 1, 0, 1):
    prob = driver.Problem(ctx, m, f"-ksp_rtol {float(np.float32(1e-9))!r} -mat_mult_type ebe -ksp_cg_resident {res} -ksp_reuse_solution 0")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U); ctx.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        prob.ksp.solve(prob.F, prob.U); ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    its = prob.ksp.get_iteration_number()
    print(f"config 2, -ksp_cg_resident {res}: {its} its, reason {prob.ksp.get_converged_reason()}, resident solves {prob.ksp.get_resident_solves()}, "
          f"solve {min(ts) * 1e3:.2f} ms = {min(ts) * 1e6 / its:.2f} us per iteration", flush=True)
    prob.destroy()
ctx.destroy()
