import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver
from fem_amd.mesh import structured_hex
ctx = capi.Context(0)
for n in (149, 200, 250, 298):
    m = structured_hex(n)
    prob = driver.Problem(ctx, m, "-pc_type mg -pc_mg_type additive -ksp_rtol 1e-9")
    prob.assemble(); ctx.synchronize()
    for rep in range(2):
        t0 = time.perf_counter(); prob.ksp.set_up(); ctx.synchronize(); t1 = time.perf_counter()
        prob.ksp.set_operators(prob.K)
        print(f"n={n} rep {rep}: mg set-up {1e3*(t1-t0):.1f} ms", flush=True)
    prob.destroy()
