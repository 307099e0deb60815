import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from fem_amd import capi, driver
from fem_amd.inp import read_inp
from fem_amd.mesh import tile_deck_connected, structured_quad
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = capi.Context(0)
mc = tile_deck_connected(read_inp(os.path.join(ROOT, "tests", "golden", "decks", "cohesive_test_2.inp")), 1000)
def strip(tag):
    for rep in range(3):
        t0 = time.perf_counter()
        r = driver.run_nonlinear(mc, "-ksp_type preonly -pc_type lu", ctx=ctx, rels_mode=1, recover_stress=True, builtin_callbacks=True)
        dt = time.perf_counter() - t0
    print(f"{tag}: strip {dt*1e3:.1f} ms", flush=True)
strip("before")
m2 = structured_quad(707)
p2 = driver.Problem(ctx, m2, "-ksp_rtol 1e-9 -pc_type mg -pc_mg_type additive -mat_mult_type ebe")
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 11
for i in range(nrep):
    if p2.K is not None:
        p2.K.destroy(); p2.K = None
    p2.assemble(); p2.ksp.set_up()
    p2.dmesh.form_rhs(p2.F, 0.0, m2.dt)
    p2.ksp.solve(p2.F, p2.U); ctx.synchronize()
print("mg its", p2.ksp.get_iteration_number())
strip(f"after {nrep} mg steps")
p2.destroy()
strip("after destroy")
