"""GPU: error behaviour of the C-ABI (PETSc-style non-zero ierr + message, never a crash or a
silent fallback) and degenerate inputs (empty element list, isolated nodes, ragged fans)."""
import numpy as np
import pytest

from fem_amd import capi, driver
from fem_amd.mesh import Mesh, TRI, QUA, structured_quad

pytestmark = pytest.mark.gpu


def _tri_fan(k):
    """k triangles around node 0: node 0 has k+1 neighbour nodes."""
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    coords = np.concatenate([[[0.0, 0.0]], np.stack([np.cos(ang), np.sin(ang)], axis=1)])
    conn = np.array([[0, 1 + i, 1 + (i + 1) % k] for i in range(k)])
    return Mesh(pdim=2, eltype=np.full(k, TRI), eptr=np.arange(k + 1) * 3, conn=conn.ravel(), mat_id=np.zeros(k),
                coords=coords, mats=[[1.0e5, 0.3, 0, 0, 0]], bcnode=[[1, 0, 0], [2, 0, 0]], bcval=[[0, 0], [0, 0]],
                fnode=[0], fval=[[1.0, 2.0, 0.0, 1.0]])


def test_bad_mesh_inputs(gpu_ctx):
    m = structured_quad(3, 3)
    bad = Mesh(pdim=2, eltype=m.eltype, eptr=m.eptr, conn=np.where(m.conn == 5, 99, m.conn), mat_id=m.mat_id,
               coords=m.coords, mats=m.mats)
    with pytest.raises(capi.FemError, match="node id out of range"):
        capi.DeviceMesh(gpu_ctx, bad)
    bad = Mesh(pdim=2, eltype=m.eltype, eptr=m.eptr, conn=m.conn, mat_id=m.mat_id + 3, coords=m.coords, mats=m.mats)
    with pytest.raises(capi.FemError, match="material id"):
        capi.DeviceMesh(gpu_ctx, bad)
    bad = Mesh(pdim=2, eltype=np.full(m.nels, 4), eptr=m.eptr, conn=m.conn, mat_id=m.mat_id, coords=m.coords, mats=m.mats)
    with pytest.raises(capi.FemError, match="wrong node count"):
        capi.DeviceMesh(gpu_ctx, bad)


def test_vector_and_solver_misuse(gpu_ctx):
    m = structured_quad(4, 4)
    dm = capi.DeviceMesh(gpu_ctx, m)
    K = capi.Mat(dm)
    x, y = capi.Vec(gpu_ctx, m.ndof), capi.Vec(gpu_ctx, m.ndof + 1)
    with pytest.raises(capi.FemError, match="size mismatch"):
        K.mult(x, y)
    ksp = capi.KSP(gpu_ctx)
    with pytest.raises(capi.FemError, match="no operator"):
        ksp.set_up()
    with pytest.raises(capi.FemError, match="unsupported"):
        ksp.set_type("gmres")
    with pytest.raises(capi.FemError, match="unsupported"):
        ksp.set_pc_type("ilu")
    with pytest.raises(capi.FemError, match="unknown option"):
        ksp.set_from_options("-ksp_foo 3")
    with pytest.raises(capi.FemError, match="bad value"):
        ksp.set_from_options("-ksp_rtol abc")
    ksp.set_from_options("-ksp_type cg -pc_type none -ksp_rtol 1e-8 -ksp_atol 1e-40 -ksp_divtol 1e6 -ksp_max_it 77")
    m2 = Mesh(pdim=2, eltype=m.eltype, eptr=m.eptr, conn=m.conn, mat_id=m.mat_id, coords=m.coords, mats=m.mats,
              telsd=[[0, 7]], tval=[[1.0, 0.0, 0.0, 1.0]])
    with pytest.raises(capi.FemError, match="bad side"):
        dm.form_rhs(x, 0.0, 1.0, mesh=m2)
    for o in (ksp, x, y, K, dm):
        o.destroy()


def test_max_iterations_and_unpreconditioned(gpu_ctx, oracle):
    m = structured_quad(12, 9, jitter=0.1)
    dm = capi.DeviceMesh(gpu_ctx, m)
    K = capi.Mat(dm); K.assemble_elements(); K.apply_kbc()
    F, U = capi.Vec(gpu_ctx, m.ndof), capi.Vec(gpu_ctx, m.ndof)
    dm.form_rhs(F, 0.0, m.dt)
    ksp = capi.KSP(gpu_ctx); ksp.set_operators(K)
    ksp.set_tolerances(rtol=1e-12, max_it=7)
    ksp.solve(F, U)
    assert ksp.get_converged_reason() == -3 and ksp.get_iteration_number() == 7      # KSP_DIVERGED_ITS
    rp, ci = oracle.csr_pattern(m); val = oracle.assemble(m, rp, ci)
    xo, its, reason, rn, hist = oracle.ksp_cg_jacobi(rp, ci, val, oracle.form_rhs(m, 0.0, m.dt), rtol=1e-12, maxit=7,
                                                     hist_len=16)
    assert reason == -3 and its == 7
    h = ksp.get_residual_history()
    assert len(h) == 8 and np.allclose(h, hist, rtol=1e-9)
    assert np.abs(U.get_array() - xo).max() <= 1e-9 * np.abs(xo).max()
    # zero right-hand side: converged at iteration 0 (PETSc: rnorm <= ttol with ttol = abstol -> CONVERGED_ATOL)
    F.set(0.0)
    ksp.set_tolerances(max_it=1000)
    ksp.solve(F, U)
    assert ksp.get_iteration_number() == 0 and ksp.get_converged_reason() == 3 and U.norm() == 0.0
    for o in (ksp, F, U, K, dm):
        o.destroy()


def test_degenerate_meshes(gpu_ctx, oracle):
    # no elements at all: every node is isolated, the Mat holds empty diagonal blocks
    m = Mesh(pdim=2, eltype=np.zeros(0), eptr=[0], conn=np.zeros(0), mat_id=np.zeros(0),
             coords=np.random.default_rng(0).random((70, 2)), mats=[[1.0, 0.3, 0, 0, 0]])
    dm = capi.DeviceMesh(gpu_ctx, m)
    K = capi.Mat(dm); K.assemble_elements()
    rp, ci, v = K.get_csr()
    assert len(v) == 70 * 4 and not v.any()
    K.destroy(); dm.destroy()
    # a fan of 40 triangles: one long row (41 neighbours) among short ones; solve against the oracle
    f = _tri_fan(40)
    from fem_amd import driver
    res = driver.run(f, "-ksp_rtol 1e-13", ctx=gpu_ctx)
    u, s, _ = oracle.run(f, "cg", rtol=1e-13)
    assert np.abs(res.aggregate_u - u).max() <= 1e-10 * np.abs(u).max()
    # more neighbours than the symbolic phase supports: loud error, not corruption
    big = _tri_fan(120)
    dm = capi.DeviceMesh(gpu_ctx, big)
    with pytest.raises(capi.FemError, match="neighbour nodes"):
        capi.Mat(dm)
    dm.destroy()


def test_device_pool_reuses_blocks_and_context_destroy_returns_them():
    """Re-creating the matrix must not grow the device footprint (released blocks are handed out
    again), and destroying the context gives the cache back to the driver."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def used():
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return total.value - free.value

    ctx = capi.Context(0)
    from fem_amd.mesh import structured_hex
    m = structured_hex(40)
    prob = driver.Problem(ctx, m, "-pc_type jacobi")
    prob.assemble(); prob.ksp.set_up(); ctx.synchronize()
    base = used()
    for _ in range(3):
        prob.K.destroy(); prob.K = None
        prob.assemble(); prob.ksp.set_up(); ctx.synchronize()
        assert used() <= base + (64 << 20)
    prob.destroy()
    held = used()
    ctx.destroy()
    assert used() < held - (100 << 20)      # the 140 MB matrix blocks went back


def test_releasing_a_block_does_not_stall_the_device():
    """pool.cpp: a released block takes an event per stream of the library and goes into the cache at once -- no device-wide
    synchronisation at the release (a host that frees inside its increment loop, src/main.F90:300-312 at the end of every run of a
    long-lived process).  With about 0.2 s of work queued on the context's stream a Vec is destroyed in far less than that; the
    next request of that size gets the block once that work has finished, and uses it safely."""
    import time
    ctx = capi.Context(0)
    n = 1 << 25
    a, b, c = capi.Vec(ctx, n), capi.Vec(ctx, n), capi.Vec(ctx, n)
    a.set(1.0); b.set(0.0); c.set(3.0)
    ctx.synchronize()
    t0 = time.perf_counter()
    b.axpy(1.0, a)
    ctx.synchronize()
    one = time.perf_counter() - t0
    reps = max(50, int(0.2 / max(one, 1e-5)))
    t0 = time.perf_counter()
    for _ in range(reps):
        c.axpy(1.0, a)                       # the block about to be released is in use by queued work
    t_queue = time.perf_counter() - t0
    t0 = time.perf_counter()
    c.destroy()
    t_release = time.perf_counter() - t0
    d = capi.Vec(ctx, n)                     # the same size: the cached block, once the queued work is through
    d.set(7.0)
    ctx.synchronize()
    t_all = time.perf_counter() - t0
    assert t_release < 0.25 * reps * one + 0.005, (t_release, reps * one, t_queue)
    assert np.all(d.get_array() == 7.0)
    b.axpy(1.0, a); ctx.synchronize()
    assert np.all(b.get_array() == 2.0) and t_all > 0
    a.destroy(); b.destroy(); d.destroy(); ctx.destroy()


def test_full_device_pool_makes_room_for_the_blocks_in_use():
    """A cache of released blocks that is full (FEM_AMD_POOL_MAX_GB; here 0.05 GB in a child process) gives its largest blocks back
    to the driver instead of refusing every later release: problems of several sizes one after the other, then a small one again
    and again -- same answers every time, and the footprint of a cycle does not grow."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes as C, numpy as np
from fem_amd import capi, driver
from fem_amd.mesh import structured_hex
hip = C.CDLL("libamdhip64.so")
def used():
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return total.value - free.value
ctx = capi.Context(0)
ref, first = None, None
for cycle in range(3):
    for n in (30, 22, 26, 8, 8):
        r = driver.run(structured_hex(n, jitter=0.1), "-ksp_rtol 1e-10", ctx=ctx)
        assert r.reasons[-1] > 0
        if n == 8:
            ref = r.aggregate_u if ref is None else ref
            assert np.array_equal(ref, r.aggregate_u)
    if first is None:
        first = used()
    assert used() <= first + (64 << 20), (used(), first)     # (the footprint of a cycle does not grow: what the cache drops goes back)
ctx.destroy()
print("POOL_OK")
'''
    env = dict(os.environ, FEM_AMD_POOL_MAX_GB="0.05", PYTHONPATH=root)
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "POOL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
