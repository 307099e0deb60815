"""GPU, BASELINE.json's full sizes (config 2: 707^2 quad4, config 3: 149^3 hex8):
size-independent properties of the hot path, where the oracle would take minutes.

* rigid-body translations lie in the null space of the assembled K (before BCs),
* symmetry:  x.(K y) == y.(K x),
* patch test on the DISTORTED mesh: a linear displacement field imposed on every
  boundary node through the penalty is reproduced exactly in the interior, and
  the recovered nodal stress is the uniform D*eps (isoparametric elements pass
  the patch test for any element shape) -- this exercises assembly, BCs, RHS,
  the CG solve and stress recovery end to end at full size,
* two solves with the same assembled matrix are bitwise identical (deterministic
  two-stage reductions).
"""
import numpy as np
import pytest

from fem_amd import capi, driver
from fem_amd.mesh import structured_hex, structured_quad

pytestmark = pytest.mark.gpu


def _boundary_nodes(m, n):
    c = np.rint(m.coords_unjittered).astype(np.int64)
    return np.nonzero(((c == 0) | (c == n)).any(axis=1))[0]


def _mesh(kind, n, jitter):
    if kind == "hex":
        m0 = structured_hex(n)
        m = structured_hex(n, jitter=jitter)
    else:
        m0 = structured_quad(n)
        m = structured_quad(n, jitter=jitter)
    m.coords_unjittered = m0.coords
    return m


def _dmat(pd, E, nu):
    c = E / ((1 + nu) * (1 - 2 * nu))
    if pd == 2:
        return c * np.array([[1 - nu, nu, 0], [nu, 1 - nu, 0], [0, 0, (1 - 2 * nu) / 2]])
    D = np.zeros((6, 6))
    D[:3, :3] = c * nu
    D[np.arange(3), np.arange(3)] = c * (1 - nu)
    D[np.arange(3, 6), np.arange(3, 6)] = c * (1 - 2 * nu) / 2
    return D


@pytest.mark.parametrize("kind,n", [("quad", 707), ("hex", 149)])
def test_full_size_properties(gpu_ctx, kind, n):
    m = _mesh(kind, n, 0.2)
    pd = m.pdim
    dm = capi.DeviceMesh(gpu_ctx, m)
    K = capi.Mat(dm)
    K.assemble_elements()
    nrows, nnz, _ = K.info()
    assert nrows == m.ndof
    if kind == "hex":
        assert nnz == 809238528
    else:
        assert nnz == 18011536
    X, Y, Z = (capi.Vec(gpu_ctx, m.ndof) for _ in range(3))
    D = capi.Vec(gpu_ctx, m.ndof)
    K.get_diagonal(D)
    dmax = np.abs(D.get_array()).max()
    # rigid translations
    for d in range(pd):
        t = np.zeros((m.nnds, pd)); t[:, d] = 1.0
        X.set_array(t.ravel())
        K.mult(X, Y)
        assert np.abs(Y.get_array()).max() <= 1e-10 * dmax
    # symmetry
    ids = np.arange(m.ndof, dtype=np.float64)
    x, y = np.sin(0.37 * ids) + 0.1, np.cos(0.11 * ids * 1.3)
    X.set_array(x); Z.set_array(y)
    K.mult(X, Y); xKy_a = float(np.dot(y, Y.get_array()))
    K.mult(Z, Y); xKy_b = float(np.dot(x, Y.get_array()))
    assert abs(xKy_a - xKy_b) <= 1e-11 * max(abs(xKy_a), dmax)
    for o in (X, Y, Z, D, K, dm):
        o.destroy()

    # patch test on the distorted mesh
    A = np.array([[1.0e-3, 2.0e-4, -1.0e-4], [-3.0e-4, 5.0e-4, 2.5e-4], [1.5e-4, -2.0e-4, -7.0e-4]])[:pd, :pd]
    b = _boundary_nodes(m, n)
    m.bcnode = np.concatenate([b[:, None], np.zeros((len(b), pd), dtype=np.int64)], axis=1).astype(np.int32)
    m.bcval = m.coords[b] @ A.T
    m.fnode = np.zeros(0, dtype=np.int32); m.fval = np.zeros((0, pd + 2))
    res = driver.run(m, "-ksp_rtol 1e-13 -ksp_max_it 20000", ctx=gpu_ctx)
    assert res.reasons[-1] > 0
    u = res.aggregate_u.reshape(-1, pd)
    uex = m.coords @ A.T
    assert np.abs(u - uex).max() <= 1e-9 * np.abs(uex).max()
    eps = 0.5 * (A + A.T)
    voigt = (np.array([eps[0, 0], eps[1, 1], 2 * eps[0, 1]]) if pd == 2 else
             np.array([eps[0, 0], eps[1, 1], eps[2, 2], 2 * eps[0, 1], 2 * eps[1, 2], 2 * eps[2, 0]]))
    sig = _dmat(pd, m.mats[0, 0], m.mats[0, 1]) @ voigt
    assert np.abs(res.aggregate_stress - sig[None]).max() <= 1e-7 * np.abs(sig).max()
    # determinism: with the SAME assembled Mat two solves are bitwise identical (fixed-order
    # two-stage reductions); a re-assembly may differ in the last bits (fp64 atomics commute
    # but do not associate), like PETSc's off-process stash order
    prob = driver.Problem(gpu_ctx, m, "-ksp_rtol 1e-13 -ksp_max_it 20000")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U); u1 = prob.U.get_array(); it1 = prob.ksp.get_iteration_number()
    prob.ksp.solve(prob.F, prob.U); u2 = prob.U.get_array(); it2 = prob.ksp.get_iteration_number()
    assert it1 == it2 and np.array_equal(u1, u2)
    assert np.abs(u1 - res.aggregate_u).max() <= 1e-11 * np.abs(u1).max()
    prob.destroy()


def test_config3_headline_solve_residual(gpu_ctx):
    """Config 3 exactly as benchmarked (rtol 1e-9, reference default): converged, and
    the true preconditioned residual |D^-1 (F - K u)| meets the PETSc test."""
    m = structured_hex(149)
    prob = driver.Problem(gpu_ctx, m)
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_converged_reason() == 2          # KSP_CONVERGED_RTOL
    its = prob.ksp.get_iteration_number()
    hist = prob.ksp.get_residual_history()
    assert len(hist) == its + 1 and hist[-1] <= capi.REFERENCE_RTOL * hist[0] < hist[-2]
    R, Dg = capi.Vec(gpu_ctx, m.ndof), capi.Vec(gpu_ctx, m.ndof)
    prob.K.mult(prob.U, R)
    prob.K.get_diagonal(Dg)
    r = (prob.F.get_array() - R.get_array()) / Dg.get_array()
    f0 = prob.F.get_array() / Dg.get_array()
    assert np.linalg.norm(r) <= 5 * capi.REFERENCE_RTOL * np.linalg.norm(f0)
    # penalty dofs carry the prescribed values
    u = prob.U.get_array().reshape(-1, 3)
    top = m.bcnode[m.bcnode[:, 3] == 0][:, 0]
    top = top[m.coords[top, 2] == 149]
    assert np.abs(u[top, 2] + 1.49).max() < 1e-9
    R.destroy(); Dg.destroy(); prob.destroy()


def test_config4_global_problem_on_one_gpu(gpu_ctx):
    """BASELINE config 4's global problem (298^3 hex8, 80.2 M DOF) on ONE MI355X: the 288 GB of HBM hold the
    57 GB matrix and the multigrid hierarchy.  Multigrid-preconditioned CG with the element-by-element product
    must converge in a mesh-independent number of iterations, and the answer must satisfy the assembled system
    (residual through the ASSEMBLED MatMult, i.e. both products are checked against each other at this size)."""
    m = structured_hex(298)
    prob = driver.Problem(gpu_ctx, m, "-pc_type mg -pc_mg_type additive -mat_mult_type ebe -ksp_rtol 1e-9")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_converged_reason() == 2
    assert prob.ksp.get_iteration_number() <= 90          # 62 at 149^3, 75-77 here; CG/Jacobi needs 1412
    prob.K.set_mult_type("assembled")
    R, D = capi.Vec(gpu_ctx, m.ndof), capi.Vec(gpu_ctx, m.ndof)
    prob.K.mult(prob.U, R); prob.K.get_diagonal(D)
    r = R.get_array() - prob.F.get_array()
    u = prob.U.get_array()
    scale = D.get_array() * np.abs(u).max() + np.abs(prob.F.get_array()).max()
    assert (np.abs(r) <= 1e-6 * scale).all()             # rtol 1e-9 on the preconditioned norm
    top = np.nonzero(np.isclose(m.coords[:, 2], 298.0))[0]
    assert np.allclose(u.reshape(-1, 3)[top, 2], u.reshape(-1, 3)[top[0], 2], rtol=0, atol=1e-9 * np.abs(u).max())
    for o in (R, D):
        o.destroy()
    prob.destroy()


@pytest.mark.parametrize("jitter", [0.0, 0.15])
def test_config3_against_the_oracle_at_full_size(gpu_ctx, oracle, jitter):
    """Config 3 (149^3 hex8, 10.125 M DOF; then the distorted variant of SURVEY 8(d)) solved at rtol 1e-13 with
    `-mat_mult_type ebe` (the headline's product) AND with the assembled SpMV, against the oracle's CG run to convergence
    on the host cores (oracle/cpu_baseline.c, same KSPCG/PCJACOBI arithmetic as the line-by-line restatement): displacements
    and nodal stresses within north_star's 1e-10 relative.  (At rtol 1e-12 the two runs take the same 929 iterations and agree to
    1e-12 in the displacements, but the nodal stresses -- differences of neighbouring displacements -- only to 0.6-1.0e-10:
    SURVEY 8(d)'s "rtol <= 1e-12 for parity runs" is taken at 1e-13.)"""
    import os
    m = structured_hex(149, jitter=jitter)
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    ob = oracle.cpu_baseline(m, threads, 20000, rtol=1e-13, want_x=True)
    assert ob["reason"] == 2
    uo = ob["x"]
    so, _ = oracle.recover_nodal_stress(m, uo)
    for mt in ("ebe", "assembled"):
        res = driver.run(m, f"-ksp_rtol 1e-13 -ksp_max_it 20000 -mat_mult_type {mt}", ctx=gpu_ctx)
        eu = np.abs(res.aggregate_u - uo).max() / np.abs(uo).max()
        es = np.abs(res.aggregate_stress - so).max() / np.abs(so).max()
        print(f"config 3 jitter {jitter} [{mt}]: {res.its[-1]} CG iterations (oracle {ob['its']}), rel err u {eu:.2e} stress {es:.2e}")
        assert res.reasons[-1] == 2
        assert abs(res.its[-1] - ob["its"]) <= 2
        assert eu <= 1e-10 and es <= 1e-10


def test_config4_in_its_own_partition_against_the_undivided_solve():
    """BASELINE config 4 in its own shape on the one GPU of the box: 2 x 2 x 2 bricks of 149^3 hex8 each (the per-GPU brick of
    `bench.py --gpus 8`: 150^2-node faces, 540 kB per face and exchange, the tile plan with the interface tiles first) as eight
    in-process domains (fem_comm_create_local_group; everything but the RCCL transport is the code the 8-GPU run executes),
    `-mat_mult_type ebe`, with CG/Jacobi, with the exchange beside the interior tiles (`-ksp_halo_overlap 1`) and with
    `-pc_type mg`, against the undivided 298^3 solve (80.2 M DOF) on the same GPU: the same iteration counts (+-2), the same
    field (rtol 1e-9: 1e-8 of the largest displacement), interface copies bitwise equal, the overlapped schedule bitwise equal
    to the plain one, and no collective that grows with the mesh.  (src/main.F90:49-96,274,286-289.)"""
    import time
    from tests import test_gpu_multidomain as md
    n, world = 149, 8
    npr = (n, n, n)
    runs = {}
    log = []
    for name, opts in (("cg_jacobi", "-mat_mult_type ebe"), ("cg_jacobi_overlap", "-mat_mult_type ebe -ksp_halo_overlap 1"),
                       ("pc_mg", "-mat_mult_type ebe -pc_type mg -pc_mg_type additive")):
        t0 = time.perf_counter()
        grid, out = md._run_domains(world, npr, 0.0, opts, extras=False)
        dt = time.perf_counter() - t0
        st = list(md.LAST_STATS)
        md.assert_no_collective_grows_with_the_mesh()
        runs[name] = out
        log.append(f"config 4 as 8 in-process domains of {n}^3 [{opts}]: iterations {sorted({o[2] for o in out})}, reasons "
                   f"{sorted({o[3] for o in out})}, wall {dt:.1f} s (8 domains time-sharing one GPU, mesh generation and fem_mesh_create "
                   f"included); largest all-reduce {max(s_['allreduce_max_doubles'] for s_ in st)} doubles, largest exchange "
                   f"{max(s_['exchange_max_doubles'] for s_ in st)} doubles of {max(s_['interface_slots'] for s_ in st)} interface "
                   f"node slots; exchanges per domain {st[0]['exchange_calls']}, all-reduces {st[0]['allreduce_calls']}")
        print(log[-1], flush=True)
        assert all(o[3] == 2 for o in out)
        assert len({o[2] for o in out}) == 1                    # every domain stops at the same iteration
        # interface copies bitwise equal (canonical-order halo sums)
        for r in range(world):
            for q in range(r + 1, world):
                common, ir, iq = np.intersect1d(out[r][5], out[q][5], return_indices=True)
                assert common.size                              # 2 x 2 x 2: every pair of bricks shares a face, an edge or the corner
                assert np.array_equal(out[r][0].reshape(-1, 3)[ir], out[q][0].reshape(-1, 3)[iq])
                assert np.array_equal(out[r][1][ir], out[q][1][iq])
    # the exchange beside the interior tiles changes the schedule, not one bit of the result
    for a, b in zip(runs["cg_jacobi"], runs["cg_jacobi_overlap"]):
        assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # face sizes: 150^2 nodes per face and neighbour
    st0 = md.LAST_STATS[0]
    assert st0["interface_slots"] == 3 * 150 * 150 + 3 * 150 + 1
    # the undivided problem on the same GPU
    g = structured_hex(2 * n)
    ctx = capi.Context(0)
    for name, opts, its_lo, its_hi in (("cg_jacobi", "-mat_mult_type ebe", 1405, 1420), ("pc_mg", "-mat_mult_type ebe -pc_type mg -pc_mg_type additive", 70, 85)):
        t0 = time.perf_counter()
        one = driver.run(g, opts, ctx=ctx)
        dt = time.perf_counter() - t0
        ug = one.aggregate_u.reshape(-1, 3)
        sg = one.aggregate_stress
        its = runs[name][0][2]
        eu = max(np.abs(o[0].reshape(-1, 3) - ug[o[5]]).max() for o in runs[name]) / np.abs(ug).max()
        es = max(np.abs(o[1] - sg[o[5]]).max() for o in runs[name]) / np.abs(sg).max()
        log.append(f"undivided 298^3 [{opts}]: {one.its[-1]} iterations (8 domains: {its}), wall {dt:.1f} s; 8 domains vs undivided: "
                   f"max rel diff u {eu:.2e}, nodal stress {es:.2e}")
        print(log[-1], flush=True)
        assert one.reasons[-1] == 2 and its_lo <= one.its[-1] <= its_hi
        assert abs(its - one.its[-1]) <= 2
        assert eu <= 1e-8 and es <= 1e-6          # two solves stopped at rtol 1e-9 (nodal stresses: differences of neighbouring displacements)
    ctx.destroy()
    import os
    d = os.environ.get("FEM_AMD_TEST_LOG_DIR")
    if d:
        with open(os.path.join(d, "config4_partitioned_full_size.txt"), "w") as f:
            f.write("\n".join(log) + "\n")


_C2_ORACLE = {}


@pytest.mark.parametrize("fuse", ["", " -ksp_cg_fuse_p 0"])
def test_config2_against_the_oracle_at_full_size(gpu_ctx, oracle, fuse):
    """Config 2 (707^2 quad4 plate, 1.0 M DOF) with the product `bench.py` times on it -- `-mat_mult_type ebe`, i.e.
    k_ebe_tile_q4x, with the p update fused into the product (the default) and as a kernel of its own -- and with the assembled
    SpMV, against the oracle's CG on the host cores: at the reference's rtol (1e-9, what bench.py times) the same iteration
    count (+-2); at rtol 1e-13 displacements and nodal stresses within north_star's 1e-10 (what
    test_config3_against_the_oracle_at_full_size does for config 3).  The iteration counts at 1e-13 are printed, not compared:
    on this plate (a cantilever, cond(D^-1 K) ~ 1e9) CG runs its last thousand iterations on round-off -- the oracle itself needs
    5,562 / 5,571 / 5,573 iterations with 16 / 8 / 4 threads (the order of its reductions), the GPU 4,690 -- and converges to the
    same field."""
    import os
    m = structured_quad(707)
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    if not _C2_ORACLE:
        _C2_ORACLE["ref"] = oracle.cpu_baseline(m, threads, 40000, rtol=capi.REFERENCE_RTOL)
        ob = oracle.cpu_baseline(m, threads, 40000, rtol=1e-13, want_x=True)
        _C2_ORACLE["tight"] = ob
        _C2_ORACLE["stress"] = oracle.recover_nodal_stress(m, ob["x"])[0]
    ob9, ob, so = _C2_ORACLE["ref"], _C2_ORACLE["tight"], _C2_ORACLE["stress"]
    assert ob9["reason"] == 2 and ob["reason"] == 2
    uo = ob["x"]
    for mt in ("ebe", "assembled"):
        r9 = driver.run(m, f"-mat_mult_type {mt}{fuse}", ctx=gpu_ctx, recover_stress=False)
        assert r9.reasons[-1] == 2 and abs(r9.its[-1] - ob9["its"]) <= 2, (r9.its, ob9["its"])
        res = driver.run(m, f"-ksp_rtol 1e-13 -ksp_max_it 40000 -mat_mult_type {mt}{fuse}", ctx=gpu_ctx)
        eu = np.abs(res.aggregate_u - uo).max() / np.abs(uo).max()
        es = np.abs(res.aggregate_stress - so).max() / np.abs(so).max()
        print(f"config 2 [{mt}{fuse}]: rtol 1e-9: {r9.its[-1]} CG iterations (oracle {ob9['its']}); rtol 1e-13: {res.its[-1]} (oracle {ob['its']}), "
              f"rel err u {eu:.2e} stress {es:.2e}")
        assert res.reasons[-1] == 2
        assert eu <= 1e-10 and es <= 1e-10
