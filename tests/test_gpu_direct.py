"""GPU: direct solve behind the KSP surface, `-ksp_type preonly -pc_type lu` (the
reference's own production pairing KSPPREONLY + PCLU, src/main.F90:354-358).  The oracle's dense
LU of the same assembled system is the checker; larger decks are checked against the CG path."""
import os

import numpy as np
import pytest

from fem_amd import capi, driver
from fem_amd.inp import read_inp
from fem_amd.mesh import structured_hex, structured_quad, hex_to_tets, quad_to_mixed

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LU = "-ksp_type preonly -pc_type lu"


def _small():
    return {
        "hex5_jitter": structured_hex(5, 4, 3, jitter=0.2),
        "hex_67nodes_x": structured_hex(66, 2, 2, jitter=0.1),
        "quad9_jitter": structured_quad(9, 7, jitter=0.2),
        "quad_strip": structured_quad(60, 1, jitter=0.1),         # half bandwidth below one 64-column block
        "tet": hex_to_tets(structured_hex(4, 3, 3, jitter=0.15)),
        "mixed_tri_qua": quad_to_mixed(structured_quad(8, 5, jitter=0.1)),
        "single_hex": structured_hex(1, 1, 1),
    }


@pytest.mark.parametrize("name", list(_small()))
def test_direct_solve_matches_oracle_dense_lu(gpu_ctx, oracle, name):
    m = _small()[name]
    res = driver.run(m, LU, ctx=gpu_ctx)
    u, s, _ = oracle.run(m, "lu")
    assert res.reasons[-1] == 4 and res.its[-1] == 1          # KSP_CONVERGED_ITS after one application
    # two factorisations of the same matrix agree to cond(K) * eps: the slender strip (a cantilever,
    # cond ~ (L/h)^4) gets the looser bound
    tol = 1e-8 if name == "quad_strip" else 1e-11
    assert np.abs(res.aggregate_u - u).max() <= tol * np.abs(u).max()
    assert np.abs(res.aggregate_stress - s).max() <= max(1e-9, 100 * tol) * np.abs(s).max()


@pytest.mark.parametrize("deck", ["elastic_test.inp", "lin_two_quads_qs.inp"])
def test_reference_decks_with_the_reference_solver_choice(gpu_ctx, oracle, deck):
    """The reference's example decks, solved the way the reference solves them (factorisation,
    every increment re-uses it)."""
    m = read_inp(os.path.join(ROOT, "tests", "golden", "decks", deck))
    res = driver.run(m, LU, ctx=gpu_ctx)
    u, s, its = oracle.run(m, "lu")
    assert len(res.its) == len(its) and all(r == 4 for r in res.reasons)
    assert np.abs(res.aggregate_u - u).max() <= 1e-11 * np.abs(u).max()
    assert np.abs(res.aggregate_stress - s).max() <= 1e-9 * max(np.abs(s).max(), 1e-300)


@pytest.mark.parametrize("name,mesh", [
    ("quad200x150", lambda: structured_quad(200, 150, jitter=0.2)),   # 60k dofs, band ~ 300: many blocks, ragged tiles
    ("hex20", lambda: structured_hex(20, 19, 18, jitter=0.2)),        # 25k dofs, band ~ 1.2k
    ("tet12", lambda: hex_to_tets(structured_hex(12, 11, 10, jitter=0.15))),
])
def test_direct_solve_agrees_with_cg_on_larger_decks(gpu_ctx, name, mesh):
    m = mesh()
    d = driver.run(m, LU, ctx=gpu_ctx)
    c = driver.run(m, "-ksp_type cg -pc_type jacobi -ksp_rtol 1e-13", ctx=gpu_ctx)
    assert d.reasons[-1] == 4 and c.reasons[-1] == 2
    assert np.abs(d.aggregate_u - c.aggregate_u).max() <= 1e-9 * np.abs(c.aggregate_u).max()
    assert np.abs(d.aggregate_stress - c.aggregate_stress).max() <= 1e-8 * np.abs(c.aggregate_stress).max()


def test_direct_residual_and_cholesky_alias(gpu_ctx):
    """||K x - b|| of the factorisation's answer, through MatMult of the same Mat."""
    m = structured_hex(14, 9, 8, jitter=0.2)
    prob = driver.Problem(gpu_ctx, m, "-ksp_type preonly -pc_type cholesky")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_converged_reason() == 4 and prob.ksp.get_iteration_number() == 1
    R = capi.Vec(gpu_ctx, m.ndof)
    prob.K.mult(prob.U, R)
    r = R.get_array() - prob.F.get_array()
    b = prob.F.get_array()
    # penalty rows carry 1e30 * u: compare per row against the row's own scale
    diag = capi.Vec(gpu_ctx, m.ndof); prob.K.get_diagonal(diag)
    scale = diag.get_array() * np.abs(prob.U.get_array()).max() + np.abs(b).max()
    assert (np.abs(r) <= 1e-10 * scale).all()
    # the same factor serves a second right-hand side
    u1 = prob.U.get_array()
    prob.ksp.solve(prob.F, prob.U)
    assert np.array_equal(u1, prob.U.get_array())
    for o in (R, diag):
        o.destroy()
    prob.destroy()


def test_unsupported_pairings_fail_loudly(gpu_ctx):
    m = structured_hex(3, 2, 2)
    for opts in ("-ksp_type preonly -pc_type jacobi", "-ksp_type cg -pc_type lu", "-ksp_type preonly"):
        prob = driver.Problem(gpu_ctx, m, opts)
        prob.assemble()
        with pytest.raises(capi.FemError):
            prob.ksp.set_up()
        prob.destroy()


def test_direct_solve_config2_full_size(gpu_ctx):
    """BASELINE config 2 at full size (707^2 quad4, 1.0 M DOF): the banded factor (11 GB) against
    CG/Jacobi at rtol 1e-13 and against the residual of the assembled operator."""
    m = structured_quad(707)
    prob = driver.Problem(gpu_ctx, m, LU)
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_converged_reason() == 4
    u_lu = prob.U.get_array()
    R, D = capi.Vec(gpu_ctx, m.ndof), capi.Vec(gpu_ctx, m.ndof)
    prob.K.mult(prob.U, R); prob.K.get_diagonal(D)
    r = R.get_array() - prob.F.get_array()
    scale = D.get_array() * np.abs(u_lu).max() + np.abs(prob.F.get_array()).max()
    assert (np.abs(r) <= 1e-9 * scale).all()
    prob.ksp.set_from_options("-ksp_type cg -pc_type jacobi -ksp_rtol 1e-13")
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_converged_reason() == 2
    assert np.abs(u_lu - prob.U.get_array()).max() <= 1e-8 * np.abs(u_lu).max()
    for o in (R, D):
        o.destroy()
    prob.destroy()


def _overwrite_with(prob, rng, scale_diag):
    """Replaces every stored entry of prob.K through MatSetValues(INSERT) by an unsymmetric matrix whose diagonal is
    `scale_diag` times smaller than what stands below it, so that every elimination step exchanges rows.  Returns it dense."""
    rp, ci, _ = prob.K.get_csr()
    n = len(rp) - 1
    A = np.zeros((n, n))
    for r in range(n):
        cols = ci[rp[r]:rp[r + 1]]
        v = rng.uniform(-1.0, 1.0, len(cols))
        v[cols == r] *= scale_diag
        prob.K.set_values([r], cols, v, "insert")
        A[r, cols] = v
    return A


@pytest.mark.parametrize("name,mesh", [
    ("one_quad", lambda: structured_quad(1, 1)),                      # 8 dofs: one component block (k_comp_lu)
    ("one_hex", lambda: structured_hex(1, 1, 1)),                     # 24 dofs, block size 3
    ("quad3x2", lambda: structured_quad(3, 2, jitter=0.1)),           # 24 dofs
    ("quad12x9", lambda: structured_quad(12, 9, jitter=0.1)),         # 260 dofs > 64: the band LU, half bandwidth > 21: k_band_lu
    ("quad90x2", lambda: structured_quad(90, 2, jitter=0.1)),         # narrow bands: one wavefront, window in registers (k_band_lu_wave<8|14>)
    ("quad300x1", lambda: structured_quad(300, 1, jitter=0.1)),       # 1,204 dofs, half bandwidth ~ 5
    ("quad70x5", lambda: structured_quad(70, 5, jitter=0.1)),         # half bandwidth ~ 15-17: k_band_lu_wave<21>
    ("hex40x1x1", lambda: structured_hex(40, 1, 1, jitter=0.1)),      # block size 3
])
def test_lu_with_row_exchanges_against_numpy(gpu_ctx, name, mesh):
    """Partial pivoting that really exchanges rows (round-3 advice: the batched block LU swapped whole rows, multipliers
    included, while its solve assumed they stay): MatSetValues(INSERT) of an unsymmetric matrix with a small diagonal on the
    sparsity of the mesh, `-ksp_type preonly -pc_type lu`, against numpy.linalg.solve."""
    m = mesh()
    rng = np.random.default_rng(7)
    prob = driver.Problem(gpu_ctx, m, LU)
    prob.create_matrix()
    prob.K.zero_entries()
    prob.K.assemble_elements()
    A = _overwrite_with(prob, rng, 1e-3)
    rp, ci, v = prob.K.get_csr()
    B = np.zeros_like(A)
    for r in range(len(rp) - 1):
        B[r, ci[rp[r]:rp[r + 1]]] = v[rp[r]:rp[r + 1]]
    assert np.array_equal(A, B)                                        # MatSetValues(INSERT) wrote what MatView reads
    prob.ksp.set_operators(prob.K)
    prob.ksp.set_up()
    b = rng.uniform(-1.0, 1.0, m.ndof)
    prob.F.set_array(b)
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_converged_reason() == 4
    x = prob.U.get_array()
    ref = np.linalg.solve(A, b)
    assert np.abs(A @ x - b).max() <= 1e-10 * np.abs(b).max() * np.linalg.cond(A)
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max() * max(1.0, np.linalg.cond(A) * 1e-6)
    prob.destroy()


def test_mat_set_values_semantics(gpu_ctx):
    """ADD adds, INSERT overwrites, negative indices are skipped, an entry outside the sparsity is refused."""
    m = structured_quad(3, 1)
    prob = driver.Problem(gpu_ctx, m, "")
    prob.assemble()
    rp, ci, v0 = prob.K.get_csr()
    prob.K.set_values([2, -1, 3], [3, 2, -3], 1e9 * np.arange(9.0).reshape(3, 3), "add")     # (node 1: not on the clamped edge)
    _, _, v1 = prob.K.get_csr()
    d = {(r, int(c)): v1[k] - v0[k] for r in range(len(rp) - 1) for k, c in zip(range(rp[r], rp[r + 1]), ci[rp[r]:rp[r + 1]])}
    want = {(2, 3): 0.0, (2, 2): 1e9, (3, 3): 6e9, (3, 2): 7e9}
    assert all(abs(d[k] - want.get(k, 0.0)) <= 1e-3 for k in d), {k: v for k, v in d.items() if v != 0.0}
    prob.K.set_values([2], [2], [5.5], "insert")
    _, _, v2 = prob.K.get_csr()
    k22 = rp[2] + list(ci[rp[2]:rp[3]]).index(2)
    assert v2[k22] == 5.5
    with pytest.raises(capi.FemError):
        prob.K.set_values([2], [m.ndof - 1], [1.0], "add")            # node 1 and the last node share no element
    prob.destroy()


@pytest.mark.parametrize("name,mesh,opts", [
    ("quad_strip_2000", lambda: structured_quad(2000, 1, jitter=0.1), "-ksp_type preonly -pc_type lu"),
    ("quad_90x2", lambda: structured_quad(90, 2, jitter=0.1), "-ksp_type preonly -pc_type lu"),
])
def test_band_lu_by_one_wavefront_equals_the_column_walk(gpu_ctx, monkeypatch, capfd, name, mesh, opts):
    """Narrow bands (3 kl + 1 <= 64) are factorised and solved by one wavefront with the window of a step in registers
    (k_band_lu_wave / k_band_solve_wave); FEM_AMD_LU_WAVE=0 selects the column walk of one workgroup (k_band_lu).  Same pivots,
    same multipliers: on an unsymmetric matrix that exchanges rows at every step the two answers agree to round-off, and both
    satisfy the system."""
    m = mesh()
    rng = np.random.default_rng(3)
    monkeypatch.setenv("FEM_AMD_DIRECT_VERBOSE", "1")
    xs = {}
    for wave in ("1", "0"):
        monkeypatch.setenv("FEM_AMD_LU_WAVE", wave)
        prob = driver.Problem(gpu_ctx, m, opts)
        prob.create_matrix()
        prob.K.zero_entries(); prob.K.assemble_elements()
        rng2 = np.random.default_rng(5)
        rp, ci, _ = prob.K.get_csr()
        import scipy.sparse as sp
        vals = rng2.uniform(-1.0, 1.0, len(ci))
        vals[ci == np.repeat(np.arange(len(rp) - 1), np.diff(rp))] *= 1e-3      # small diagonal: a row exchange at every step
        for r in range(len(rp) - 1):
            prob.K.set_values([r], ci[rp[r]:rp[r + 1]], vals[rp[r]:rp[r + 1]], "insert")
        prob.ksp.set_operators(prob.K); prob.ksp.set_up()
        b = rng.uniform(-1.0, 1.0, m.ndof) if "b" not in xs else xs["b"]
        xs["b"] = b
        prob.F.set_array(b)
        prob.ksp.solve(prob.F, prob.U)
        assert prob.ksp.get_converged_reason() == 4
        xs[wave] = prob.U.get_array()
        A = sp.csr_matrix((vals, ci, rp))
        assert np.abs(A @ xs[wave] - b).max() <= 1e-8 * np.abs(b).max() * max(1.0, np.abs(xs[wave]).max())
        prob.destroy()
        log = capfd.readouterr().err
        assert ("one wavefront" in log) == (wave == "1"), log
    assert np.abs(xs["1"] - xs["0"]).max() <= 1e-9 * np.abs(xs["0"]).max()


# The three forms of the reduction: "pcr" = one resident launch per solve, parallel cyclic reduction (the default where a wavefront per
# block fits the chip: every mesh of these tests); "cr" = one resident launch, cyclic reduction with its way up (FEM_AMD_LU_BCR_PCR=0:
# what long strips get); "0" = a launch per level with the trial reduction at set-up (FEM_AMD_LU_BCR_RESIDENT=0: what the resident
# launches fall back to when their grid barrier gives up)
def _bcr_form(monkeypatch, form):
    monkeypatch.setenv("FEM_AMD_LU_BCR_RESIDENT", "0" if form == "0" else "1")
    monkeypatch.setenv("FEM_AMD_LU_BCR_PCR", "1" if form == "pcr" else "0")


@pytest.mark.parametrize("resident", ["pcr", "cr", "0"])
@pytest.mark.parametrize("name,mesh", [("quad_400x1", lambda: structured_quad(400, 1, jitter=0.1)),      # half bandwidth 7 (blocks of 9 rows)
                                       ("quad_300x2", lambda: structured_quad(300, 2, jitter=0.1)),      # 11 (13)
                                       ("quad_320x3", lambda: structured_quad(320, 3, jitter=0.1)),      # 15 (17)
                                       ("quad_330x4", lambda: structured_quad(330, 4, jitter=0.1)),      # 19 (21)
                                       ("quad_70x2", lambda: structured_quad(70, 2, jitter=0.1))])       # 11 (13), 33 blocks: odd counts on the way down
def test_block_cyclic_reduction_on_a_stiffness_like_band(gpu_ctx, monkeypatch, capfd, name, mesh, resident):
    """Narrow bands with many columns go through block cyclic reduction (k_bcr_*): Mat_K with its penalties plus a small unsymmetric
    perturbation (what a cohesive tangent does to the Newton Jacobian) -- the set-up's trial reduction meets no small pivot, the
    solve passes its backward-error test, and the answer agrees with scipy's sparse LU and with the band LU (FEM_AMD_LU_BCR=0)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    m = mesh()
    monkeypatch.setenv("FEM_AMD_DIRECT_VERBOSE", "1")
    _bcr_form(monkeypatch, resident)
    xs = {}
    for bcr in ("1", "0"):
        monkeypatch.setenv("FEM_AMD_LU_BCR", bcr)
        prob = driver.Problem(gpu_ctx, m, LU)
        prob.assemble()
        rp, ci, v = prob.K.get_csr()
        # an unsymmetric term of the size of the entries themselves on one off-diagonal pair per tenth row
        rows = np.arange(3, m.ndof - 3, 10)
        scale = np.abs(v[v < 1e29]).max()
        for r in rows:
            cols = ci[rp[r]:rp[r + 1]]
            c = int(cols[cols != r][0])
            prob.K.set_values([r], [c], [0.05 * scale], "add")
        rp, ci, v = prob.K.get_csr()
        prob.ksp.set_operators(prob.K); prob.ksp.set_up()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        b = prob.F.get_array()
        prob.ksp.solve(prob.F, prob.U)
        assert prob.ksp.get_converged_reason() == 4
        xs[bcr] = prob.U.get_array()
        log = capfd.readouterr().err
        if bcr == "1":
            assert "block cyclic reduction" in log and "used" in log, log
            # (the parallel form may fail its backward-error test on these slender cantilevers -- quad_400x1 does, at 1.1e-10 -- and
            # hand over to the form with a way up; that one, and the launches per level, must pass)
            assert log.count("backward error") == log.count("parallel form: backward error"), log
            assert log.count("parallel form:") <= (1 if resident == "pcr" else 0) and "band LU from here on" not in log, log
        else:
            assert "block cyclic reduction" not in log
        A = sp.csr_matrix((v, ci, rp))
        d = A.diagonal()
        ref = sla.spsolve(sp.diags(1.0 / d) @ A, b / d)          # (rows scaled: the penalties are 1e30)
        # a slender cantilever (cond ~ (L / h)^4 ~ 1e9 ... 1e10): the answer is judged by its componentwise backward error; two
        # backward-stable solvers agree in the forward error only to cond x eps
        x = xs[bcr]
        bwd = np.abs(b - A @ x) / (abs(A) @ np.abs(x) + np.abs(b) + 1e-300)
        assert bwd.max() <= 1e-10, bwd.max()
        assert np.abs(x - ref).max() <= 1e-5 * np.abs(ref).max()
        prob.destroy()
    assert np.abs(xs["1"] - xs["0"]).max() <= 1e-5 * np.abs(xs["0"]).max()


@pytest.mark.parametrize("resident", ["pcr", "cr", "0"])
def test_null_pivots_are_fixed_the_same_way_by_every_band_kernel(gpu_ctx, monkeypatch, capfd, resident):
    """A free-floating strip (no Dirichlet condition: three rigid-body modes) with a self-equilibrated load: the factorisation
    meets pivots below 1e-6 of the largest entry and fixes those unknowns to zero (what MUMPS does under ICNTL(24) = 1,
    src/main.F90:368).  The band LU by one wavefront, the column walk of one workgroup and the way through block cyclic
    reduction (whose trial reduction must notice the small pivot and hand over to the band LU) give the same field, and it
    satisfies the equations it did not drop."""
    import scipy.sparse as sp
    m = structured_quad(120, 2, jitter=0.1)
    m.bcnode = np.zeros((0, 3), dtype=np.int32); m.bcval = np.zeros((0, 2))
    # two opposite forces along x on the two ends (in equilibrium: no net force, no net moment)
    left = int(np.argmin(m.coords[:, 0] + 1e-3 * np.abs(m.coords[:, 1] - 1.0)))
    right = int(np.argmax(m.coords[:, 0] - 1e-3 * np.abs(m.coords[:, 1] - 1.0)))
    m.fnode = np.array([left, right], dtype=np.int32)
    m.fval = np.array([[-1e6, 0.0, 0.0, m.t], [1e6, 0.0, 0.0, m.t]])
    monkeypatch.setenv("FEM_AMD_DIRECT_VERBOSE", "1")
    _bcr_form(monkeypatch, resident)
    xs = {}
    for name, env in (("wave", {"FEM_AMD_LU_WAVE": "1", "FEM_AMD_LU_BCR": "0"}), ("walk", {"FEM_AMD_LU_WAVE": "0", "FEM_AMD_LU_BCR": "0"}),
                      ("bcr", {"FEM_AMD_LU_WAVE": "1", "FEM_AMD_LU_BCR": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        prob = driver.Problem(gpu_ctx, m, LU)
        prob.assemble()
        rp, ci, v = prob.K.get_csr()
        r0 = m.ndof // 2
        cols = ci[rp[r0]:rp[r0 + 1]]
        prob.K.set_values([r0], [int(cols[cols != r0][0])], [1e-3 * np.abs(v).max()], "add")     # (flags the Mat unsymmetric: LU, not Cholesky)
        rp, ci, v = prob.K.get_csr()
        prob.ksp.set_operators(prob.K); prob.ksp.set_up()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        b = prob.F.get_array()
        prob.ksp.solve(prob.F, prob.U)
        xs[name] = prob.U.get_array()
        log = capfd.readouterr().err
        if name == "bcr":
            assert "a small pivot on the way: band LU" in log, log
        A = sp.csr_matrix((v, ci, rp))
        res = np.abs(A @ xs[name] - b)
        # all equations but those of the (at most three, plus the perturbed row's neighbourhood) dropped unknowns hold
        assert np.sort(res)[:-6].max() <= 1e-6 * np.abs(b).max(), np.sort(res)[-8:]
        assert np.isfinite(xs[name]).all()
        prob.destroy()
    scale = np.abs(xs["walk"]).max()
    assert np.abs(xs["wave"] - xs["walk"]).max() <= 1e-6 * scale
    assert np.abs(xs["bcr"] - xs["walk"]).max() <= 1e-6 * scale


@pytest.mark.parametrize("resident", ["pcr", "cr", "0"])
def test_a_solve_that_fails_the_backward_error_test_goes_to_the_band_lu(gpu_ctx, monkeypatch, capfd, resident):
    """The safety net of block cyclic reduction: a solve whose componentwise backward error exceeds the bound is repeated by the band
    LU (factorised then), and so is every later solve with that matrix.  The bound is set to 0 here (FEM_AMD_LU_BCR_TOL), so that
    the first solve takes that way: same answer as with the band LU from the start, and a second right-hand side is solved too."""
    m = structured_quad(300, 2, jitter=0.1)
    monkeypatch.setenv("FEM_AMD_DIRECT_VERBOSE", "1")
    _bcr_form(monkeypatch, resident)
    xs = {}
    for name, env in (("fallback", {"FEM_AMD_LU_BCR": "1", "FEM_AMD_LU_BCR_TOL": "0"}), ("band", {"FEM_AMD_LU_BCR": "0", "FEM_AMD_LU_BCR_TOL": "1e-10"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        prob = driver.Problem(gpu_ctx, m, LU)
        prob.assemble()
        rp, ci, v = prob.K.get_csr()
        r0 = m.ndof // 2
        cols = ci[rp[r0]:rp[r0 + 1]]
        prob.K.set_values([r0], [int(cols[cols != r0][0])], [1e-3 * np.abs(v[v < 1e29]).max()], "add")      # (unsymmetric: LU)
        prob.ksp.set_operators(prob.K); prob.ksp.set_up()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        prob.ksp.solve(prob.F, prob.U)
        x1 = prob.U.get_array()
        prob.F.set_array(np.cos(0.01 * np.arange(m.ndof)) * np.abs(prob.F.get_array()).max())
        prob.ksp.solve(prob.F, prob.U)
        xs[name] = (x1, prob.U.get_array())
        log = capfd.readouterr().err
        if name == "fallback":
            # the second solve no longer tries the reduction (the parallel form hands over to the form with a way up first)
            assert "used" in log and log.count("band LU from here on") == 1 and log.count("parallel form:") == (1 if resident == "pcr" else 0), log
        prob.destroy()
    for a, b in zip(xs["fallback"], xs["band"]):
        assert np.array_equal(a, b)
