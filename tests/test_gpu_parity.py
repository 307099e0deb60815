"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on
the same seeded inputs.  Tolerances: element level 1e-13 relative to the
largest entry; assembled matrix 1e-13; solve 1e-10 relative (north_star)."""
import numpy as np
import pytest

from fem_amd import capi, driver
from fem_amd.mesh import TRI, QUA, TET, HEX, structured_hex, structured_quad, hex_to_tets, quad_to_mixed

pytestmark = pytest.mark.gpu

BASE = {
    TRI: np.array([[0, 0], [1, 0], [0, 1.0]]),
    QUA: np.array([[0, 0], [1, 0], [1, 1], [0, 1.0]]),
    TET: np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]),
    HEX: np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1.0]]),
}


def _rand_elems(t, nel, seed):
    rng = np.random.default_rng(seed)
    b = BASE[t]
    ec = b[None] * rng.uniform(0.5, 2.0, (nel, 1, 1)) + rng.uniform(-0.2, 0.2, (nel,) + b.shape)
    E = rng.uniform(1e3, 1e11, nel)
    nu = rng.uniform(0.0, 0.45, nel)
    return ec, E, nu


@pytest.mark.parametrize("t", [TRI, QUA, TET, HEX])
def test_form_el_ke_matches_oracle(gpu_ctx, oracle, t):
    nel = 203  # ragged: not a multiple of the workgroup's element count
    ec, E, nu = _rand_elems(t, nel, 100 + t)
    k = gpu_ctx.form_el_ke(t, ec, E, nu)
    for e in range(nel):
        kr = oracle.form_el_ke(t, ec[e], E[e], nu[e])
        assert np.abs(k[e] - kr).max() <= 1e-13 * np.abs(kr).max()


@pytest.mark.parametrize("t", [TRI, QUA, TET, HEX])
def test_calc_el_stress_matches_oracle(gpu_ctx, oracle, t):
    nel = 131
    ec, E, nu = _rand_elems(t, nel, 200 + t)
    rng = np.random.default_rng(7)
    u = rng.standard_normal((nel, ec.shape[1] * ec.shape[2]))
    s = gpu_ctx.calc_el_stress(t, ec, u, E, nu)
    for e in range(nel):
        sr = oracle.calc_el_stress(t, ec[e], u[e], E[e], nu[e])
        assert np.abs(s[e] - sr).max() <= 1e-13 * np.abs(sr).max()


def test_empty_batch(gpu_ctx):
    k = gpu_ctx.form_el_ke(HEX, np.zeros((0, 8, 3)), np.zeros(0), np.zeros(0))
    assert k.shape == (0, 24, 24)


def _meshes():
    return {
        "hex5_jitter": structured_hex(5, 4, 3, jitter=0.2),
        "hex_67nodes_x": structured_hex(66, 2, 2, jitter=0.1),   # rows cross several 64-node slices
        "quad9_jitter": structured_quad(9, 7, jitter=0.2),
        "tet": hex_to_tets(structured_hex(4, 3, 3, jitter=0.15)),
        "mixed_tri_qua": quad_to_mixed(structured_quad(8, 5, jitter=0.1)),
        # every element the same element, bit for bit (exact coordinates, one material): from the third assembly on such a mesh the
        # blocks come from a table of the first element's (elem_kernels.hip, AsmUniform); the second one compares as it goes
        "hex_uniform": structured_hex(9, 7, 5),
        "quad_uniform": structured_quad(21, 13),
        "hex_one_element_stiffer": _one_element_stiffer(structured_hex(7, 5, 4)),   # ... and must not where a single element differs
    }


def _one_element_stiffer(m):
    m.mats = np.vstack([m.mats, m.mats[0] * np.array([1.5, 1.0, 1.0, 1.0, 1.0])])   # a second material, E x 1.5
    m.mat_id = m.mat_id.copy()
    m.mat_id[m.nels // 2] = 1
    return m


@pytest.mark.parametrize("name", list(_meshes()))
def test_assembled_matrix_matches_oracle(gpu_ctx, oracle, name):
    m = _meshes()[name]
    rp, ci = oracle.csr_pattern(m)
    vr = oracle.assemble(m, rp, ci, apply_bc=True)
    dm = capi.DeviceMesh(gpu_ctx, m)
    K = capi.Mat(dm)
    K.assemble_elements()
    for _ in range(2):                           # (the third assembly on a mesh: from the table where the second found it uniform)
        K.zero_entries(); K.assemble_elements()
    K.apply_kbc()
    grp, gci, gv = K.get_csr()
    assert np.array_equal(grp, rp) and np.array_equal(gci, ci)
    pen = vr > 1e29
    assert np.array_equal(gv[pen], vr[pen])                       # INSERTed penalty is bit-exact
    scale = np.abs(vr[~pen]).max()
    assert np.abs(gv[~pen] - vr[~pen]).max() <= 1e-13 * scale
    # MatMult against the oracle SpMV
    rng = np.random.default_rng(3)
    x = rng.standard_normal(m.ndof)
    X, Y = capi.Vec(gpu_ctx, m.ndof), capi.Vec(gpu_ctx, m.ndof)
    X.set_array(x)
    K.mult(X, Y)
    yr = oracle.spmv(rp, ci, vr, x)
    assert np.abs(Y.get_array() - yr).max() <= 1e-12 * np.abs(yr).max()
    for o in (X, Y, K, dm):
        o.destroy()


@pytest.mark.parametrize("name", list(_meshes()))
def test_assembly_is_bitwise_reproducible_and_adds(gpu_ctx, oracle, name):
    """The row-owner assembly has one writer per matrix entry and a fixed summation order: two assemblies of one
    mesh are bitwise equal (also across two Mats), MatZeroEntries really zeroes (it is lazy inside), and a second
    element loop without MatZeroEntries ADDs (MatSetValues(ADD_VALUES), src/main.F90:162)."""
    m = _meshes()[name]
    dm = capi.DeviceMesh(gpu_ctx, m)
    K, K2 = capi.Mat(dm), capi.Mat(dm)
    assert np.all(K.get_csr()[2] == 0.0)                          # a new Mat is zero
    K.assemble_elements()
    v1 = K.get_csr()[2]
    K2.assemble_elements()
    assert np.array_equal(K2.get_csr()[2], v1)
    K.zero_entries(); K.assemble_elements()
    assert np.array_equal(K.get_csr()[2], v1)
    K.assemble_elements()                                          # no zeroing in between: the values add up
    assert np.array_equal(K.get_csr()[2], v1 + v1)
    K.zero_entries()
    assert np.all(K.get_csr()[2] == 0.0)
    rp, ci = oracle.csr_pattern(m)
    vr = oracle.assemble(m, rp, ci, apply_bc=False)
    assert np.abs(v1 - vr).max() <= 1e-13 * np.abs(vr).max()
    for o in (K, K2, dm):
        o.destroy()


def test_assembly_through_a_narrow_accumulator_window(oracle):
    """A slice wider than the LDS image is assembled window by window; forced here by FEM_AMD_ASM_WCAP in a child
    process (the cap is read once per process)."""
    import os
    import subprocess
    import sys
    code = """
import numpy as np
from fem_amd import capi
from fem_amd.mesh import structured_hex, hex_to_tets
from oracle.oracle import Oracle
o = Oracle(); ctx = capi.Context(0)
for m in (structured_hex(5, 4, 3, jitter=0.2), hex_to_tets(structured_hex(4, 3, 3, jitter=0.15))):
    rp, ci = o.csr_pattern(m); vr = o.assemble(m, rp, ci, apply_bc=False)
    dm = capi.DeviceMesh(ctx, m); K = capi.Mat(dm); K.assemble_elements()
    g = K.get_csr()
    assert np.array_equal(g[0], rp) and np.array_equal(g[1], ci)
    assert np.abs(g[2] - vr).max() <= 1e-13 * np.abs(vr).max()
    K.destroy(); dm.destroy()
ctx.destroy(); print('WINDOWED-OK')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEM_AMD_ASM_WCAP="5", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert "WINDOWED-OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("name", list(_meshes()))
def test_rhs_matches_oracle(gpu_ctx, oracle, name):
    m = _meshes()[name]
    dm = capi.DeviceMesh(gpu_ctx, m)
    F = capi.Vec(gpu_ctx, m.ndof)
    dm.form_rhs(F, 0.0, m.dt)
    fr = oracle.form_rhs(m, 0.0, m.dt)
    assert np.array_equal(F.get_array(), fr)
    F.destroy(); dm.destroy()


@pytest.mark.parametrize("name", list(_meshes()))
def test_solve_and_stress_match_oracle(gpu_ctx, oracle, name):
    m = _meshes()[name]
    res = driver.run(m, "-ksp_rtol 1e-13", ctx=gpu_ctx)
    u, s, its = oracle.run(m, "cg", rtol=1e-13)
    assert res.reasons[-1] > 0
    assert np.abs(res.aggregate_u - u).max() <= 1e-10 * np.abs(u).max()
    assert np.abs(res.aggregate_stress - s).max() <= 1e-10 * np.abs(s).max()
    # same algorithm, same semantics: iteration counts agree up to rounding noise
    assert abs(res.its[-1] - its[-1]) <= max(2, its[-1] // 50)


@pytest.mark.parametrize("n,jitter", [(40, 0.0), (56, 0.2)])
def test_medium_cube_solution_matches_oracle(gpu_ctx, oracle, n, jitter):
    """Config-3 setup at a size the oracle still solves in seconds (56^3 = 555k DOF): same CG
    semantics on both sides, rtol 1e-13 -> displacements within 1e-10, nodal stress within 1e-10."""
    m = structured_hex(n, jitter=jitter)
    res = driver.run(m, "-ksp_rtol 1e-13", ctx=gpu_ctx)
    u, s, its = oracle.run(m, "cg", rtol=1e-13, nthreads=8)
    assert res.reasons[-1] == 2
    assert np.abs(res.aggregate_u - u).max() <= 1e-10 * np.abs(u).max()
    assert np.abs(res.aggregate_stress - s).max() <= 1e-10 * np.abs(s).max()
    assert abs(res.its[-1] - its[-1]) <= max(3, its[-1] // 50)


@pytest.mark.parametrize("opts", ["", "-mat_mult_type ebe", "-mat_mult_type ebe_gather"])
@pytest.mark.parametrize("n,jitter", [((9, 8, 7), 0.0), ((7, 6, 5), 0.2)])
def test_streaming_cg_is_bitwise_the_default_cg(gpu_ctx, opts, n, jitter):
    """Problems larger than the Infinity Cache run the CG with the once-per-iteration vectors (x, K p) moved with the streaming
    hint (ksp.hip, -ksp_stream_min_dof / FEM_AMD_STREAM_MIN_DOF): forced on a small mesh, the same iterations and fields to the bit."""
    m = structured_hex(*n, jitter=jitter)
    a = driver.run(m, f"-ksp_rtol 1e-12 {opts}", ctx=gpu_ctx)
    b = driver.run(m, f"-ksp_rtol 1e-12 {opts} -ksp_stream_min_dof 0", ctx=gpu_ctx)
    assert a.reasons == b.reasons and a.reasons[-1] > 0 and a.its == b.its
    assert np.array_equal(a.aggregate_u, b.aggregate_u)
    assert np.array_equal(a.aggregate_stress, b.aggregate_stress)


@pytest.mark.parametrize("n,jitter", [((9, 7), 0.2), ((40, 33), 0.0), ((70, 45), 0.15)])
def test_cg_opened_by_the_quad4_product_is_bitwise_the_three_kernel_cg(gpu_ctx, n, jitter):
    """Quad4 meshes that sit in the caches: the tile product opens the CG iteration itself (convergence test, p = M^-1 r + b p in its
    stage-in, the owed x += a p; cg_device.h, -ksp_cg_fuse_p).  Same iterations, residual norm and fields to the bit as the CG with
    k_cg_update_p (-ksp_cg_fuse_p 0), and both agree with the assembled product to round-off."""
    from fem_amd.mesh import structured_quad
    m = structured_quad(*n, jitter=jitter)
    a = driver.run(m, "-ksp_rtol 1e-12 -mat_mult_type ebe -ksp_cg_fuse_p 0", ctx=gpu_ctx)
    b = driver.run(m, "-ksp_rtol 1e-12 -mat_mult_type ebe -ksp_cg_fuse_p 1", ctx=gpu_ctx)
    c = driver.run(m, "-ksp_rtol 1e-12", ctx=gpu_ctx)
    assert a.reasons == b.reasons and a.reasons[-1] > 0 and a.its == b.its
    assert np.array_equal(a.aggregate_u, b.aggregate_u)
    assert np.array_equal(a.aggregate_stress, b.aggregate_stress)
    assert np.abs(b.aggregate_u - c.aggregate_u).max() <= 1e-9 * np.abs(c.aggregate_u).max()


@pytest.mark.parametrize("opts", ["", "-mat_mult_type ebe", "-pc_type mg -pc_mg_type additive", "-ksp_type bcgs"])
def test_repeated_right_hand_side_returns_the_kept_solution(gpu_ctx, opts):
    """-ksp_reuse_solution (default on): the reference's increment loop solves K du = F once per increment (src/main.F90:236-250),
    with the same F every time on a linear deck with ramped loads.  A solve that repeats the previous one -- same operator state,
    same options, right-hand side equal bit for bit -- returns the kept solution with the kept iteration count; a changed
    right-hand side, a changed operator (ApplyKBC again, MatSetValues) or new options make it iterate again.  Same fields to the
    bit as with the shortcut off."""
    m = structured_hex(7, 6, 5, jitter=0.15)
    m.t, m.dt = 1.0, 0.25                                    # four increments, linear ramp: four identical right-hand sides
    a = driver.run(m, f"-ksp_rtol 1e-12 {opts} -ksp_reuse_solution 0", ctx=gpu_ctx)
    b = driver.run(m, f"-ksp_rtol 1e-12 {opts}", ctx=gpu_ctx)
    assert len(a.its) == 4 and a.its == b.its and a.reasons == b.reasons and a.reasons[-1] > 0
    assert np.array_equal(a.aggregate_u, b.aggregate_u) and np.array_equal(a.aggregate_stress, b.aggregate_stress)
    prob = driver.Problem(gpu_ctx, m, f"-ksp_rtol 1e-12 {opts}")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    U2 = capi.Vec(gpu_ctx, m.ndof)
    prob.ksp.solve(prob.F, prob.U)
    its = prob.ksp.get_iteration_number()
    assert prob.ksp.get_reused_solves() == 0
    prob.ksp.solve(prob.F, U2)                                # the same again: kept
    assert prob.ksp.get_reused_solves() == 1 and prob.ksp.get_iteration_number() == its
    assert np.array_equal(prob.U.get_array(), U2.get_array())
    f = prob.F.get_array(); j = int(np.argmax(np.abs(f))); f[j] = np.nextafter(f[j], 0.0); prob.F.set_array(f)
    prob.ksp.solve(prob.F, U2)                                # one bit of F differs: a real solve
    assert prob.ksp.get_reused_solves() == 1
    prob.ksp.solve(prob.F, U2)
    assert prob.ksp.get_reused_solves() == 2
    prob.K.apply_kbc()                                        # the operator's state moved (same values): a real solve
    prob.ksp.solve(prob.F, U2)
    assert prob.ksp.get_reused_solves() == 2
    prob.ksp.set_from_options("-ksp_rtol 1e-10")              # new options: a real solve
    prob.ksp.solve(prob.F, U2)
    assert prob.ksp.get_reused_solves() == 2
    prob.ksp.solve(prob.F, U2)
    assert prob.ksp.get_reused_solves() == 3
    U2.destroy(); prob.destroy()


@pytest.mark.parametrize("opts", ["", "-mat_mult_type ebe"])
def test_tightened_tolerances_are_honoured_after_a_kept_solve(gpu_ctx, oracle, opts):
    """KSPSetTolerances between two solves of one right-hand side (the reference's own call, src/main.F90:349-351, then KSPSolve :247):
    the second solve runs under the new tolerances -- it never hands out the solution kept by -ksp_reuse_solution (default on) from
    the looser solve.  The same for KSPSetType / PCSetType / KSPSetOperators."""
    m = structured_hex(9, 8, 7, jitter=0.15)
    uo, _, _ = oracle.run(m, "cg", rtol=1e-13)
    prob = driver.Problem(gpu_ctx, m, f"-ksp_rtol 1e-9 {opts}")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U)
    its9, u9 = prob.ksp.get_iteration_number(), prob.U.get_array()
    prob.ksp.set_tolerances(rtol=1e-13)
    prob.ksp.solve(prob.F, prob.U)
    its13, u13 = prob.ksp.get_iteration_number(), prob.U.get_array()
    assert prob.ksp.get_reused_solves() == 0 and prob.ksp.get_converged_reason() > 0
    assert its13 > its9 and not np.array_equal(u9, u13)
    assert np.abs(u13 - uo).max() <= 1e-10 * np.abs(uo).max()
    prob.ksp.solve(prob.F, prob.U)                            # unchanged options: kept
    assert prob.ksp.get_reused_solves() == 1 and prob.ksp.get_iteration_number() == its13
    prob.ksp.set_tolerances(max_it=its13 + 50)                # every field of KSPSetTolerances counts
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_reused_solves() == 1
    prob.ksp.set_tolerances(abstol=1e-40)
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_reused_solves() == 1
    prob.ksp.set_pc_type("jacobi"); prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_reused_solves() == 1 and prob.ksp.get_iteration_number() == its13
    prob.ksp.set_type("cg"); prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_reused_solves() == 1
    prob.ksp.set_operators(prob.K); prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_reused_solves() == 1
    assert np.array_equal(prob.U.get_array(), u13)            # the deterministic solver: the same bits every time
    prob.destroy()


def test_a_changing_operator_keeps_nothing(gpu_ctx):
    """A host whose operator changes between solves (Newton / cohesive increments, src/m_global.F90:98-158) never repeats a solve:
    after the first change of state nothing is kept, until two solves in a row see one state again."""
    m = structured_hex(6, 5, 4, jitter=0.1)
    prob = driver.Problem(gpu_ctx, m, "-ksp_rtol 1e-12")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U); u0 = prob.U.get_array()
    for _ in range(3):
        prob.K.apply_kbc()                                    # state moves (same values)
        prob.ksp.solve(prob.F, prob.U)
        assert prob.ksp.get_reused_solves() == 0 and np.array_equal(prob.U.get_array(), u0)
    prob.ksp.solve(prob.F, prob.U)                            # first solve on an unchanged state: computed, and kept again
    assert prob.ksp.get_reused_solves() == 0
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_reused_solves() == 1 and np.array_equal(prob.U.get_array(), u0)
    prob.destroy()


@pytest.mark.parametrize("n,jitter", [((9, 7), 0.2), ((40, 33), 0.0), ((70, 45), 0.15), ((150, 120), 0.1)])
def test_cg_as_one_resident_launch_matches_the_kernel_per_iteration_cg(gpu_ctx, oracle, n, jitter):
    """Quad4 meshes whose tiles are all resident at once (config 2's class): KSPCG / PCJACOBI runs as ONE launch, a workgroup per tile
    for the whole solve and two XCD-hierarchical grid barriers per iteration (k_cg_resident_q4, -ksp_cg_resident, default on).  Same
    KSPSolve semantics (src/main.F90:247): iteration count, converged reason and residual history of the launch-per-kernel path up to
    round-off (z.z and z.r are summed tile by tile instead of in index order), displacements within 1e-10 of the oracle."""
    from fem_amd.mesh import structured_quad
    m = structured_quad(*n, jitter=jitter)
    uo, so, ito = oracle.run(m, "cg", rtol=1e-13)
    got = {}
    for res in (0, 1):
        prob = driver.Problem(gpu_ctx, m, f"-ksp_rtol 1e-13 -mat_mult_type ebe -ksp_cg_resident {res}")
        prob.assemble(); prob.ksp.set_up()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        prob.ksp.solve(prob.F, prob.U)
        assert prob.ksp.get_resident_solves() == res
        got[res] = (prob.U.get_array(), prob.ksp.get_iteration_number(), prob.ksp.get_converged_reason(), prob.ksp.get_residual_history())
        prob.ksp.solve(prob.F, prob.U)                        # (-ksp_reuse_solution: the kept one, whichever path computed it)
        assert prob.ksp.get_reused_solves() == 1 and np.array_equal(prob.U.get_array(), got[res][0])
        prob.destroy()
    (ul, il, rl, hl), (ur, ir, rr, hr) = got[0], got[1]
    assert rl == rr == 2 and abs(il - ir) <= max(2, il // 100) and abs(ir - ito[-1]) <= max(2, ir // 50)
    k = min(len(hl), len(hr)) // 2
    assert np.allclose(hl[:10], hr[:10], rtol=1e-9, atol=0.0)       # the same residual history while round-off has not spread ...
    assert np.all(np.abs(np.log(hl[:k] / hr[:k])) < 0.2)           # ... and the same curve once it has (CG on a penalty matrix amplifies it)
    assert np.abs(ur - ul).max() <= 1e-11 * np.abs(uo).max()
    assert np.abs(ur - uo).max() <= 1e-10 * np.abs(uo).max()


def test_resident_cg_is_bitwise_reproducible_and_honours_the_ksp_surface(gpu_ctx):
    """Two resident solves give the same bits; -ksp_max_it ends it with KSP_DIVERGED_ITS and the iterate of that iteration; a zero
    right-hand side converges at iteration 0; a penalty deck whose tile holds more constrained dofs than the kernel's LDS list
    (every node clamped) takes the list-free branch."""
    from fem_amd.mesh import structured_quad
    m = structured_quad(60, 40, jitter=0.1)
    prob = driver.Problem(gpu_ctx, m, "-ksp_rtol 1e-12 -mat_mult_type ebe -ksp_reuse_solution 0")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U); u1, it1 = prob.U.get_array(), prob.ksp.get_iteration_number()
    prob.ksp.solve(prob.F, prob.U); u2 = prob.U.get_array()
    assert prob.ksp.get_resident_solves() == 2 and np.array_equal(u1, u2)
    prob.ksp.set_tolerances(max_it=it1 // 2)
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_converged_reason() == -3 and prob.ksp.get_iteration_number() == it1 // 2
    um = prob.U.get_array()
    ref = driver.Problem(gpu_ctx, m, f"-ksp_rtol 1e-12 -mat_mult_type ebe -ksp_cg_resident 0 -ksp_max_it {it1 // 2}")
    ref.assemble(); ref.ksp.set_up(); ref.ksp.solve(prob.F, ref.U)
    assert ref.ksp.get_converged_reason() == -3 and np.abs(ref.U.get_array() - um).max() <= 1e-5 * np.abs(um).max()   # (mid-way iterates: round-off amplified by CG)
    ref.destroy()
    prob.ksp.set_tolerances(max_it=10000)
    prob.F.set(0.0)
    prob.ksp.solve(prob.F, prob.U)
    assert prob.ksp.get_iteration_number() == 0 and prob.ksp.get_converged_reason() > 0 and not prob.U.get_array().any()
    prob.destroy()
    m.bcnode = np.column_stack([np.arange(m.nnds), np.zeros(m.nnds), np.zeros(m.nnds)]).astype(np.int32)   # every dof constrained
    m.bcval = np.tile(np.array([[0.01, -0.02]]), (m.nnds, 1))
    for res in (0, 1):
        p2 = driver.Problem(gpu_ctx, m, f"-ksp_rtol 1e-12 -mat_mult_type ebe -ksp_cg_resident {res}")
        p2.assemble(); p2.ksp.set_up()
        p2.dmesh.form_rhs(p2.F, 0.0, m.dt)
        p2.ksp.solve(p2.F, p2.U)
        assert p2.ksp.get_converged_reason() > 0 and p2.ksp.get_resident_solves() == res
        if res == 0: ua = p2.U.get_array()
        else: assert np.abs(p2.U.get_array() - ua).max() <= 1e-12 * np.abs(ua).max()
        p2.destroy()


def test_resident_cg_whose_barrier_gives_up_is_answered_by_the_launch_path():
    """A resident launch needs every workgroup resident and present: if one never arrives (test hook: workgroup 1 leaves before the
    census) the bounded spins of the grid barrier end the launch, the KSP takes the kernels-per-iteration path for this and every
    later solve, and the answer is the launch path's.  (A child process: the hook is read from the environment.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import numpy as np
from fem_amd import capi, driver
from fem_amd.mesh import structured_quad
ctx = capi.Context(0)
m = structured_quad(70, 45, jitter=0.1)
out = {}
for res in (0, 1):
    p = driver.Problem(ctx, m, f"-ksp_rtol 1e-12 -mat_mult_type ebe -ksp_cg_resident {res} -ksp_reuse_solution 0")
    p.assemble(); p.ksp.set_up()
    p.dmesh.form_rhs(p.F, 0.0, m.dt)
    p.ksp.solve(p.F, p.U)
    out[res] = (p.U.get_array(), p.ksp.get_iteration_number(), p.ksp.get_converged_reason(), p.ksp.get_resident_solves())
    if res:
        p.ksp.solve(p.F, p.U)     # the KSP does not try again
        assert p.ksp.get_resident_solves() == 0 and np.array_equal(p.U.get_array(), out[1][0])
    p.destroy()
assert out[1][3] == 0 and out[0][2] == out[1][2] == 2 and out[0][1] == out[1][1]
assert np.array_equal(out[0][0], out[1][0])
print("FALLBACK_OK")
'''
    env = dict(os.environ, FEM_AMD_CG_RESIDENT_TEST_LEAVE="1", PYTHONPATH=root)
    pr = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert pr.returncode == 0 and "FALLBACK_OK" in pr.stdout, pr.stdout[-2000:] + pr.stderr[-2000:]
