"""GPU: `-pc_type mg` (box-aggregation multigrid) behind the same KSP surface.
A preconditioner changes the iteration, not the answer: solutions are checked against the
oracle's CG/Jacobi solution of the same system, and the iteration count must drop."""
import numpy as np
import pytest

from fem_amd import capi, driver
from fem_amd.mesh import structured_hex, structured_quad, hex_to_tets, quad_to_mixed

pytestmark = pytest.mark.gpu


def _cases():
    return {
        "hex12_jitter": structured_hex(12, 10, 9, jitter=0.2),
        "hex_tiny": structured_hex(3, 2, 2),                      # a single box level (coarsest only)
        "quad40_jitter": structured_quad(40, 33, jitter=0.2),
        "tet": hex_to_tets(structured_hex(7, 6, 5, jitter=0.15)),
        "mixed_tri_qua": quad_to_mixed(structured_quad(20, 14, jitter=0.1)),
    }


@pytest.mark.parametrize("mgtype", ["multiplicative", "additive"])
@pytest.mark.parametrize("name", list(_cases()))
def test_mg_solution_matches_oracle(gpu_ctx, oracle, name, mgtype):
    m = _cases()[name]
    res = driver.run(m, f"-pc_type mg -pc_mg_type {mgtype} -ksp_rtol 1e-13", ctx=gpu_ctx)
    jac = driver.run(m, "-pc_type jacobi -ksp_rtol 1e-13", ctx=gpu_ctx, recover_stress=False)
    u, s, its = oracle.run(m, "cg", rtol=1e-13)
    assert res.reasons[-1] == 2
    assert np.abs(res.aggregate_u - u).max() <= 1e-10 * np.abs(u).max()
    assert np.abs(res.aggregate_stress - s).max() <= 1e-10 * np.abs(s).max()
    if mgtype == "multiplicative":
        assert res.its[-1] <= jac.its[-1]
    if name in ("hex12_jitter", "quad40_jitter") and mgtype == "multiplicative":
        assert res.its[-1] * 2 < jac.its[-1]


@pytest.mark.parametrize("mgtype", ["multiplicative", "additive"])
def test_mg_transfer_kernels_with_coordinates_or_kept_offsets(gpu_ctx, monkeypatch, mgtype):
    """The cycle's fine-level transfer kernels read the nodes' offsets from their box centres in single precision (kept with the
    aggregates); FEM_AMD_MG_FLOAT_OFFSETS=0 makes them read the coordinates as set-up does: the same preconditioner to 6e-8 --
    iteration counts within one of each other, the same field."""
    out = {}
    for m in (structured_hex(12, 10, 9, jitter=0.2), structured_quad(40, 33, jitter=0.2)):
        for fo in ("1", "0"):
            monkeypatch.setenv("FEM_AMD_MG_FLOAT_OFFSETS", fo)
            res = driver.run(m, f"-pc_type mg -pc_mg_type {mgtype} -ksp_rtol 1e-13", ctx=gpu_ctx, recover_stress=False)
            assert res.reasons[-1] == 2
            out[fo] = (res.aggregate_u, res.its[-1])
        assert abs(out["1"][1] - out["0"][1]) <= 1
        assert np.abs(out["1"][0] - out["0"][0]).max() <= 1e-10 * np.abs(out["0"][0]).max()


def test_mg_is_deterministic_and_gamg_alias(gpu_ctx):
    m = structured_hex(10, jitter=0.2)
    prob = driver.Problem(gpu_ctx, m, "-pc_type gamg -ksp_rtol 1e-12")
    prob.assemble(); prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U); u1 = prob.U.get_array(); i1 = prob.ksp.get_iteration_number()
    prob.ksp.solve(prob.F, prob.U); u2 = prob.U.get_array(); i2 = prob.ksp.get_iteration_number()
    assert i1 == i2 and np.array_equal(u1, u2)
    prob.destroy()


def test_mg_full_size_config3(gpu_ctx):
    """Config 3 at full size with -pc_type mg: converges to the reference rtol in far fewer
    iterations and to the same displacement field as CG/Jacobi (both at rtol 1e-12)."""
    m = structured_hex(149)
    out = {}
    # (-pc_mg_stencil_f32 0: the cycle reads the level-1 stencil in doubles instead of its single-precision copy -- the same
    # preconditioner to 6e-8: iteration counts within one of each other, the same field)
    for pc in ("jacobi", "mg", "mg -pc_mg_type additive", "mg -pc_mg_type additive -pc_mg_stencil_f32 0"):
        prob = driver.Problem(gpu_ctx, m, f"-pc_type {pc} -ksp_rtol 1e-12")
        prob.assemble(); prob.ksp.set_up()
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        prob.ksp.solve(prob.F, prob.U)
        assert prob.ksp.get_converged_reason() == 2
        out[pc] = (prob.U.get_array(), prob.ksp.get_iteration_number())
        prob.destroy()
    assert out["mg"][1] * 4 < out["jacobi"][1] and out["mg -pc_mg_type additive"][1] * 2 < out["jacobi"][1]
    for k in ("mg", "mg -pc_mg_type additive", "mg -pc_mg_type additive -pc_mg_stencil_f32 0"):
        assert np.abs(out[k][0] - out["jacobi"][0]).max() <= 1e-9 * np.abs(out["jacobi"][0]).max()
    assert abs(out["mg -pc_mg_type additive"][1] - out["mg -pc_mg_type additive -pc_mg_stencil_f32 0"][1]) <= 1


@pytest.mark.parametrize("mgtype", ["additive", "multiplicative"])
@pytest.mark.parametrize("world,npr", [(2, (9, 8, 7)), (4, (6, 6, 9)), (8, (9, 10, 11)), (6, (14, 9, 12)), (3, (23, 8, 8))])
def test_mg_on_a_partitioned_mesh(oracle, world, npr, mgtype):
    """Partitioned mesh (brick domains, in-process communicator on one GPU): the preconditioner is the
    same operator as on the undivided mesh -- one box grid over the global bounding box, every domain holding its sub-grid and
    its part of the Galerkin operators, shared boxes summed by neighbour exchanges, only the coarsest level replicated -- so
    the solution matches the serial oracle, the iteration count matches the single-domain multigrid run, interface copies
    stay bitwise identical, and no collective grows with the mesh."""
    from tests.test_gpu_multidomain import _run_domains, assert_no_collective_grows_with_the_mesh
    from fem_amd import partition
    opts = f"-pc_type mg -pc_mg_type {mgtype} -ksp_rtol 1e-12"
    grid, out = _run_domains(world, npr, 0.15, opts)
    assert_no_collective_grows_with_the_mesh()
    gn = tuple(grid[d] * npr[d] for d in range(3))
    g = structured_hex(*gn, jitter=0.15)
    ug, sg, _ = oracle.run(g, "cg", rtol=1e-13)
    ctx = capi.Context(0)
    single = driver.run(g, opts, ctx=ctx, recover_stress=False)
    ctx.destroy()
    for r, (u, s, its, reason, y, gid) in enumerate(out):
        assert reason == 2
        assert np.abs(u.reshape(-1, 3) - ug.reshape(-1, 3)[gid]).max() <= 1e-9 * np.abs(ug).max()
        assert np.abs(s - sg[gid]).max() <= 1e-8 * np.abs(sg).max()
        assert abs(its - single.its[-1]) <= 2
    for r in range(world):
        for q in range(r + 1, world):
            common, ir, iq = np.intersect1d(out[r][5], out[q][5], return_indices=True)
            if common.size:
                assert np.array_equal(out[r][0].reshape(-1, 3)[ir], out[q][0].reshape(-1, 3)[iq])


def test_mg_coarsens_its_boxes_on_a_graded_mesh(gpu_ctx, oracle, monkeypatch, capfd):
    """Geometric grading 1:1000 along x: the largest element is wider than the default box (4 mean node
    spacings), so it would span three boxes; set-up must coarsen the grid by itself and still converge
    to the oracle's solution."""
    m = structured_hex(24, 6, 6)
    x = m.coords[:, 0] / m.coords[:, 0].max()
    m.coords[:, 0] = 24.0 * (np.exp(7.0 * x) - 1.0) / (np.exp(7.0) - 1.0)
    monkeypatch.setenv("FEM_AMD_MG_VERBOSE", "1")
    res = driver.run(m, "-pc_type mg -pc_mg_type additive -ksp_rtol 1e-12", ctx=gpu_ctx)
    log = capfd.readouterr().err
    H = float(log.split("level 1:")[1].split("H = ")[1].split(",")[0])
    hnode = (24.0 * 6 * 6 / m.nnds) ** (1 / 3)
    assert H >= 1.4 * 4 * hnode                       # at least one coarsening step was taken
    u, s, _ = oracle.run(m, "cg", rtol=1e-13)
    assert res.reasons[-1] == 2
    assert np.abs(res.aggregate_u - u).max() <= 1e-9 * np.abs(u).max()
