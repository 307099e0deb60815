"""GPU: the Newton increment around the cohesive element (BASELINE config 5): the device primitives of
the two SNES callbacks against the oracle's CalcJacobian / CalcResidual, the Newton line-search solver against the
oracle's restatement of SNESNEWTONLS, on examples/cohesive_test_2.inp and on its tiling."""
import os

import numpy as np
import pytest

from fem_amd import capi, driver
from fem_amd.inp import read_inp
from fem_amd.mesh import tile_deck

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DECK = os.path.join(HERE, "golden", "decks", "cohesive_test_2.inp")
LEGACY = os.path.join(HERE, "golden", "decks", "cohesive_test_legacy.inp")


@pytest.mark.parametrize("rels", [(0, None), (0, (1, 2, 4, 5)), (1, None)])
def test_callbacks_match_oracle(gpu_ctx, oracle, rels):
    """CalcJacobian (MatCopy + applyStiff_1 + 64 MatSetValue(ADD) + ApplyKBC) and CalcResidual (VecCopy + applyTract_1 +
    EnforceBCForce + VecScale + MatMultAdd) on a non-trivial state, both readings of getCohRels' node list."""
    mode, nodes = rels
    m = read_inp(DECK)
    prob = driver.NonlinearProblem(gpu_ctx, m, "", rels_mode=mode, rels_nodes=nodes)
    rng = np.random.default_rng(4)
    agg = 1e-3 * rng.standard_normal(m.ndof)
    du = 1e-3 * rng.standard_normal(m.ndof)
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.agg.set_array(agg)
    X = capi.Vec(gpu_ctx, m.ndof); X.set_array(du)
    prob.calc_jacobian(X, prob.J)
    prob.calc_residual(X, prob.R)
    rp, ci = oracle.csr_pattern(m)
    Kbc = oracle.assemble(m, rp, ci, apply_bc=True)
    F = oracle.form_rhs(m, 0.0, m.dt)
    Jr, Rr = oracle.snes_callbacks(m, rp, ci, Kbc, F, agg, du, rels_mode=mode, rels_nodes=nodes or (0, 1, 2, 3))
    grp, gci, gv = prob.J.get_csr()
    assert np.array_equal(grp, rp) and np.array_equal(gci, ci)      # sparsity incl. the cohesive node pairs
    pen = Jr > 1e29
    assert np.array_equal(gv[pen], Jr[pen])
    assert np.abs(gv[~pen] - Jr[~pen]).max() <= 1e-13 * np.abs(Jr[~pen]).max()
    R = prob.R.get_array()
    big = np.abs(Rr) > 1e20                                          # penalty rows: 1e30 * (du - prescribed)
    assert np.abs(R[big] - Rr[big]).max() <= 1e-13 * np.abs(Rr[big]).max()
    assert np.abs(R[~big] - Rr[~big]).max() <= 1e-12 * np.abs(Rr[~big]).max()
    X.destroy(); prob.destroy()


def test_element_by_element_product_is_refused_on_a_jacobian_with_cohesive_tangents(gpu_ctx):
    """The matrix-free product recomputes the solid elements only: asking for it on a Mat that carries (or is about to
    receive) cohesive tangents must fail loudly, not drop them."""
    m = read_inp(DECK)
    prob = driver.NonlinearProblem(gpu_ctx, m, "")
    X = capi.Vec(gpu_ctx, m.ndof); X.set_array(np.zeros(m.ndof))
    prob.calc_jacobian(X, prob.J)
    with pytest.raises(capi.FemError, match="cohesive tangents"):
        prob.J.set_mult_type("ebe")
    K2 = prob.K.duplicate()
    K2.set_mult_type("ebe_gather")                 # fine on the elastic matrix ...
    with pytest.raises(capi.FemError, match="solid elements only"):
        K2.add_cohesive(X, m.dt, 0, None)           # ... until tangents are added to it
    K2.destroy(); X.destroy(); prob.destroy()


def test_first_increment_reproduces_the_reference_log(gpu_ctx):
    """examples/SNES_test/cohesive_test/tmp:20-37 through the HIP path (see tests/test_oracle_golden.py for the deck):
    the old revision scaled the penalty force by dt, i.e. the prescribed values of this deck are t = 0.05 times the
    log's -- Vec_U of the first increment scales with them."""
    m = read_inp(LEGACY)
    res = driver.run_nonlinear(m, "", ctx=gpu_ctx, rels_mode=0, rels_nodes=(1, 2, 4, 5), recover_stress=False)
    log = np.array([2.59576e-32, 5e-05, 2.59576e-32, 2.52785e-05, -2.59576e-32, -6.05676e-32, -2.59576e-32, 5e-05,
                    -2.59576e-32, 2.52785e-05, 2.59576e-32, -6.05676e-32, -2.59576e-32, -2.47215e-05, 2.59576e-32, -2.47215e-05])
    u1 = res.U_hist[0] * m.t          # dt/t -> dt
    big = np.abs(log) > 1e-10
    assert res.snes_its[0] == 1 and res.snes_reasons[0] == 3
    assert np.abs(u1[big] / log[big] - 1.0).max() < 5e-6


@pytest.mark.parametrize("inner,tol", [("-ksp_type preonly -pc_type lu", 1e-9),
                                       ("-ksp_type bcgs -pc_type jacobi -ksp_rtol 1e-12 -ksp_atol 1e-16", 1e-5)])
def test_newton_on_the_reference_deck_matches_oracle(gpu_ctx, oracle, inner, tol):
    """Both inner solvers against the oracle's loop with its dense LU (the MUMPS stand-in).  With BiCGStab the first
    increment agrees to the linear tolerance; the second starts from a state that already solves it, and what the
    iterative solve then does with a right-hand side made of 1e30-weighted round-off is not comparable digit by digit."""
    m = read_inp(DECK)
    res = driver.run_nonlinear(m, inner, ctx=gpu_ctx)
    ref = oracle.snes_run(m, inner="lu")
    assert len(res.snes_its) == len(ref["its"]) == 2
    assert res.snes_reasons[0] == ref["reasons"][0] == 3 and res.snes_its[0] == ref["its"][0]
    assert min(res.snes_reasons) > 0
    if ref["reasons"][1] > 0 and "lu" in inner:      # (where both converge: the same verdict, not one of a pair)
        assert res.snes_reasons[1] == ref["reasons"][1], (res.snes_reasons, ref["reasons"])
    assert np.abs(res.U_hist[0] - ref["U_hist"][0]).max() <= 1e-9 * np.abs(ref["U_hist"][0]).max()
    assert np.abs(res.aggregate_u - ref["u"]).max() <= tol * np.abs(ref["u"]).max()
    assert np.abs(res.aggregate_stress - ref["stress"]).max() <= 10 * tol * np.abs(ref["stress"]).max()


def test_newton_with_the_elements_own_nodes(gpu_ctx, oracle):
    """rels_mode 1 (the element's own nodes in getCohRels), first increment.  (Later increments of these decks end in
    SNES_DIVERGED_LINE_SEARCH or in a stagnation verdict -- the reference's "residual" J du - F is not the gradient the
    line search assumes -- and the trial steps before that verdict amplify round-off: not comparable digit by digit.)"""
    for deck in (DECK, LEGACY):
        m = read_inp(deck)
        res = driver.run_nonlinear(m, "", ctx=gpu_ctx, rels_mode=1, recover_stress=False)
        ref = oracle.snes_run(m, rels_mode=1)
        assert res.snes_its[0] == ref["its"][0] == 1 and res.snes_reasons[0] == ref["reasons"][0] == 3
        assert np.abs(res.U_hist[0] - ref["U_hist"][0]).max() <= 1e-9 * np.abs(ref["U_hist"][0]).max()


def test_config5_tiling(gpu_ctx, oracle):
    """Config 5: examples/cohesive_test_2.inp x1000 (disjoint copies: 3,000 elements, 16,000 dofs), getCohRels as
    written -- every cohesive element reads the first four nodes of the rank, i.e. copy 0's, and copy 0 behaves like
    the single deck, so every copy must reproduce the single deck's increments.  Against the oracle on 40 copies (its
    dense LU is still cheap there)."""
    m1 = read_inp(DECK)
    one = driver.run_nonlinear(m1, "", ctx=gpu_ctx, recover_stress=False)
    big = driver.run_nonlinear(tile_deck(m1, 1000), "", ctx=gpu_ctx, recover_stress=False)
    # (the second increment starts from a state that already solves it: its Newton direction is round-off, and whether the line
    # search accepts it -- reason 4 -- or gives up on it -- reason -6 -- depends on that round-off, here as in the oracle; what
    # is compared is the first increment's verdict and the displacements after both)
    assert big.snes_reasons[0] == one.snes_reasons[0] == 3 and big.snes_its[0] == one.snes_its[0] == 1 and len(big.snes_its) == 2
    assert big.snes_reasons[1] in (4, -6) and one.snes_reasons[1] in (4, -6)
    U = big.aggregate_u.reshape(1000, -1)
    assert np.abs(U - one.aggregate_u[None]).max() <= 1e-10 * np.abs(one.aggregate_u).max()
    # whichever verdict the second increment gets, its step is round-off: the increment's displacement is the first one's again
    for r in (big, one):
        assert np.abs(r.U_hist[1] - r.U_hist[0]).max() <= 1e-9 * np.abs(r.U_hist[0]).max()
    m40 = tile_deck(m1, 40)
    a = driver.run_nonlinear(m40, "", ctx=gpu_ctx, recover_stress=False)
    b = oracle.snes_run(m40)
    # (the second increment starts converged: how its empty Newton step is counted and which verdict the line search
    # gives on a direction made of round-off differs between the two factorizations; the increments do not)
    assert a.snes_its[0] == b["its"][0] == 1 and a.snes_reasons[0] == b["reasons"][0] == 3 and min(a.snes_reasons) > 0
    if b["reasons"][1] > 0:                          # (where both converge: the same verdict, not one of a pair)
        assert a.snes_reasons[1] == b["reasons"][1], (a.snes_reasons, b["reasons"])
    assert np.abs(a.aggregate_u - b["u"]).max() <= 1e-9 * np.abs(b["u"]).max()


def test_config5_the_two_factorisations_agree(gpu_ctx, monkeypatch):
    """Round-3 advice: the batched block LU (one wavefront per connected component, the default on the x1000 tiling) replaced the
    band LU there in the same change that loosened test_config5_tiling.  Both factorisations on the same decks (FEM_AMD_LU_COMPONENTS
    is read at every set-up): the same verdict and iteration count in the first increment, the same displacements after both, and in
    the second increment -- whose Newton direction is round-off -- a verdict from the same pair and a step below 1e-9 of the field."""
    m1 = read_inp(DECK)
    for m in (m1, tile_deck(m1, 40), tile_deck(m1, 1000)):
        runs = {}
        for comp in ("1", "0"):
            monkeypatch.setenv("FEM_AMD_LU_COMPONENTS", comp)
            runs[comp] = driver.run_nonlinear(m, "", ctx=gpu_ctx, recover_stress=False)
        a, b = runs["1"], runs["0"]
        assert a.snes_reasons[0] == b.snes_reasons[0] == 3 and a.snes_its[0] == b.snes_its[0] == 1
        assert len(a.snes_its) == len(b.snes_its) == 2
        assert a.snes_reasons[1] in (4, -6) and b.snes_reasons[1] in (4, -6)
        assert np.abs(a.U_hist[0] - b.U_hist[0]).max() <= 1e-11 * np.abs(b.U_hist[0]).max()
        assert np.abs(a.aggregate_u - b.aggregate_u).max() <= 1e-9 * np.abs(b.aggregate_u).max()
        for r in (a, b):
            assert np.abs(r.U_hist[1] - r.U_hist[0]).max() <= 1e-9 * np.abs(r.U_hist[0]).max()


@pytest.mark.parametrize("rels_mode", [0, 1])
def test_config5_connected_strip(gpu_ctx, oracle, rels_mode):
    """Config 5 as SURVEY 8(d) words it -- the deck tiled along x with SHARED edge nodes: one connected strip, `copies`
    cohesive elements on one interface (a node sits in two of them: the coloured, atomic-free adds of the tangent and the
    traction).  40 copies against the oracle's Newton loop (first increment; the later ones end in a line-search verdict on
    both sides, see above), the callbacks' Jacobian twice for bitwise equality, and the x1000 strip (8,008 dofs) must run."""
    from fem_amd.mesh import tile_deck_connected
    m1 = read_inp(DECK)
    m40 = tile_deck_connected(m1, 40)
    a = driver.run_nonlinear(m40, "", ctx=gpu_ctx, rels_mode=rels_mode, recover_stress=False)
    b = oracle.snes_run(m40, rels_mode=rels_mode)
    assert a.snes_its[0] == b["its"][0] and a.snes_reasons[0] == b["reasons"][0] == 3
    assert np.abs(a.U_hist[0] - b["U_hist"][0]).max() <= 1e-9 * np.abs(b["U_hist"][0]).max()
    # the Jacobian of a state with open gaps, assembled twice: no atomics, same bits
    prob = driver.NonlinearProblem(gpu_ctx, m40, "", rels_mode, None)
    rng = np.random.default_rng(2)
    X = capi.Vec(gpu_ctx, m40.ndof); X.set_array(1e-3 * rng.standard_normal(m40.ndof))
    prob.calc_jacobian(X, prob.J)
    v1 = prob.J.get_csr()[2].copy()
    prob.calc_jacobian(X, prob.J)
    assert np.array_equal(v1, prob.J.get_csr()[2])
    X.destroy(); prob.destroy()
    big = driver.run_nonlinear(tile_deck_connected(m1, 1000), "", ctx=gpu_ctx, rels_mode=rels_mode, recover_stress=False)
    assert big.snes_reasons[0] == 3 and np.isfinite(big.aggregate_u).all()


@pytest.mark.parametrize("rels", [(0, None), (0, (1, 2, 4, 5)), (1, None)])
def test_builtin_callbacks_are_the_hand_written_ones(gpu_ctx, rels):
    """fem_snes_set_defmod_callbacks installs the reference's CalcJacobian / CalcResidual as the library's own callbacks: the same
    primitives in the same order as the hand-written pair of driver.NonlinearProblem -- the same bits, increment by increment, on
    the reference deck, on two_quads_qs.inp (shared cohesive nodes) and on a connected strip."""
    from fem_amd.mesh import tile_deck_connected
    mode, nodes = rels
    m1 = read_inp(DECK)
    decks = [m1, read_inp(os.path.join(HERE, "golden", "decks", "two_quads_qs.inp")), tile_deck_connected(m1, 40)]
    for m in decks:
        a = driver.run_nonlinear(m, "", ctx=gpu_ctx, rels_mode=mode, rels_nodes=nodes)
        b = driver.run_nonlinear(m, "", ctx=gpu_ctx, rels_mode=mode, rels_nodes=nodes, builtin_callbacks=True)
        assert list(a.snes_its) == list(b.snes_its) and list(a.snes_reasons) == list(b.snes_reasons)
        assert len(a.U_hist) == len(b.U_hist) and all(np.array_equal(x, y) for x, y in zip(a.U_hist, b.U_hist))
        assert np.array_equal(a.aggregate_u, b.aggregate_u) and np.array_equal(a.aggregate_stress, b.aggregate_stress)
