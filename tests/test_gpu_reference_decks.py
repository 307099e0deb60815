"""GPU: the reference's example decks that round 3 never ran (verdict, missing #3), HIP path against the oracle:

* examples/two_quads_qs.inp as it stands (the reference's largest deck in the current 8-integer format: tri + qua, two
  cohesive elements sharing nodes 5 / 8, nodal forces, a BC record with free flags),
* examples/el_test.inp (its only deck with tractions -> ApplyTraction, src/m_global.F90:343-368), element record re-typed,
* examples/cohesive_test_microns.inp (older header) re-typed like cohesive_test_legacy.inp.
"""
import os

import numpy as np
import pytest

from fem_amd import driver
from fem_amd.inp import read_inp

pytestmark = pytest.mark.gpu
DECKS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decks")


@pytest.mark.parametrize("opts,tol", [("-ksp_type preonly -pc_type lu", 1e-12), ("-ksp_type cg -pc_type jacobi -ksp_rtol 1e-14", 1e-11),
                                      ("-ksp_type cg -pc_type jacobi -ksp_rtol 1e-14 -mat_mult_type ebe", 1e-11)])
def test_el_test_tractions(gpu_ctx, oracle, opts, tol):
    """Three tractions on one quad: ApplyTraction's edge integrals + nodal forces; the deck is a uniform stress state
    (sigma_xx = sigma_yy = tau_xy = -1000 exactly), which pins the traction sign and side numbering end to end."""
    m = read_inp(os.path.join(DECKS, "el_test.inp"))
    assert len(m.telsd) == 3 and len(m.fnode) == 3
    res = driver.run(m, opts, ctx=gpu_ctx)
    u, s, _ = oracle.run(m, "lu")
    free = np.abs(u) > 1e-20                        # (the penalised dofs are 1e-27: round-off of F / 1e30)
    assert np.abs(res.aggregate_u[free] - u[free]).max() <= tol * np.abs(u).max()
    assert np.abs(res.aggregate_stress - s).max() <= 100 * tol * np.abs(s).max()
    assert np.abs(res.aggregate_stress + 1000.0).max() <= 1e-9 * 1000.0
    F = oracle.form_rhs(m, 0.0, m.dt)
    prob = driver.Problem(gpu_ctx, m, opts)
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    assert np.array_equal(prob.F.get_array(), F)    # FormRHS with tractions: bit for bit
    prob.destroy()


@pytest.mark.parametrize("deck", ["two_quads_qs.inp", "cohesive_test_microns.inp"])
@pytest.mark.parametrize("rels_mode", [0, 1])
@pytest.mark.parametrize("inner", ["-ksp_type preonly -pc_type lu", "-ksp_type bcgs -pc_type jacobi -ksp_rtol 1e-12 -ksp_atol 1e-16"])
def test_cohesive_decks_newton(gpu_ctx, oracle, deck, rels_mode, inner):
    """Newton increments of the two 8-element cohesive decks, both readings of getCohRels (as written: the first four local
    nodes; the element's own nodes), both inner solvers, against the oracle's loop with its dense LU: the same verdict and
    iteration count per increment, the same iterate wherever an increment converged."""
    m = read_inp(os.path.join(DECKS, deck))
    assert m.nels == 8 and len(m.cohmats) == 1
    res = driver.run_nonlinear(m, inner, ctx=gpu_ctx, rels_mode=rels_mode, recover_stress=True)
    ref = oracle.snes_run(m, rels_mode=rels_mode, inner="lu")
    n = min(len(res.snes_its), len(ref["its"]))
    assert n >= 1
    lu = "lu" in inner
    for k in range(n):
        ok_ref, ok = ref["reasons"][k] > 0, res.snes_reasons[k] > 0
        if lu:
            # (from the second increment on the Newton loop starts next to its answer -- the loads of these decks do not change --
            # and whether the step-norm test passes after one iteration or two is decided by round-off: +-1)
            assert res.snes_reasons[k] == ref["reasons"][k], (k, res.snes_reasons, ref["reasons"])
            assert abs(res.snes_its[k] - ref["its"][k]) <= (0 if k == 0 else 1), (k, res.snes_its, ref["its"])
        else:
            if res.snes_reasons[k] == -3 and ok_ref:
                # SNES_DIVERGED_LINEAR_SOLVE from BiCGStab where the dense LU went through: legitimate only where the Jacobian is
                # singular to working precision (two_quads_qs.inp at zero opening: the block under the interface is held by the
                # cohesive tangent alone, cond(D^-1 J) ~ 1e16 -- LU with pivoting returns *a* vector, a Krylov method cannot)
                rp, ci = oracle.csr_pattern(m)
                Kbc = oracle.assemble(m, rp, ci, apply_bc=True)
                z = np.zeros(m.ndof)
                Jr, _ = oracle.snes_callbacks(m, rp, ci, Kbc, oracle.form_rhs(m, 0.0, m.dt), z, z, rels_mode=rels_mode, rels_nodes=(0, 1, 2, 3))
                A = np.zeros((m.ndof, m.ndof))
                for r in range(m.ndof):
                    A[r, ci[rp[r]:rp[r + 1]]] = Jr[rp[r]:rp[r + 1]]
                assert np.linalg.cond(A / np.diag(A)[:, None]) > 1e14
                break
            assert ok == ok_ref, (k, res.snes_reasons, ref["reasons"])
        if not (ok and ok_ref):
            break
        tol = 1e-9 if lu else 1e-6
        assert np.abs(res.U_hist[k] - ref["U_hist"][k]).max() <= tol * max(np.abs(ref["U_hist"][k]).max(), 1e-300)
    if all(r > 0 for r in ref["reasons"]) and lu:
        assert np.abs(res.aggregate_u - ref["u"]).max() <= 1e-9 * np.abs(ref["u"]).max()
        assert np.abs(res.aggregate_stress - ref["stress"]).max() <= 1e-8 * np.abs(ref["stress"]).max()


def test_two_quads_callbacks_with_shared_cohesive_nodes(gpu_ctx, oracle):
    """two_quads_qs.inp: nodes 5 and 8 sit in both cohesive elements -- the coloured (atomic-free) adds of tangent and traction
    must add both elements' contributions; Jacobian and residual on a random state against the oracle, twice for bitwise
    reproducibility."""
    from fem_amd import capi
    m = read_inp(os.path.join(DECKS, "two_quads_qs.inp"))
    for mode in (0, 1):
        prob = driver.NonlinearProblem(gpu_ctx, m, "", rels_mode=mode)
        rng = np.random.default_rng(11)
        agg = 1e-3 * rng.standard_normal(m.ndof)
        du = 1e-3 * rng.standard_normal(m.ndof)
        prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
        prob.agg.set_array(agg)
        X = capi.Vec(gpu_ctx, m.ndof); X.set_array(du)
        prob.calc_jacobian(X, prob.J); prob.calc_residual(X, prob.R)
        j1, r1 = prob.J.get_csr()[2], prob.R.get_array()
        prob.calc_jacobian(X, prob.J); prob.calc_residual(X, prob.R)
        assert np.array_equal(j1, prob.J.get_csr()[2]) and np.array_equal(r1, prob.R.get_array())
        rp, ci = oracle.csr_pattern(m)
        Kbc = oracle.assemble(m, rp, ci, apply_bc=True)
        F = oracle.form_rhs(m, 0.0, m.dt)
        Jr, Rr = oracle.snes_callbacks(m, rp, ci, Kbc, F, agg, du, rels_mode=mode, rels_nodes=(0, 1, 2, 3))
        pen = Jr > 1e29
        assert np.array_equal(j1[pen], Jr[pen])
        assert np.abs(j1[~pen] - Jr[~pen]).max() <= 1e-13 * np.abs(Jr[~pen]).max()
        big = np.abs(Rr) > 1e20
        assert np.abs(r1[big] - Rr[big]).max() <= 1e-13 * np.abs(Rr[big]).max()
        assert np.abs(r1[~big] - Rr[~big]).max() <= 1e-12 * np.abs(Rr[~big]).max()
        X.destroy(); prob.destroy()
