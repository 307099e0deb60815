"""CPU: the oracle (C restatement) against the committed golden vectors that
were generated from the REFERENCE's own element routines
(tests/golden/element_golden.json, made by tests/golden/gen_golden.py from
oracle/_ref), against the numbers the reference's own examples hold
(tests/golden/reference_examples.json), and -- where oracle/_ref has been
built -- directly against the reference library."""
import json
import os

import numpy as np
import pytest

from fem_amd.inp import read_inp
from oracle.oracle import RefLib, TRI, QUA, TET, HEX

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "element_golden.json")))
EX = json.load(open(os.path.join(HERE, "golden", "reference_examples.json")))
TYPES = {"tri": TRI, "qua": QUA, "tet": TET, "hex": HEX}
RT = 1e-13


def _close(a, b, rt=RT):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.abs(a - b).max() <= rt * max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", list(TYPES))
def test_tables_match_reference(oracle, name):
    t, g = TYPES[name], GOLD["types"][name]
    assert oracle.counts(t) == (g["nodecount"], g["nip"], g["dim"], g["nps"], g["vtkid"])
    ip, w, shp, inv = oracle.tables(t, g["dim"])
    assert _close(ip, g["ipoints"]) and _close(w, g["weights"]) and _close(shp, g["shape"])
    if g["n2inv"] is None:
        assert inv is None
    else:
        assert _close(inv, g["n2inv"], 1e-12)


@pytest.mark.parametrize("name", list(TYPES))
def test_element_routines_match_reference(oracle, name):
    t, g = TYPES[name], GOLD["types"][name]
    for c in g["cases"]:
        ec = np.array(c["ecoords"])
        assert _close(oracle.form_el_ke(t, ec, c["E"], c["nu"]), c["k"])
        assert _close(oracle.calc_el_stress(t, ec, c["edisp"], c["E"], c["nu"]), c["estress"])
        dN, detj = oracle.form_dn_detj(t, c["xi"], ec)
        assert _close(dN, c["dN"]) and abs(detj - c["detj"]) <= RT * abs(c["detj"])
        assert _close(oracle.dmat(g["dim"], c["E"], c["nu"]), c["dmat"])
        for s in c["sides"]:
            a, sn = oracle.edge_area_nodes(t, c["enodes"], ec, s["side"])
            assert abs(a - s["area"]) <= RT * s["area"] and list(sn) == s["snodes"]


def test_known_answers_quoted_in_survey(oracle):
    """unit elements with E=3640, nu=0.3 (values the reference's examples/tmp.txt also holds)."""
    k = oracle.form_el_ke(QUA, [[0, 0], [1, 0], [1, 1], [0, 1]], 3640.0, 0.3)
    assert np.allclose(k[0], [2100, 875, -1400, 175, -1050, -875, 350, -175], rtol=0, atol=1e-9)
    k = oracle.form_el_ke(HEX, [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], 3640.0, 0.3)
    assert np.allclose(k[0, :6], [855.5556, 291.6667, 291.6667, -388.8889, 58.3333, 58.3333], atol=1e-3)
    assert np.abs(k - k.T).max() < 1e-10 and np.abs(k.sum(1)).max() < 1e-9   # symmetric, rigid-body null space
    k = oracle.form_el_ke(TET, [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], 3640.0, 0.3)
    assert np.allclose(k[0, :6], [1283.33, 583.33, 583.33, -816.67, -233.33, -233.33], atol=1e-2)
    k = oracle.form_el_ke(TRI, [[0, 0], [1, 0], [0, 1]], 3640.0, 0.3)
    assert np.allclose(k[0], [3150, 1750, -2450, -700, -700, -1050], atol=1e-9)


def test_assembled_rows_of_reference_matview(oracle):
    """examples/tmp.txt: MatView rows of the assembled Jacobian (elastic block + 1e30 penalties)."""
    ex = EX["tmp_txt_jacobian"]
    conn = ex["conn"]
    nn = 5
    coords = np.zeros((nn, 2))
    for a, v in enumerate(conn):
        coords[v] = ex["coords"][a]
    coords[2] = [5.0, 5.0]   # node 2 does not belong to this element
    m = dict(pdim=2, eltype=[QUA], eptr=[0, 4], conn=conn, mat_id=[0], coords=coords,
             mats=[[ex["E"], ex["nu"], 0, 0, 0]], bcnode=[[0, 0, 0], [1, 0, 0]], bcval=[[0, 0], [0, 0]],
             fnode=[], fval=np.zeros((0, 4)), telsd=np.zeros((0, 2)), tval=np.zeros((0, 4)), t=1.0, dt=1.0)
    rp, ci = oracle.csr_pattern(m)
    v = oracle.assemble(m, rp, ci, apply_bc=True)
    for r, cols in ex["rows"].items():
        r = int(r)
        got = {int(ci[k]): v[k] for k in range(rp[r], rp[r + 1])}
        for c, val in cols.items():
            if val >= 1e29:
                assert got[int(c)] == float(np.float32(1e30))   # PENALTY_PARAM is a default-real literal
            else:
                assert abs(got[int(c)] - val) <= 1e-9, (r, c, got[int(c)], val)


def test_elastic_example_displacements_and_stress(oracle):
    """examples/SNES_test/elastic/*.vtk: u_y = 0.05 / 0.10, nodal stress (105, 245, 0) after 10 increments."""
    ex = EX["elastic_vtk"]
    m = read_inp(os.path.join(HERE, "golden", "decks", "elastic_test.inp"))
    assert m.nsteps == 10
    for ksp in ("lu", "cg"):
        u, s, its = oracle.run(m, ksp, rtol=1e-14)
        assert len(its) == 10
        assert np.abs(u.reshape(-1, 2) - np.array(ex["displacements"])).max() < 1e-12
        assert np.abs(s - np.array(ex["stress"])[None]).max() < 1e-9
        # and to the 6 decimals the reference wrote
        assert np.array_equal(np.round(u.reshape(-1, 2), 6) + 0.0, np.array(ex["displacements"]))
    assert np.allclose(m.coords, ex["points"])


def test_cg_against_dense_lu_on_force_loaded_mixed_deck(oracle):
    """examples/lin_two_quads_qs.inp (tri+qua, nodal forces): no reference output exists; the
    KSPCG restatement must agree with the dense LU of the same assembled system."""
    m = read_inp(os.path.join(HERE, "golden", "decks", "lin_two_quads_qs.inp"))
    u1, s1, _ = oracle.run(m, "lu")
    u2, s2, its = oracle.run(m, "cg", rtol=1e-14)
    assert np.abs(u1 - u2).max() <= 1e-10 * np.abs(u1).max()
    assert np.abs(s1 - s2).max() <= 1e-9 * np.abs(s1).max()
    assert np.abs(u1).max() > 0


@pytest.mark.skipif(not RefLib.available(), reason="oracle/_ref not built (reference absent)")
@pytest.mark.parametrize("name", list(TYPES))
def test_oracle_against_live_reference_library(oracle, name):
    t = TYPES[name]
    r = RefLib()
    base = np.array(GOLD["types"][name]["cases"][0]["ecoords"])
    pd = base.shape[1]
    rng = np.random.default_rng(11 + t)
    for _ in range(25):
        ec = base * rng.uniform(0.5, 2) + rng.uniform(-0.25, 0.25, base.shape)
        E, nu = rng.uniform(1, 1e12), rng.uniform(0, 0.49)
        u = rng.standard_normal(base.size)
        assert _close(oracle.form_el_ke(t, ec, E, nu), r.form_el_ke(t, ec, E, nu, pd), 2e-13)
        assert _close(oracle.calc_el_stress(t, ec, u, E, nu), r.calc_el_stress(t, ec, u, E, nu, pd), 2e-13)
    assert np.array_equal(r.form_el_indx(np.array([4, 9, 2], dtype=np.int32) + 1, pd) - 1,
                          np.concatenate([[pd * v + d for d in range(pd)] for v in (4, 9, 2)]))


def _vtk_numbers(text):
    """(header tokens, numeric rows) -- trailing blanks and the sign of a printed zero are not significant."""
    rows = []
    for ln in text.splitlines():
        t = ln.split()
        try:
            rows.append([float(x) + 0.0 for x in t])
        except ValueError:
            rows.append(t)
    return rows


def test_vtk_writer_reproduces_reference_rank_files(oracle):
    """examples/SNES_test/elastic/{0,1}_output_000000.vtk (2-rank run: rank 0 holds element 2,
    rank 1 element 1, nodes renumbered locally) from the oracle solution of the same deck."""
    from fem_amd.vtk import vtk_text
    m = read_inp(os.path.join(HERE, "golden", "decks", "elastic_test.inp"))
    u, s, _ = oracle.run(m, "lu")
    for rank, elems in ((0, [1]), (1, [0])):
        ref = open(os.path.join(HERE, "golden", "vtk", f"elastic_rank{rank}_output_000000.vtk")).read()
        got = vtk_text(m, u, s, elements=elems)
        assert _vtk_numbers(got) == _vtk_numbers(ref)
        # the byte layout is the reference's, line by line: no trailing blank, F0.3 / F0.6 digits, keywords;
        # only the sign of a printed zero (`-.000000`: round-off of the reference's MUMPS solve) is not pinned
        unsign = lambda ln: ln.replace("-.000000", ".000000").replace("-.000", ".000")   # noqa: E731
        assert [unsign(ln) for ln in got.splitlines()] == [unsign(ln) for ln in ref.splitlines()]
        assert got.endswith("\n") and not any(ln != ln.rstrip() for ln in got.splitlines())


def test_cohesive_helpers_match_reference_golden(oracle):
    """SURVEY 8 f1, element level: getCohValues / getCohRels / getCohGaps / Seplaw_1_Tract / Seplaw_1_Stiff of the
    oracle against vectors produced by the reference's own compiled modules
    (tests/golden/cohesive_golden.json, generator beside it)."""
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cohesive_golden.json")))
    assert len(d["cases"]) >= 20
    for c in d["cases"]:
        t, n, det = oracle.coh_values(c["ecoords"])
        assert np.allclose(t, c["tangent"], rtol=1e-15, atol=0) and np.allclose(n, c["normal"], rtol=1e-15, atol=0)
        assert abs(det - c["det"]) <= 1e-15 * abs(c["det"])
        for gp, g in enumerate(c["gp"]):
            ur, vr = oracle.coh_rels(c["u8"], gp, c["dt"])
            gg, vg = oracle.coh_gaps(t, n, ur, vr)
            tr, st = oracle.seplaw1(c["props"], gg, vg, c["dt"])
            for a, b in ((ur, g["urel"]), (vr, g["vrel"]), (gg, g["gap"]), (vg, g["vgap"]), (tr, g["tract"]), (st, g["stiff"])):
                b = np.asarray(b)
                assert np.abs(a - b).max() <= 1e-14 * max(np.abs(b).max(), 1e-300)


def test_cohesive_element_vector_and_matrix_follow_the_reference_as_written(oracle):
    """applyTract_1 / applyStiff_1 (src/m_global.F90:622-682, 762-845) are restated with the reference's two
    peculiarities; both are checked here from the pinned helpers: (1) the traction vector is the LAST
    integration point's only, (2) the stiffness adds an unscaled first term at every node pair."""
    rng = np.random.default_rng(7)
    ec = np.array([[0, 0], [1.3, 0.2], [1.3, 0.2], [0, 0.0]]) + rng.uniform(-1e-3, 1e-3, (4, 2))
    u8 = rng.uniform(-0.02, 0.02, 8)
    props = [100.0, 0.01, 0.01, 1.0, 0.0, 0.02]
    dt = 0.1
    tr, K = oracle.coh_apply_1(ec, u8, props, dt)
    t, n, det = oracle.coh_values(ec)
    cn = 0.5773502691896260
    N = lambda eta: np.array([0.5 * (1 - eta), 0.5 * (1 + eta), 0.5 * (1 + eta), 0.5 * (1 - eta)])  # noqa: E731
    sig = np.array([1.0, 1.0, -1.0, -1.0])
    pts = []
    for gp, eta in enumerate((-cn, cn)):
        ur, vr = oracle.coh_rels(u8, gp, dt)
        g, vg = oracle.coh_gaps(t, n, ur, vr)
        f, st = oracle.seplaw1(props, g, vg, dt)
        pts.append((N(eta), f, st))
    Nl, fl, _ = pts[-1]
    expect = np.array([sig[a] * Nl[a] * (fl[0] * n[d] + fl[1] * t[d]) * det for a in range(4) for d in range(2)])
    assert np.allclose(tr, expect, rtol=1e-14, atol=0)
    Ke = np.zeros((8, 8))
    for Ng, _, st in pts:
        for a in range(4):
            for b in range(4):
                for d1 in range(2):
                    for d2 in range(2):
                        first = (st[0, 0] * n[d1] + st[1, 0] * t[d1]) * n[d2]
                        second = (st[0, 1] * n[d1] + st[1, 1] * t[d1]) * t[d2] * sig[a] * sig[b] * Ng[a] * Ng[b] * det
                        Ke[2 * a + d1, 2 * b + d2] += first + second
    assert np.allclose(K, Ke, rtol=1e-13, atol=1e-13 * np.abs(Ke).max())


def test_cpu_baseline_leg_computes_what_the_oracle_computes(oracle):
    """bench.py's cpu_baseline (oracle/cpu_baseline.c: parallel sparsity, first-touch placement, fused CG
    passes, element-by-element product) against the line-by-line restatement on small meshes."""
    from fem_amd.mesh import structured_hex, structured_quad, hex_to_tets, quad_to_mixed
    for m in (structured_hex(5, 4, 3, jitter=0.2), quad_to_mixed(structured_quad(7, 5, jitter=0.1)),
              hex_to_tets(structured_hex(3, 3, 2, jitter=0.1))):
        rp, ci = oracle.csr_pattern(m)
        val = oracle.assemble(m, rp, ci, apply_bc=True)
        F = oracle.form_rhs(m, 0.0, m.dt)
        x, its, reason, rn, _ = oracle.ksp_cg_jacobi(rp, ci, val, F, rtol=1e-12)
        for nt in (1, 4):
            b = oracle.cpu_baseline(m, nt, 10000, rtol=1e-12, nprod=1, want_arrays=True)
            assert np.array_equal(b["rowptr"], rp) and np.array_equal(b["colind"], ci)
            assert np.abs(b["val"] - val).max() <= 1e-13 * np.abs(val[val < 1e29]).max()
            assert b["reason"] == reason and abs(b["its"] - its) <= 2
            assert np.abs(b["x"] - x).max() <= 1e-9 * np.abs(x).max()
        # matrix-free product = assembled product without the penalty INSERT
        v0 = oracle.assemble(m, rp, ci, apply_bc=False)
        xx = np.random.default_rng(3).standard_normal(m.ndof)
        y = oracle.cpu_ebe_mult(m, xx, nthreads=3)
        yr = oracle.spmv(rp, ci, v0, xx)
        assert np.abs(y - yr).max() <= 1e-12 * np.abs(yr).max()


def test_newton_cohesive_first_increment_reproduces_the_reference_log(oracle):
    """examples/SNES_test/cohesive_test/tmp:20-37 -- the reference's own two-rank run of the cohesive deck (re-typed for
    the current reader as tests/golden/decks/cohesive_test_legacy.inp; that older revision scaled the penalty force by
    dt, `bc_legacy`; rank 0 held global nodes 2,3,5,6,7,8, so getCohRels read nodes 2,3,5,6: 0_output_000000.vtk).
    Increment 1: SNES iteration count 1 and Vec_U to the six printed digits, the 1e-32 round-off entries included."""
    m = read_inp(os.path.join(HERE, "golden", "decks", "cohesive_test_legacy.inp"))
    r = oracle.snes_run(m, rels_mode=0, rels_nodes=(1, 2, 4, 5), bc_legacy=True, maxsteps=1)
    log = np.array([2.59576e-32, 5e-05, 2.59576e-32, 2.52785e-05, -2.59576e-32, -6.05676e-32, -2.59576e-32, 5e-05,
                    -2.59576e-32, 2.52785e-05, 2.59576e-32, -6.05676e-32, -2.59576e-32, -2.47215e-05, 2.59576e-32, -2.47215e-05])
    assert r["its"][0] == 1 and r["reasons"][0] == 3          # SNES_CONVERGED_FNORM_RELATIVE after one iteration
    assert np.abs(r["U_hist"][0] - log).max() <= 6e-6 * np.abs(log).max()
    big = np.abs(log) > 1e-10
    assert np.abs(r["U_hist"][0][big] / log[big] - 1.0).max() < 5e-6
    small = ~big                                                # the round-off entries: same sign, same six digits
    assert np.abs(r["U_hist"][0][small] / log[small] - 1.0).max() < 5e-6


def test_newton_cohesive_current_deck_runs_and_inner_solvers_agree(oracle):
    """examples/cohesive_test_2.inp (config 5's deck) through the restated SNES loop: both inner solvers (dense LU
    stand-in for MUMPS, BiCGStab + Jacobi as in the HIP library) give the same increments."""
    m = read_inp(os.path.join(HERE, "golden", "decks", "cohesive_test_2.inp"))
    a = oracle.snes_run(m, inner="lu")
    b = oracle.snes_run(m, inner="bcgs", ksp_rtol=1e-12, ksp_abstol=1e-16)
    assert len(a["its"]) == 2 and (a["reasons"] > 0).all() and (b["reasons"] > 0).all()
    assert a["its"][0] == b["its"][0] == 1
    assert np.abs(a["U_hist"][0] - b["U_hist"][0]).max() <= 1e-9 * np.abs(a["U_hist"][0]).max()
    # the second increment starts from a state that solves it: the direct solve confirms it (an empty Newton step), the
    # iterative one, whose first increment carries its own 1e-12 of error into 1e30-weighted rows, moves a little
    assert np.abs(a["u"] - b["u"]).max() <= 1e-5 * np.abs(a["u"]).max()


def test_el_test_deck_is_a_uniform_stress_state(oracle):
    """examples/el_test.inp (the reference's only traction deck, re-typed: tests/golden/decks/el_test.inp): the three tractions
    and three nodal forces are in equilibrium with sigma_xx = sigma_yy = tau_xy = -1000 -- ApplyTraction's side numbering and
    sign (src/m_global.F90:343-368, src/m_elems.F90:286-288) pinned by an exact solution; LU and CG agree."""
    m = read_inp(os.path.join(HERE, "golden", "decks", "el_test.inp"))
    u, s, _ = oracle.run(m, "lu")
    assert np.abs(s + 1000.0).max() <= 1e-9 * 1000.0
    u2, s2, _ = oracle.run(m, "cg", rtol=1e-14)
    assert np.abs(u - u2).max() <= 1e-12 * np.abs(u).max()
    # plane strain, E = 3e6, nu = 0.3: eps_xx = eps_yy = (1 + nu)(1 - 2 nu) sigma / E ... gamma_xy = tau / G
    E, nu = 3.0e6, 0.3
    exx = (1 + nu) * ((1 - nu) * -1000.0 - nu * -1000.0) / E
    gxy = -1000.0 * 2 * (1 + nu) / E
    U = u.reshape(-1, 2)
    assert abs((U[1, 0] - U[0, 0]) / 2.0 - exx) <= 1e-12 * abs(exx)            # node 2 is 2.0 to the right of node 1
    assert abs((U[3, 1] - U[0, 1]) / 1.0 - exx) <= 1e-12 * abs(exx)
    assert abs((U[3, 0] - U[0, 0]) / 1.0 + (U[1, 1] - U[0, 1]) / 2.0 - gxy) <= 1e-12 * abs(gxy)


def test_two_quads_qs_deck_reads_as_the_reference_wrote_it(oracle):
    """examples/two_quads_qs.inp verbatim (current 8-integer header): 4 tri + 2 qua + 2 coh sharing nodes 5 / 8, one
    separation law, three BC records of which one has free flags, two nodal forces; the oracle's Newton loop runs it."""
    m = read_inp(os.path.join(HERE, "golden", "decks", "two_quads_qs.inp"))
    assert (m.nels, m.nnds, len(m.cohmats), len(m.fnode), len(m.bcnode)) == (8, 12, 1, 2, 3)
    assert list(m.bcnode[2]) == [1, 1, 1] and m.cohmats[0] == (1, [100.0, 0.01, 0.01, 1.0, 0.0, 0.0])
    r = oracle.snes_run(m, rels_mode=0)
    assert list(r["its"]) == [1, 1] and all(x > 0 for x in r["reasons"])
