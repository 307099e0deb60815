"""GPU: the host-Fortran face of the boundary -- the ISO_C_BINDING interface
module, the `local`-compatible element module and the main.F90-shaped driver,
built by AMD flang (fortran/Makefile), run against the reference's example deck."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fortran", "defmod_amd")
TEL = os.path.join(ROOT, "fortran", "test_elems_amd")


def _need(path):
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (no flang at build time)")


def test_fortran_element_api_matches_reference_goldens(oracle):
    """The element-level Fortran API with the reference's own names and signatures (Initialize(ipoints, weights),
    FormElKE, CalcElStress, FormdNdetJ, BMat, DMat, ShapeFunc, ShapeFuncd, FormElIndx, get*) over the 28 golden cases
    the reference's compiled modules produced (tests/golden/element_golden.json): 1e-13 of the largest entry."""
    import json
    _need(TEL)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "element_golden.json")))["types"]
    p = subprocess.run([TEL, os.path.join(ROOT, "tests", "golden", "element_cases.txt")], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    cases, cur = [], None
    for ln in p.stdout.splitlines():
        t = ln.split()
        if not t:
            continue
        if t[0] == "CASE":
            cur = {"type": t[1], "n": int(t[2]), "pd": int(t[3]), "nip": int(t[4]), "cp": int(t[5])}
            cases.append(cur)
        elif t[0] in ("INDX", "CNT"):
            cur[t[0]] = [int(x) for x in t[1:]]
        else:
            cur[t[0]] = np.array([float(x) for x in t[1:]])
    assert len(cases) == 28
    close = lambda a, b, tol=1e-13: np.abs(np.asarray(a) - np.asarray(b)).max() <= tol * max(np.abs(np.asarray(b)).max(), 1e-300)  # noqa: E731
    seen = {}
    for c in cases:
        g = gold[c["type"]]
        k = seen.get(c["type"], 0)
        seen[c["type"]] = k + 1
        ref = g["cases"][k]
        n, pd, nip, cp = c["n"], c["pd"], c["nip"], c["cp"]
        assert close(c["K"].reshape(n * pd, n * pd, order="F"), np.array(ref["k"]))
        assert close(c["S"].reshape(nip, cp, order="F"), np.array(ref["estress"]))
        dN = c["DN"].reshape(pd, n, order="F")
        assert close(dN, np.array(ref["dN"])) and abs(c["DETJ"][0] - ref["detj"]) <= 1e-13 * abs(ref["detj"])
        assert close(c["D"].reshape(cp, cp, order="F"), np.array(ref["dmat"]))
        assert close(c["B"].reshape(cp, n * pd, order="F"), oracle.bmat(pd, n, np.array(ref["dN"])), 1e-13)
        assert close(c["N1"], np.array(g["shape"])[0])
        assert close(c["IP"].reshape(nip, pd, order="F"), np.array(g["ipoints"])) and close(c["W"], g["weights"])
        # ShapeFuncd at xi, then J = dNxi X must reproduce the golden detJ
        dnxi = c["DNXI"].reshape(pd, n, order="F")
        assert abs(np.linalg.det(dnxi @ np.array(ref["ecoords"])) - ref["detj"]) <= 1e-12 * abs(ref["detj"])
        assert c["INDX"] == [int(v) for v in ref["indx"]]
        assert c["CNT"] == [g["nodecount"], g["nip"], g["dim"], g["nps"], g["vtkid"]]


def test_fortran_driver_reproduces_elastic_golden(tmp_path, oracle):
    _need(EXE)
    deck = os.path.join(ROOT, "tests", "golden", "decks", "elastic_test.inp")
    p = subprocess.run([EXE, "-f", deck, "-ksp_rtol", "1e-14", "-o", str(tmp_path / "0_output")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("Interval:") == 10 and "Finished" in p.stdout
    full = np.loadtxt(tmp_path / "0_output_000000.full.txt", skiprows=1)
    u, s = full[:, :2], full[:, 2:]
    expect_u = np.array([[0, .1], [0, .05], [0, 0], [0, .1], [0, .05], [0, 0]])
    assert np.abs(u - expect_u).max() < 1e-12 and np.abs(s - np.array([105.0, 245.0, 0.0])).max() < 1e-9
    # the 6-decimal VTK block carries the reference's numbers (examples/SNES_test/elastic/*.vtk)
    vtk = open(tmp_path / "0_output_000000.vtk").read()
    assert "105.000000 245.000000" in vtk and ".000000 .100000 .000000" in vtk and "CELL_TYPES 2" in vtk
    # and the same deck through the oracle
    from fem_amd.inp import read_inp
    uo, so, _ = oracle.run(read_inp(deck), "cg", rtol=1e-14)
    assert np.abs(u.ravel() - uo).max() < 1e-12 and np.abs(s - so).max() < 1e-9


def test_fortran_driver_mixed_deck_matches_oracle(tmp_path, oracle):
    _need(EXE)
    deck = os.path.join(ROOT, "tests", "golden", "decks", "lin_two_quads_qs.inp")
    p = subprocess.run([EXE, "-f", deck, "-ksp_rtol", "1e-14", "-o", str(tmp_path / "0_output")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    full = np.loadtxt(tmp_path / "0_output_000000.full.txt", skiprows=1)
    from fem_amd.inp import read_inp
    uo, so, _ = oracle.run(read_inp(deck), "lu")
    assert np.abs(full[:, :2].ravel() - uo).max() <= 1e-10 * np.abs(uo).max()
    assert np.abs(full[:, 2:] - so).max() <= 1e-9 * np.abs(so).max()


def test_fortran_driver_on_the_traction_deck(tmp_path, oracle):
    """examples/el_test.inp (the reference's only deck with tractions; element record re-typed) through the Fortran driver: the
    Fortran reader's traction records, ApplyTraction inside fem_form_rhs -- the uniform stress state sigma_xx = sigma_yy = tau_xy =
    -1000 of that deck, and the oracle's displacements."""
    _need(EXE)
    deck = os.path.join(ROOT, "tests", "golden", "decks", "el_test.inp")
    for extra in ([], ["-ksp_type", "preonly", "-pc_type", "lu"], ["-mat_mult_type", "ebe"]):
        p = subprocess.run([EXE, "-f", deck, "-ksp_rtol", "1e-14", "-o", str(tmp_path / "0_output")] + extra,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        full = np.loadtxt(tmp_path / "0_output_000000.full.txt", skiprows=1)
        from fem_amd.inp import read_inp
        uo, so, _ = oracle.run(read_inp(deck), "lu")
        free = np.abs(uo) > 1e-20
        assert np.abs(full[:, :2].ravel()[free] - uo[free]).max() <= 1e-10 * np.abs(uo).max()
        assert np.abs(full[:, 2:] + 1000.0).max() <= 1e-8 * 1000.0


def test_fortran_driver_with_the_reference_solver_options(tmp_path, oracle):
    """`-ksp_type preonly -pc_type lu` (what src/main.F90:355-358 selects) through the Fortran driver:
    ten increments re-use one factorisation and give the golden numbers of the elastic deck."""
    _need(EXE)
    deck = os.path.join(ROOT, "tests", "golden", "decks", "elastic_test.inp")
    p = subprocess.run([EXE, "-f", deck, "-ksp_type", "preonly", "-pc_type", "lu", "-o", str(tmp_path / "0_output")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("Interval:") == 10 and "Finished" in p.stdout
    full = np.loadtxt(tmp_path / "0_output_000000.full.txt", skiprows=1)
    expect_u = np.array([[0, .1], [0, .05], [0, 0], [0, .1], [0, .05], [0, 0]])
    assert np.abs(full[:, :2] - expect_u).max() < 1e-12
    assert np.abs(full[:, 2:] - np.array([105.0, 245.0, 0.0])).max() < 1e-9


def test_fortran_driver_with_the_element_by_element_product(tmp_path, oracle):
    """`-mat_mult_type ebe` through the Fortran driver on the mixed tri+qua deck (forces over a time window,
    several increments): same answer as the oracle."""
    _need(EXE)
    deck = os.path.join(ROOT, "tests", "golden", "decks", "lin_two_quads_qs.inp")
    p = subprocess.run([EXE, "-f", deck, "-ksp_rtol", "1e-14", "-mat_mult_type", "ebe", "-o", str(tmp_path / "0_output")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    full = np.loadtxt(tmp_path / "0_output_000000.full.txt", skiprows=1)
    from fem_amd.inp import read_inp
    uo, so, _ = oracle.run(read_inp(deck), "lu")
    assert np.abs(full[:, :2].ravel() - uo).max() <= 1e-10 * np.abs(uo).max()
    assert np.abs(full[:, 2:] - so).max() <= 1e-9 * np.abs(so).max()


def test_fortran_driver_tightens_and_re_solves(tmp_path):
    """`defmod_amd -ksp_rtol 1e-6 -resolve_rtol 1e-13` on a hex8 deck with four identical increments: increments 2-4 are the
    kept solution (-ksp_reuse_solution, default on), and the KSPSetTolerances + KSPSolve of the same F after them (the
    reference's calls, src/main.F90:349-351,247) iterates under the new tolerance instead of handing out the kept 1e-6 one."""
    _need(EXE)
    import re
    from fem_amd.inp import write_inp
    from fem_amd.mesh import structured_hex
    m = structured_hex(7, 6, 5, jitter=0.12)
    m.t, m.dt = 1.0, 0.25
    deck = str(tmp_path / "hex.inp")
    write_inp(deck, m)
    p = subprocess.run([EXE, "-f", deck, "-ksp_rtol", "1e-6", "-resolve_rtol", "1e-13", "-o", str(tmp_path / "0_output")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    g = re.search(r"Re-solve at rtol\s+\S+: iterations (\d+)\s+\(last increment (\d+)\)\s+reason (-?\d+)\s+reused before (\d+) after (\d+)", p.stdout)
    assert g, p.stdout
    its13, its6, reason, before, after = (int(v) for v in g.groups())
    assert before == 3 and after == 3 and reason > 0 and its13 > its6


@pytest.mark.parametrize("np_", [1, 2])
def test_fortran_driver_on_a_hex8_deck_with_the_one_pass_product(tmp_path, oracle, np_):
    """A 3-D deck of distorted hex8 elements written in the reference's format, through the Fortran driver with
    `-mat_mult_type ebe` -- on such a mesh the one-pass tile product -- on one domain and on two (in-process communicator):
    displacements and nodal stresses of the oracle."""
    _need(EXE)
    from fem_amd.inp import write_inp
    from fem_amd.mesh import structured_hex
    m = structured_hex(7, 6, 5, jitter=0.12)
    deck = str(tmp_path / "hex.inp")
    write_inp(deck, m)
    cmd = [EXE, "-f", deck, "-ksp_rtol", "1e-13", "-mat_mult_type", "ebe", "-o", str(tmp_path / "0_output")]
    if np_ > 1:
        cmd += ["-np", str(np_)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    uo, so, _ = oracle.run(m, "lu")
    if np_ == 1:
        full = np.loadtxt(tmp_path / "0_output_000000.full.txt", skiprows=1)
        assert np.abs(full[:, :3].ravel() - uo).max() <= 1e-10 * np.abs(uo).max()
        assert np.abs(full[:, 3:] - so).max() <= 1e-9 * np.abs(so).max()
    else:
        seen = np.zeros(m.nnds, dtype=bool)
        for r in range(np_):
            full = np.loadtxt(tmp_path / f"{r}_output_000000.full.txt", skiprows=1, ndmin=2)
            gid = full[:, 0].astype(int)
            seen[gid] = True
            assert np.abs(full[:, 1:4] - uo.reshape(-1, 3)[gid]).max() <= 1e-10 * np.abs(uo).max()
            assert np.abs(full[:, 4:] - so[gid]).max() <= 1e-9 * np.abs(so).max()
        assert seen.all()


def test_plain_c_client_of_the_abi(tmp_path):
    """include/fem_amd.h consumed by gcc as C99 (no C++ types leak through the boundary)."""
    exe = tmp_path / "abi_smoke"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe),
                           "-L", os.path.join(ROOT, "fem_amd", "lib"), "-lfem_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "fem_amd", "lib")])
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    rows = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("NODE")]
    assert "STEPS 10" in p.stdout and len(rows) == 6
    # fem_ksp_set_tolerances between two solves of one F: the tighter solve iterates (no kept solution), the third is the kept one
    its_a, its_b, its_c, reused_b, reused_c = [int(v) for v in next(ln for ln in p.stdout.splitlines() if ln.startswith("RETOL")).split()[1:]]
    assert reused_b == 0 and reused_c == 1 and its_b >= its_a and its_c == its_b
    got = np.array([[float(x) for x in r[2:]] for r in rows])
    expect_u = np.array([[0, .1], [0, .05], [0, 0], [0, .1], [0, .05], [0, 0]])
    assert np.abs(got[:, :2] - expect_u).max() < 1e-12
    assert np.abs(got[:, 2:] - np.array([105.0, 245.0, 0.0])).max() < 1e-9


def test_fortran_two_domains_reproduce_the_reference_rank_files(tmp_path):
    """`defmod_amd -np 2` on the elastic deck -- the reference's own test is `mpiexec -n 2` (Makefile:6): the RCB
    partition puts element 2 on rank 0 and element 1 on rank 1, as METIS did in the run that wrote
    examples/SNES_test/elastic/{0,1}_output_000000.vtk; the two rank files must come out line by line (only the
    sign of a printed zero is round-off of the reference's MUMPS solve)."""
    _need(EXE)
    deck = os.path.join(ROOT, "tests", "golden", "decks", "elastic_test.inp")
    p = subprocess.run([EXE, "-f", deck, "-np", "2", "-ksp_rtol", "1e-14", "-o", str(tmp_path / "0_output")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("Interval:") == 10 and "Partitioning mesh" in p.stdout and "Finished" in p.stdout
    unsign = lambda ln: ln.replace("-.000000", ".000000").replace("-.000", ".000").rstrip()   # noqa: E731
    for rank in (0, 1):
        got = open(tmp_path / f"{rank}_output_000000.vtk").read().splitlines()
        ref = open(os.path.join(ROOT, "tests", "golden", "vtk", f"elastic_rank{rank}_output_000000.vtk")).read().splitlines()
        assert [unsign(x) for x in got] == [unsign(x) for x in ref]


@pytest.mark.parametrize("np_", [2, 3])
def test_fortran_multi_domain_mixed_deck_matches_oracle(tmp_path, oracle, np_):
    """The force-loaded tri+qua deck on 2 and 3 domains of one GPU through the Fortran driver: every rank's
    displacements and nodal stresses (global node ids in the companion file) against the serial oracle, and the
    copies of an interface node bitwise equal on all ranks that hold it."""
    _need(EXE)
    deck = os.path.join(ROOT, "tests", "golden", "decks", "lin_two_quads_qs.inp")
    p = subprocess.run([EXE, "-f", deck, "-np", str(np_), "-ksp_rtol", "1e-14", "-o", str(tmp_path / "0_output")],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    from fem_amd.inp import read_inp
    uo, so, _ = oracle.run(read_inp(deck), "lu")
    uo = uo.reshape(-1, 2)
    seen = {}
    covered = set()
    for rank in range(np_):
        full = np.loadtxt(tmp_path / f"{rank}_output_000000.full.txt", skiprows=1, ndmin=2)
        gid = full[:, 0].astype(int)
        assert np.abs(full[:, 1:3] - uo[gid]).max() <= 1e-10 * np.abs(uo).max()
        assert np.abs(full[:, 3:] - so[gid]).max() <= 1e-9 * np.abs(so).max()
        for g, row in zip(gid, full[:, 1:]):
            if g in seen:
                assert np.array_equal(seen[g], row)
            seen[g] = row
        covered |= set(gid.tolist())
    assert covered == set(range(uo.shape[0]))
