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


def test_fortran_element_api_known_answers():
    _need(TEL)
    p = subprocess.run([TEL], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = {ln.split()[0]: [float(x) for x in ln.split()[1:]] for ln in p.stdout.splitlines() if ln.strip()}
    assert np.allclose(lines["KQ1"], [2100, 875, -1400, 175, -1050, -875, 350, -175], atol=1e-3)   # examples/tmp.txt:23
    assert lines["INDX"] == [7, 8, 17, 18, 3, 4, 13, 14]                                          # FormElIndx, 1-based
    assert np.allclose(lines["SQ1"], [105.0, 245.0, 0.0], atol=1e-6)                              # elastic golden stress
    assert lines["CNT"] == [8, 8, 3, 4, 12]
    assert np.allclose(lines["KH1"], [855.5556, 291.6667, 291.6667, -388.8889, 58.3333, 58.3333], atol=1e-3)


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
    got = np.array([[float(x) for x in r[2:]] for r in rows])
    expect_u = np.array([[0, .1], [0, .05], [0, 0], [0, .1], [0, .05], [0, 0]])
    assert np.abs(got[:, :2] - expect_u).max() < 1e-12
    assert np.abs(got[:, 2:] - np.array([105.0, 245.0, 0.0])).max() < 1e-9
