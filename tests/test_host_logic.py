"""CPU: host-side logic -- deck reader, mesh generators, partition/halo plans,
and that the C-ABI library loads and exports every symbol include/fem_amd.h declares."""
import os
import re

import numpy as np
import pytest

from fem_amd import partition
from fem_amd.inp import parse_inp, read_inp
from fem_amd.mesh import HEX, QUA, TRI, structured_hex, structured_quad, hex_to_tets, quad_to_mixed

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_inp_reader_current_format():
    m = read_inp(os.path.join(HERE, "golden", "decks", "lin_two_quads_qs.inp"))
    assert (m.pdim, m.nels, m.nnds, m.mats.shape[0]) == (2, 6, 9, 2)
    assert list(m.eltype) == [QUA, TRI, TRI, TRI, TRI, QUA]
    assert list(m.element_nodes(0)) == [3, 4, 7, 6] and list(m.element_nodes(1)) == [4, 8, 7]
    assert m.bcnode.tolist() == [[0, 0, 0], [6, 0, 0], [1, 1, 1]]
    assert m.fnode.tolist() == [8, 2] and m.fval[0].tolist() == [-10.0e10, 0.0, 0.0, 0.1]
    assert (m.t, m.dt, m.nsteps) == (0.1, 0.1, 1)


def test_inp_reader_clamps_force_window_and_rejects_bad_decks():
    deck = """implicit 2 10
1 4 1 0 0 1 1 1
0.5 0.25
qua 1 2 3 4 1 0
0 0
1 0
1 1
0 1
100.0 0.25 0 0 0
1 0 0 0.0 0.0
3 1.0 2.0 0.0 9.0   ! t2 beyond t is clamped (m_io.F90:357-358)
1 2 5.0 0.0 0.0 1.0
"""
    m = parse_inp(deck)
    assert m.fval[0].tolist() == [1.0, 2.0, 0.0, 0.5]
    assert m.telsd.tolist() == [[0, 2]] and m.tval[0].tolist() == [5.0, 0.0, 0.0, 1.0]
    assert m.nsteps == 2
    with pytest.raises(ValueError):
        parse_inp(deck.replace("qua 1 2 3 4 1 0", "oct 1 2 3 4 1 0"))
    with pytest.raises(ValueError):
        parse_inp("implicit 2 10\n1 4 1 0 1 0 0 0\n1 1\n")   # nceqs != 0
    with pytest.raises(ValueError):
        parse_inp("implicit 2 10\n1 4 1 0 0 0 0 0\n1 1\nqua 1 2 3 4 1 0\n")  # truncated


def test_generator_sizes_match_survey():
    """C2 = 707^2 quad4, C3 = 149^3 hex8 (checked by formula, tiny instances built)."""
    n = 149
    assert (n + 1) ** 3 == 3375000 and 3 * (n + 1) ** 3 == 10125000 and n ** 3 == 3307949
    assert 9 * (3 * n + 1) ** 3 == 809238528
    assert 4 * (3 * 707 + 1) ** 2 == 18011536
    m = structured_hex(3, 2, 2)
    assert (m.nels, m.nnds) == (12, 36)
    assert list(m.element_nodes(0)) == [0, 1, 5, 4, 12, 13, 17, 16]        # bottom CCW then top CCW
    assert m.bcnode.shape[0] == 2 * 12
    q = structured_quad(3, 2)
    assert list(q.element_nodes(0)) == [0, 1, 5, 4] and q.fnode.tolist() == [3, 7, 11]
    t = hex_to_tets(structured_hex(2, 2, 2, jitter=0.1))
    c = t.coords[t.conn.reshape(-1, 4)]
    vol = np.einsum("ij,ij->i", np.cross(c[:, 1] - c[:, 0], c[:, 2] - c[:, 0]), c[:, 3] - c[:, 0]) / 6
    assert (vol > 0).all() and abs(vol.sum() - 8.0) < 0.5
    mix = quad_to_mixed(structured_quad(4, 3))
    assert set(mix.eltype.tolist()) == {TRI, QUA}


def test_jitter_is_deterministic_and_partition_invariant():
    g = structured_hex(4, 4, 4, jitter=0.2)
    grid = (2, 2, 1)
    for r in range(4):
        b = partition.brick_mesh(r, grid, (2, 2, 4), jitter=0.2)
        gid = partition.global_node_ids(r, grid, (2, 2, 4))
        assert np.array_equal(b.coords, g.coords[gid])


@pytest.mark.parametrize("nranks", [1, 2, 4, 8, 6])
def test_brick_halo_plans_are_symmetric(nranks):
    grid = partition.brick_grid(nranks)
    assert grid[0] * grid[1] * grid[2] == nranks
    npr = (3, 2, 2)
    plans = [partition.brick_halo(r, grid, npr) for r in range(nranks)]
    gids = [partition.global_node_ids(r, grid, npr) for r in range(nranks)]
    for r, (nr, ptr, nodes) in enumerate(plans):
        assert list(nr) == sorted(nr) and r not in nr
        for i, q in enumerate(nr):
            mine = gids[r][nodes[ptr[i]:ptr[i + 1]]]
            qnr, qptr, qnodes = plans[q]
            j = list(qnr).index(r)
            theirs = gids[q][qnodes[qptr[j]:qptr[j + 1]]]
            assert np.array_equal(mine, theirs)              # same nodes, same order, both sides
            assert np.array_equal(mine, np.intersect1d(gids[r], gids[q]))
        # the analytic plan equals the generic one built from the local->global maps
        g2 = partition.general_halo(r, gids)
        assert np.array_equal(nr, g2[0]) and np.array_equal(ptr, g2[1]) and np.array_equal(nodes, g2[2])


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads (no GPU needed for that) and exports exactly what include/fem_amd.h declares."""
    from fem_amd import capi
    hdr = open(os.path.join(ROOT, "include", "fem_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fem_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 50
    for name in declared:
        assert hasattr(capi._lib, name), f"{name} declared in include/fem_amd.h but not exported"
    assert declared == set(capi.EXPORTED), declared ^ set(capi.EXPORTED)
    assert capi._lib.fem_version() == 100


def test_library_fails_loudly_without_gpu():
    from fem_amd import capi
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(capi.FemError, match="no HIP device"):
        capi.Context(0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under fem_amd/ or include/ may reference it."""
    bad = []
    for base in ("fem_amd", "include", "fortran"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".F90", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"\boracle[/.]|liboracle|defmod_oracle|libdefmod_ref", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


@pytest.mark.parametrize("nparts", [2, 3, 5])
def test_rcb_partition_and_split(nparts):
    m = quad_to_mixed(structured_quad(9, 6, jitter=0.1))
    m.telsd = np.array([[3, 1], [40, 2]], dtype=np.int32)
    m.tval = np.array([[1.0, 0.0, 0.0, 1.0], [0.0, 2.0, 0.0, 1.0]])
    epart = partition.rcb_partition(m, nparts)
    counts = np.bincount(epart, minlength=nparts)
    assert counts.sum() == m.nels and counts.max() - counts.min() <= nparts
    parts = [partition.split_mesh(m, epart, r) for r in range(nparts)]
    assert sum(p[0].nels for p in parts) == m.nels
    assert sum(p[0].telsd.shape[0] for p in parts) == 2          # each traction follows its element
    for lm, nl2g in parts:
        assert np.array_equal(nl2g, np.sort(nl2g))                # ascending global ids (main.F90:78-96)
        assert np.array_equal(lm.coords, m.coords[nl2g])
        for e in range(lm.nels):                                  # connectivity survives the renumbering
            assert lm.coords[lm.element_nodes(e)].shape[1] == 2
        if lm.bcnode.shape[0]:
            assert set(nl2g[lm.bcnode[:, 0]]) <= set(m.bcnode[:, 0])
    nl2g = [p[1] for p in parts]
    seen = np.zeros(m.nnds, dtype=int)
    for r in range(nparts):
        nr, ptr, nodes = partition.general_halo(r, nl2g)
        seen[nl2g[r]] += 1
        for i, q in enumerate(nr):
            qnr, qptr, qnodes = partition.general_halo(q, nl2g)
            j = list(qnr).index(r)
            assert np.array_equal(nl2g[r][nodes[ptr[i]:ptr[i + 1]]], nl2g[q][qnodes[qptr[j]:qptr[j + 1]]])
    assert (seen >= 1).all()


def test_product_element_tables_match_reference_goldens():
    """fem_el_counts / fem_el_tables are host-side (no GPU needed): the constants the kernels read
    from __constant__ memory, pinned against the reference-generated goldens."""
    import json
    from fem_amd import capi
    gold = json.load(open(os.path.join(HERE, "golden", "element_golden.json")))["types"]
    for name, t in (("tri", 1), ("qua", 2), ("tet", 3), ("hex", 4)):
        g = gold[name]
        assert capi.el_counts(t) == (g["nodecount"], g["nip"], g["dim"], g["nps"], g["vtkid"])
        ip, w, shp, inv = capi.el_tables(t)
        assert np.abs(ip - np.array(g["ipoints"])).max() < 1e-15
        assert np.abs(w - np.array(g["weights"])).max() < 1e-15
        assert np.abs(shp - np.array(g["shape"])).max() < 1e-15
        if g["n2inv"] is None:
            assert inv is None
        else:
            assert np.abs(inv - np.array(g["n2inv"])).max() < 1e-13
    assert capi.el_counts(5) == (4, 2, 2, 2, 9)                       # coh
    assert list(capi.form_el_indx(3, [4, 9, 2])) == [12, 13, 14, 27, 28, 29, 6, 7, 8]


def test_capi_error_codes_without_gpu():
    from fem_amd import capi
    with pytest.raises(capi.FemError, match="unknown element type"):
        capi.el_counts(9)
    import ctypes as C
    out = (C.c_int * 4)()
    assert capi._lib.fem_form_el_indx(4, 1, out, out) != 0            # pdim must be 2 or 3
    assert b"pdim" in capi._lib.fem_last_error()


def test_header_is_plain_c(tmp_path):
    """include/fem_amd.h compiles as strict C99 and the plain-C client links against the library."""
    import subprocess
    obj = tmp_path / "abi_smoke.o"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(obj)])
    exe = tmp_path / "abi_smoke"
    subprocess.check_call(["gcc", str(obj), "-o", str(exe), "-L", os.path.join(ROOT, "fem_amd", "lib"), "-lfem_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "fem_amd", "lib")])
    from fem_amd import capi
    if capi.device_count() == 0:
        p = subprocess.run([str(exe)], capture_output=True, text=True)
        assert p.returncode != 0 and "no HIP device" in p.stderr       # loud failure, no fallback


def test_rcm_order_is_a_bandwidth_reducing_permutation():
    """Host part of the direct solve (-pc_type lu): reverse Cuthill-McKee over a scrambled grid
    graph with two components; the band must shrink to about the short grid edge."""
    from fem_amd import capi
    nx, ny = 30, 7
    rng = np.random.default_rng(0)
    n = 2 * nx * ny
    ids = rng.permutation(n).reshape(2, nx, ny)
    nbrs = [[] for _ in range(n)]
    for c in range(2):
        for i in range(nx):
            for j in range(ny):
                for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                    a, b = i + di, j + dj
                    if 0 <= a < nx and 0 <= b < ny:
                        nbrs[ids[c, i, j]].append(ids[c, a, b])
    ptr = np.cumsum([0] + [len(x) for x in nbrs])
    adj = np.concatenate(nbrs)
    perm = capi.rcm_order(ptr, adj)
    assert sorted(perm) == list(range(n))
    band = max(abs(int(perm[v]) - int(perm[u])) for v in range(n) for u in nbrs[v])
    assert band <= ny + 2
    # a graph without edges keeps every node
    assert sorted(capi.rcm_order([0, 0, 0, 0], [])) == [0, 1, 2]


def test_fortran_interface_module_covers_the_whole_c_abi():
    """fortran/fem_amd_capi.F90 must bind every entry point include/fem_amd.h declares (a Fortran host
    is the reference's own language, north_star) -- keeps the two files in step."""
    import re
    hdr = open(os.path.join(ROOT, "include", "fem_amd.h")).read()
    f90 = open(os.path.join(ROOT, "fortran", "fem_amd_capi.F90")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*\*?\s*(fem_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    bound = set(re.findall(r'name="(fem_[a-z0-9_]+)"', f90))
    assert len(declared) > 60
    assert declared - bound == set(), sorted(declared - bound)
    assert bound - declared == set(), sorted(bound - declared)


def test_element_module_builds_under_the_reference_module_name(tmp_path):
    """fortran/m_local_amd.F90 compiled with -DFEM_AMD_AS_LOCAL is a module called `local`: a reference-side file
    that says `use local` and calls the element routines with the reference's signatures compiles against it
    unchanged (compile only: running needs the GPU)."""
    import shutil
    import subprocess
    flang = "/opt/rocm/lib/llvm/bin/flang"
    if not os.path.exists(flang):
        pytest.skip("no flang in this image")
    src = tmp_path / "caller.F90"
    src.write_text("""subroutine caller(k, estress, B, D, detj)
  use local
  implicit none
  real(8) :: k(8, 8), estress(4, 3), B(3, 8), D(3, 3), detj, ec(4, 2), dN(2, 4), u(8)
  real(8), pointer :: N(:)
  integer :: indx(8)
  pdim = 2
  call Initialize(ipoints, weights)
  ec = reshape((/0.d0, 1.d0, 1.d0, 0.d0, 0.d0, 0.d0, 1.d0, 1.d0/), (/4, 2/)); u = 0.d0
  call FormElKE("qua", ec, 3640.d0, 0.3d0, 1.d0, k)
  call CalcElStress("qua", ec, u, 3640.d0, 0.3d0, estress)
  call FormdNdetJ("qua", ipoints(getElTypeNo("qua"))%array(1, :), ec, dN, detj)
  call BMat(4, dN, B)
  call DMat(D, 3640.d0, 0.3d0)
  call ShapeFunc("qua", 1, N)
  call ShapeFuncd("qua", dN, (/0.d0, 0.d0/))
  call FormElIndx((/1, 2, 3, 4/), indx)
end subroutine caller
""")
    fdir = os.path.join(ROOT, "fortran")
    for f in ("fem_amd_capi.F90", "m_local_amd.F90"):
        subprocess.check_call([flang, "-c", "-DFEM_AMD_AS_LOCAL", "-module-dir", str(tmp_path), "-I", str(tmp_path),
                               os.path.join(fdir, f), "-o", str(tmp_path / (f + ".o"))])
    subprocess.check_call([flang, "-c", "-module-dir", str(tmp_path), "-I", str(tmp_path), str(src), "-o", str(tmp_path / "caller.o")])
    assert (tmp_path / "local.mod").exists()


def test_deck_writer_round_trips(tmp_path):
    """fem_amd.inp.write_inp -> read_inp: the reference's deck format carries every array of a Mesh unchanged (hex, quad,
    mixed and tet decks with several materials, forces with time windows, traction sides, partial constraints)."""
    import numpy as np
    from fem_amd.inp import read_inp, write_inp
    from tests.test_gpu_random_decks import random_deck
    for seed in range(8):
        m = random_deck(seed)
        path = str(tmp_path / f"d{seed}.inp")
        write_inp(path, m)
        r = read_inp(path)
        for a in ("eltype", "eptr", "conn", "mat_id", "coords", "mats", "bcnode", "bcval", "fnode", "telsd", "tval"):
            assert np.array_equal(getattr(m, a), getattr(r, a)), (seed, a)
        fval = m.fval.copy()
        fval[:, m.pdim:] = np.minimum(fval[:, m.pdim:], m.t)     # the reader clamps t1, t2 to t (src/m_io.F90:357-358)
        assert np.array_equal(fval, r.fval), seed
        assert (r.t, r.dt, r.pdim) == (m.t, m.dt, m.pdim)


def test_bench_work_table_matches_the_compiled_kernels():
    """bench.py prices the element-by-element kernels with the fp64 work per element visit AS COMPILED (EBE_WORK): the table
    has to follow the code.  tools/count_ebe_flops.py compiles the device code for gfx950 (no GPU needed) and counts."""
    import re
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "count_ebe_flops.py")], cwd=root, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = {}
    for line in out.stdout.splitlines():
        m = re.match(r"^(k_ebe_\w+<[^>]+>)[^:]*: .*fp64 instr (\d+), flops (\d+), all vector instr (\d+)", line)
        if m:
            got[m.group(1)] = tuple(int(v) for v in m.group(2, 3, 4))
    sys.path.insert(0, root)
    import bench
    for name in ("k_ebe_tile_x<cartesian>", "k_ebe_tile_x<cartesian, one record>", "k_ebe_tile_x4<cartesian>", "k_ebe_tile<cartesian>", "k_ebe_tile<affine>", "k_ebe_apply<3,8,8,3>"):
        assert got[name] == tuple(bench.EBE_WORK[name]), (name, got[name], bench.EBE_WORK[name])
    # distorted elements: the arithmetic of the general path alone (the tool's count of k_ebe_tile<general> includes the
    # parallelepiped shortcut that a wavefront of distorted elements never takes)
    assert got["k_ebe_apply<3,8,8,2>"][:2] == tuple(bench.EBE_WORK["k_ebe_tile<general>"][:2])


def test_connected_tiling_of_the_cohesive_deck_shares_its_edge_nodes():
    """Config 5 as one strip: `copies` copies of examples/cohesive_test_2.inp joined at their x = 0 / x = 1 edges (four nodes
    each, the duplicated interface nodes pairing up in the order of their numbers): 4 + 4 copies nodes, every cohesive element
    still spans its own interface pair, and the oracle's Newton loop runs on it."""
    from fem_amd.inp import read_inp
    from fem_amd.mesh import tile_deck, tile_deck_connected
    deck = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decks", "cohesive_test_2.inp")
    m1 = read_inp(deck)
    for copies in (1, 2, 7):
        m = tile_deck_connected(m1, copies)
        assert m.nnds == 4 + 4 * copies and m.nels == 3 * copies
        assert m.conn.min() == 0 and m.conn.max() == m.nnds - 1
        d = tile_deck(m1, copies)                     # the disjoint tiling: same elements, copy c shifted by 2 (extent 1 + 1)
        conn_c, conn_d = m.conn.reshape(-1, 4), d.conn.reshape(-1, 4)
        for e in range(m.nels):
            c = e // 3
            assert np.allclose(m.coords[conn_c[e]], d.coords[conn_d[e]] - np.array([c * 1.0, 0.0]))
        if copies > 1:   # the right edge of copy 0 is the left edge of copy 1
            e0, e1 = conn_c[0], conn_c[3]             # upper quads of copies 0 and 1: (1 2 5 4) -> nodes 4, 5 are copy 1's 1, 2
            assert e0[3] == e1[0] and e0[2] == e1[1]
    from oracle.oracle import Oracle, build
    build(ref=False)
    r = Oracle().snes_run(tile_deck_connected(m1, 5), rels_mode=1)
    assert r["its"][0] == 1 and r["reasons"][0] == 3 and np.isfinite(r["u"]).all()


def test_bench_helpers_that_only_an_eight_gpu_run_reaches():
    """bench.py's N = 8 line carries `config.same_problem_on_one_gpu`; no run of this repository's own can reach that branch (the
    rehearsals stop at four ranks: the box allows six processes on the card), so the helper is exercised here, with the line's
    JSON serialisation."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    d = b.same_problem_on_one_gpu(1.0e8)
    assert abs(d["speedup_cg_jacobi"] - 1.0e8 / 25.94e6) < 1e-12 and d["cg_jacobi"]["cg_iterations"] == 1413
    json.dumps(d)
    assert b.spmv_algorithmic_bytes(809238528, 10125000) == 809238528 * 12.0 + 10125001 * 4.0 + 10125000 * 16.0
    hc = b.host_cores()
    assert hc["use"] >= 1 and hc["logical"] >= hc["use"]
