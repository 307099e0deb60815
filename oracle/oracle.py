"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes loaders for the two checkers:

* ``Oracle``  -> oracle/liboracle.so, the plain-C restatement of the reference
  hot path (oracle/defmod_oracle.c).
* ``RefLib``  -> oracle/_ref/libdefmod_ref.so, the reference's own element
  modules compiled from /root/reference by oracle/Makefile (present only where
  that build has run; the prebuilt .so travels to the GPU box).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (fem_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

TRI, QUA, TET, HEX, COH = 1, 2, 3, 4, 5
TYPE_NAMES = {"tri": TRI, "qua": QUA, "tet": TET, "hex": HEX, "coh": COH}
PENALTY_PARAM = float(np.float32(1.0e30))
DEFAULT_RTOL = float(np.float32(1.0e-9))


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def build(ref: bool = True) -> None:
    """Compile the C restatement (and oracle/_ref when the reference is present)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class Problem(C.Structure):
    _fields_ = [
        ("pdim", C.c_int), ("nels", C.c_int), ("nnds", C.c_int),
        ("eltype", _ip), ("eptr", _ip), ("conn", _ip), ("mat_id", _ip),
        ("coords", _dp), ("nmats", C.c_int), ("mats", _dp),
        ("nbcs", C.c_int), ("bcnode", _ip), ("bcval", _dp),
        ("nfrcs", C.c_int), ("fnode", _ip), ("fval", _dp),
        ("ntrcs", C.c_int), ("telsd", _ip), ("tval", _dp),
        ("t", C.c_double), ("dt", C.c_double),
    ]


def make_problem(m) -> tuple[Problem, list]:
    """Build the C struct from a mesh-like object/dict with numpy fields:
    pdim, eltype, eptr, conn, mat_id, coords, mats, bcnode, bcval, fnode, fval,
    telsd, tval, t, dt (0-based node/element ids)."""
    g = (lambda k: m[k]) if isinstance(m, dict) else (lambda k: getattr(m, k))
    pd = int(g("pdim"))
    keep = []

    def arr(x, dt, shape=None):
        a = np.ascontiguousarray(np.asarray(x, dtype=dt))
        if shape is not None:
            a = a.reshape(shape)
        keep.append(a)
        return a

    eltype = arr(g("eltype"), np.int32)
    eptr = arr(g("eptr"), np.int32)
    conn = arr(g("conn"), np.int32)
    mat_id = arr(g("mat_id"), np.int32)
    coords = arr(g("coords"), np.float64)
    mats = arr(g("mats"), np.float64).reshape(-1, 5)
    bcnode = arr(g("bcnode"), np.int32).reshape(-1, 1 + pd)
    bcval = arr(g("bcval"), np.float64).reshape(-1, pd)
    fnode = arr(g("fnode"), np.int32).reshape(-1)
    fval = arr(g("fval"), np.float64).reshape(-1, pd + 2)
    telsd = arr(g("telsd"), np.int32).reshape(-1, 2)
    tval = arr(g("tval"), np.float64).reshape(-1, pd + 2)
    keep += [mats, bcnode, bcval, fnode, fval, telsd, tval]
    p = Problem(pd, len(eltype), coords.size // pd, _i(eltype), _i(eptr), _i(conn), _i(mat_id),
                _d(coords), mats.shape[0], _d(mats), bcnode.shape[0], _i(bcnode), _d(bcval),
                fnode.shape[0], _i(fnode), _d(fval), telsd.shape[0], _i(telsd), _d(tval),
                float(g("t")), float(g("dt")))
    return p, keep


class Oracle:
    def __init__(self, path: str | None = None):
        path = path or os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.L = C.CDLL(path)
        L.dm_mat_det.restype = C.c_double
        L.dm_csr_pattern.restype = C.c_longlong
        for f in ("dm_node_count", "dm_nip", "dm_dim", "dm_nps", "dm_vtkid"):
            getattr(L, f).restype = C.c_int

    # ---- element level ----
    def counts(self, t):
        L = self.L
        return L.dm_node_count(t), L.dm_nip(t), L.dm_dim(t), L.dm_nps(t), L.dm_vtkid(t)

    def tables(self, t, pdim):
        n, nip, dim, _, _ = self.counts(t)
        ip = np.zeros(nip * dim); w = np.zeros(nip)
        self.L.dm_sam_pts(t, _d(ip), _d(w))
        ipm = ip.reshape(dim, nip).T.copy()  # (nip, pdim)
        shp = np.zeros((nip, n))
        for g in range(nip):
            N = np.zeros(n)
            xi = np.ascontiguousarray(ipm[g])
            self.L.dm_shape_func(t, _d(xi), _d(N))
            shp[g] = N
        inv = np.zeros(nip * n)
        has = self.L.dm_nodal_stress_inv(t, pdim, _d(inv))
        return ipm, w, shp, (inv.reshape(n, nip).T.copy() if has else None)

    def form_el_ke(self, t, ecoords, E, nu):
        """ecoords (n, pdim) -> k (n*pdim, n*pdim)."""
        n, _, dim, _, _ = self.counts(t)
        ec = np.asfortranarray(np.asarray(ecoords, dtype=np.float64).reshape(n, dim)).ravel(order="F").copy()
        k = np.zeros((n * dim) ** 2)
        self.L.dm_form_el_ke(t, _d(ec), C.c_double(E), C.c_double(nu), _d(k))
        return k.reshape(n * dim, n * dim, order="F")

    def calc_el_stress(self, t, ecoords, edisp, E, nu):
        n, nip, dim, _, _ = self.counts(t)
        cp = 3 if dim == 2 else 6
        ec = np.asarray(ecoords, dtype=np.float64).reshape(n, dim).ravel(order="F").copy()
        ed = np.ascontiguousarray(edisp, dtype=np.float64)
        es = np.zeros(nip * cp)
        self.L.dm_calc_el_stress(t, _d(ec), _d(ed), C.c_double(E), C.c_double(nu), _d(es))
        return es.reshape(nip, cp, order="F")

    def form_dn_detj(self, t, xi, ecoords):
        n, _, dim, _, _ = self.counts(t)
        ec = np.asarray(ecoords, dtype=np.float64).reshape(n, dim).ravel(order="F").copy()
        x = np.ascontiguousarray(xi, dtype=np.float64)
        dN = np.zeros(dim * n); dj = C.c_double(0)
        self.L.dm_form_dn_detj(t, _d(x), _d(ec), _d(dN), C.byref(dj))
        return dN.reshape(dim, n, order="F"), dj.value

    def bmat(self, pdim, n, dN):
        """BMat: dN(pdim, n) -> B(cpdim, n*pdim)."""
        cp = 3 if pdim == 2 else 6
        d = np.asfortranarray(np.asarray(dN, dtype=np.float64).reshape(pdim, n))
        B = np.zeros((cp, n * pdim), order="F")
        self.L.dm_bmat(int(pdim), int(n), d.ctypes.data_as(_dp), B.ctypes.data_as(_dp))
        return np.array(B)

    def dmat(self, pdim, E, nu):
        cp = 3 if pdim == 2 else 6
        D = np.zeros(cp * cp)
        self.L.dm_dmat(pdim, C.c_double(E), C.c_double(nu), _d(D))
        return D.reshape(cp, cp, order="F")

    def edge_area_nodes(self, t, enodes, ecoords, side):
        n, _, dim, nps, _ = self.counts(t)
        ec = np.asarray(ecoords, dtype=np.float64).reshape(n, dim).ravel(order="F").copy()
        en = np.ascontiguousarray(enodes, dtype=np.int32)
        sn = np.zeros(nps, dtype=np.int32); a = C.c_double(0)
        self.L.dm_edge_area_nodes(t, _i(en), _d(ec), side, C.byref(a), _i(sn))
        return a.value, sn

    # ---- global level ----
    def csr_pattern(self, m, nnz=None):
        """nnz=None: a counting pass first; pass a known nnz to run the fill pass only."""
        p, keep = make_problem(m)
        N = p.nnds * p.pdim
        rowptr = np.zeros(N + 1, dtype=np.int32)
        if nnz is None:
            nnz = self.L.dm_csr_pattern(C.byref(p), _i(rowptr), None)
        colind = np.zeros(max(nnz, 1), dtype=np.int32)
        self.L.dm_csr_pattern(C.byref(p), _i(rowptr), _i(colind))
        return rowptr, colind[:nnz]

    def assemble(self, m, rowptr, colind, nthreads=1, apply_bc=True):
        p, keep = make_problem(m)
        val = np.zeros(max(len(colind), 1))
        self.L.dm_assemble(C.byref(p), _i(rowptr), _i(colind), _d(val), nthreads)
        if apply_bc:
            self.L.dm_apply_kbc(C.byref(p), _i(rowptr), _i(colind), _d(val))
        return val[:len(colind)]

    def form_rhs(self, m, t_init, dt):
        p, keep = make_problem(m)
        F = np.zeros(p.nnds * p.pdim)
        self.L.dm_form_rhs(C.byref(p), C.c_double(t_init), C.c_double(dt), _d(F))
        return F

    def spmv(self, rowptr, colind, val, x, nthreads=1):
        y = np.zeros(len(rowptr) - 1)
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.L.dm_spmv(len(y), _i(rowptr), _i(colind), _d(val), _d(x), _d(y), nthreads)
        return y

    def ksp_cg_jacobi(self, rowptr, colind, val, b, rtol=DEFAULT_RTOL, abstol=1e-50, dtol=1e5,
                      maxit=10000, nthreads=1, hist_len=0):
        n = len(rowptr) - 1
        x = np.zeros(n); its = C.c_int(0); rn = C.c_double(0)
        hist = np.zeros(max(hist_len, 1))
        b = np.ascontiguousarray(b, dtype=np.float64)
        reason = self.L.dm_ksp_cg_jacobi(n, _i(rowptr), _i(colind), _d(val), _d(b), _d(x),
                                         C.c_double(rtol), C.c_double(abstol), C.c_double(dtol), maxit,
                                         C.byref(its), C.byref(rn), _d(hist) if hist_len else None,
                                         hist_len, nthreads)
        return x, its.value, reason, rn.value, hist[:min(hist_len, its.value + 1)]

    def dense_lu_solve(self, rowptr, colind, val, b):
        n = len(rowptr) - 1
        x = np.zeros(n)
        b = np.ascontiguousarray(b, dtype=np.float64)
        rc = self.L.dm_dense_lu_solve(n, _i(rowptr), _i(colind), _d(val), _d(b), _d(x))
        if rc:
            raise RuntimeError(f"singular matrix at pivot {rc}")
        return x

    def recover_nodal_stress(self, m, uu):
        p, keep = make_problem(m)
        cp = 3 if p.pdim == 2 else 6
        sn = np.zeros(p.nnds * cp); cnt = np.zeros(p.nnds, dtype=np.int32)
        uu = np.ascontiguousarray(uu, dtype=np.float64)
        self.L.dm_recover_nodal_stress(C.byref(p), _d(uu), _d(sn), _i(cnt))
        return sn.reshape(p.nnds, cp, order="F"), cnt

    # ---- cohesive element, element level ----
    def coh_values(self, ecoords):
        ec = np.asfortranarray(np.asarray(ecoords, dtype=np.float64).reshape(4, 2))
        t, n = np.zeros(2), np.zeros(2); det = C.c_double(0)
        self.L.dm_coh_values(_d(ec), _d(t), _d(n), C.byref(det))
        return t, n, det.value

    def coh_rels(self, u8, gp, dt):
        u = np.ascontiguousarray(u8, dtype=np.float64); ur, vr = np.zeros(2), np.zeros(2)
        self.L.dm_coh_rels(_d(u), int(gp), C.c_double(dt), _d(ur), _d(vr))
        return ur, vr

    def coh_gaps(self, t, n, urel, vrel):
        g, vg = np.zeros(2), np.zeros(2)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (t, n, urel, vrel)]
        self.L.dm_coh_gaps(_d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(g), _d(vg))
        return g, vg

    def seplaw1(self, props, gap, vgap, dtime):
        p, g, vg = (np.ascontiguousarray(x, dtype=np.float64) for x in (props, gap, vgap))
        tr, st = np.zeros(2), np.zeros(4)
        self.L.dm_seplaw1_tract(_d(p), _d(g), _d(vg), C.c_double(dtime), _d(tr))
        self.L.dm_seplaw1_stiff(_d(p), _d(g), _d(vg), C.c_double(dtime), _d(st))
        return tr, st.reshape(2, 2, order="F")

    def coh_apply_1(self, ecoords, u8, props, dt):
        ec = np.asfortranarray(np.asarray(ecoords, dtype=np.float64).reshape(4, 2))
        u, p = (np.ascontiguousarray(x, dtype=np.float64) for x in (u8, props))
        tr, st = np.zeros(8), np.zeros(64)
        self.L.dm_coh_apply_tract_1(_d(ec), _d(u), _d(p), C.c_double(dt), _d(tr))
        self.L.dm_coh_apply_stiff_1(_d(ec), _d(u), _d(p), C.c_double(dt), _d(st))
        return tr, st.reshape(8, 8, order="F")

    def run(self, m, ksp_type="cg", rtol=DEFAULT_RTOL, abstol=1e-50, dtol=1e5, maxit=10000, nthreads=1):
        p, keep = make_problem(m)
        cp = 3 if p.pdim == 2 else 6
        u = np.zeros(p.nnds * p.pdim); s = np.zeros(p.nnds * cp)
        its = np.zeros(4096, dtype=np.int32)
        nsteps = self.L.dm_run(C.byref(p), 0 if ksp_type == "cg" else 1, C.c_double(rtol),
                               C.c_double(abstol), C.c_double(dtol), maxit, _d(u), _d(s), _i(its),
                               len(its), nthreads)
        if nsteps < 0:
            raise RuntimeError(f"oracle solve failed: {nsteps}")
        return u, s.reshape(p.nnds, cp, order="F"), its[:nsteps]

    # ---- nonlinear increment around the cohesive element (oracle/snes_oracle.c) ----
    class Cohesive(C.Structure):
        _fields_ = [("nlmat", _ip), ("ncohmats", C.c_int), ("props", _dp), ("rels_mode", C.c_int),
                    ("rels_nodes", C.c_int * 4), ("bc_legacy", C.c_int), ("variant", C.c_int)]

    def _cohesive(self, m, rels_mode=0, rels_nodes=(0, 1, 2, 3), bc_legacy=False, variant=0):
        nlmat = np.ascontiguousarray(m.nlmat if m.nlmat is not None else np.zeros(m.nels), dtype=np.int32)
        props = np.ascontiguousarray([pr for _, pr in m.cohmats], dtype=np.float64).reshape(-1, 6)
        c = self.Cohesive(_i(nlmat), props.shape[0], _d(props), int(rels_mode), (C.c_int * 4)(*[int(v) for v in rels_nodes]),
                          int(bool(bc_legacy)), int(variant))
        return c, [nlmat, props]

    def snes_run(self, m, rels_mode=0, rels_nodes=(0, 1, 2, 3), bc_legacy=False, inner="lu", ksp_rtol=1e-12, ksp_abstol=1e-50,
                 maxsteps=1000, variant=0):
        """main.F90:216-296 on a deck with cohesive elements.  Returns dict(u, stress, U_hist, its, reasons, fnorms).
        variant: 0 = the current sources; bits of dm_cohesive.variant for tools/snes_log_variants.py."""
        p, keep = make_problem(m)
        c, keep2 = self._cohesive(m, rels_mode, rels_nodes, bc_legacy, variant)
        N = p.nnds * p.pdim
        nst = min(int(m.nsteps), maxsteps)
        u = np.zeros(N); s = np.zeros(p.nnds * 3); hist = np.zeros((nst, N))
        its = np.zeros(nst, dtype=np.int32); reasons = np.zeros(nst, dtype=np.int32); fn = np.zeros(nst)
        self.L.dm_snes_run.restype = C.c_int
        n = self.L.dm_snes_run(C.byref(p), C.byref(c), 0 if inner == "lu" else 1, C.c_double(ksp_rtol), C.c_double(ksp_abstol), _d(u), _d(s),
                               _d(hist),
                               _i(its), _i(reasons), _d(fn), nst)
        if n < 0:
            raise RuntimeError(f"dm_snes_run failed: {n}")
        return {"u": u, "stress": s.reshape(p.nnds, 3, order="F"), "U_hist": hist[:n], "its": its[:n], "reasons": reasons[:n],
                "fnorms": fn[:n]}

    def snes_callbacks(self, m, rowptr, colind, Kbc, F, aggregate_u, du, rels_mode=0, rels_nodes=(0, 1, 2, 3), bc_legacy=False):
        p, keep = make_problem(m)
        c, keep2 = self._cohesive(m, rels_mode, rels_nodes, bc_legacy)
        J = np.zeros(len(colind)); R = np.zeros(p.nnds * p.pdim)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (Kbc, F, aggregate_u, du)]
        self.L.dm_snes_callbacks(C.byref(p), C.byref(c), _i(rowptr), _i(colind), _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(J), _d(R))
        return J, R

    # ---- bench.py's cpu_baseline leg (oracle/cpu_baseline.c) ----
    def cpu_baseline(self, m, nthreads, ncg, rtol=DEFAULT_RTOL, nprod=0, want_arrays=False, want_x=False):
        """Timed all-core run of the path; returns a dict of timings (and the matrix / iterate when asked; want_x: the
        iterate alone -- the full-size parity tests, where a second copy of the 10 GB matrix would be wasted)."""
        p, keep = make_problem(m)
        t = np.zeros(6); info = np.zeros(3, dtype=np.int64)
        N = p.nnds * p.pdim
        x = rp = ci = val = None
        if want_x and not want_arrays:
            x = np.zeros(N)
        if want_arrays:
            nnz = len(self.csr_pattern(m)[1])
            x = np.zeros(N); rp = np.zeros(N + 1, dtype=np.int32); ci = np.zeros(nnz, dtype=np.int32); val = np.zeros(nnz)
        self.L.dm_cpu_baseline.restype = C.c_int
        rc = self.L.dm_cpu_baseline(C.byref(p), int(nthreads), int(ncg), C.c_double(rtol), int(nprod), _d(t),
                                    info.ctypes.data_as(C.POINTER(C.c_longlong)),
                                    _d(x) if (want_arrays or want_x) else None, _i(rp) if want_arrays else None,
                                    _i(ci) if want_arrays else None, _d(val) if want_arrays else None)
        if rc != 0:
            raise RuntimeError(f"dm_cpu_baseline failed: {rc}")
        out = {"t_pattern": t[0], "t_assemble": t[1], "t_setup": t[2], "t_cg_iteration": t[3], "t_ebe_product": t[4],
               "t_spmv": t[5], "nnz": int(info[0]), "its": int(info[1]), "reason": int(info[2])}
        if want_arrays:
            out.update(x=x, rowptr=rp, colind=ci, val=val)
        elif want_x:
            out.update(x=x)
        return out

    def cpu_ebe_mult(self, m, x, nthreads=1):
        p, keep = make_problem(m)
        y = np.zeros(p.nnds * p.pdim)
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.L.dm_cpu_ebe_mult(C.byref(p), int(nthreads), _d(x), _d(y))
        return y


class RefLib:
    """The reference's own element routines (flang build of /root/reference)."""

    def __init__(self, path: str | None = None):
        path = path or os.path.join(HERE, "_ref", "libdefmod_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.L = C.CDLL(path)
        self.pdim = None

    @staticmethod
    def available() -> bool:
        return os.path.exists(os.path.join(HERE, "_ref", "libdefmod_ref.so"))

    def init(self, pdim):
        if self.pdim != pdim:
            self.L.ref_init(int(pdim))
            self.pdim = pdim

    def counts(self, t):
        v = [C.c_int(0) for _ in range(5)]
        self.L.ref_counts(t, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)

    def tables(self, t, pdim):
        self.init(pdim)
        n, nip, dim, _, _ = self.counts(t)
        ip = np.zeros(nip * pdim); w = np.zeros(nip); shp = np.zeros(nip * n); inv = np.zeros(nip * n)
        has = C.c_int(0)
        self.L.ref_tables(t, _d(ip), _d(w), _d(shp), _d(inv), C.byref(has))
        return (ip.reshape(nip, pdim, order="F"), w, shp.reshape(nip, n, order="F"),
                inv.reshape(nip, n, order="F") if has.value else None)

    def form_el_ke(self, t, ecoords, E, nu, pdim):
        self.init(pdim)
        n = self.counts(t)[0]
        ec = np.asarray(ecoords, dtype=np.float64).reshape(n, pdim).ravel(order="F").copy()
        k = np.zeros((n * pdim) ** 2)
        self.L.ref_form_el_ke(t, _d(ec), C.c_double(E), C.c_double(nu), _d(k))
        return k.reshape(n * pdim, n * pdim, order="F")

    def calc_el_stress(self, t, ecoords, edisp, E, nu, pdim):
        self.init(pdim)
        n, nip = self.counts(t)[:2]
        cp = 3 if pdim == 2 else 6
        ec = np.asarray(ecoords, dtype=np.float64).reshape(n, pdim).ravel(order="F").copy()
        ed = np.ascontiguousarray(edisp, dtype=np.float64)
        es = np.zeros(nip * cp)
        self.L.ref_calc_el_stress(t, _d(ec), _d(ed), C.c_double(E), C.c_double(nu), _d(es))
        return es.reshape(nip, cp, order="F")

    def form_dn_detj(self, t, xi, ecoords, pdim):
        self.init(pdim)
        n = self.counts(t)[0]
        ec = np.asarray(ecoords, dtype=np.float64).reshape(n, pdim).ravel(order="F").copy()
        x = np.ascontiguousarray(xi, dtype=np.float64)
        dN = np.zeros(pdim * n); dj = C.c_double(0)
        self.L.ref_form_dn_detj(t, _d(x), _d(ec), _d(dN), C.byref(dj))
        return dN.reshape(pdim, n, order="F"), dj.value

    def dmat(self, pdim, E, nu):
        self.init(pdim)
        cp = 3 if pdim == 2 else 6
        D = np.zeros(cp * cp)
        self.L.ref_dmat(C.c_double(E), C.c_double(nu), _d(D))
        return D.reshape(cp, cp, order="F")

    def form_el_indx(self, enodes, pdim):
        self.init(pdim)
        en = np.ascontiguousarray(enodes, dtype=np.int32)
        ix = np.zeros(len(en) * pdim, dtype=np.int32)
        self.L.ref_form_el_indx(len(en), _i(en), _i(ix))
        return ix

    def edge_area_nodes(self, t, enodes, ecoords, side, pdim):
        self.init(pdim)
        n, _, _, nps, _ = self.counts(t)
        ec = np.asarray(ecoords, dtype=np.float64).reshape(n, pdim).ravel(order="F").copy()
        en = np.ascontiguousarray(enodes, dtype=np.int32)
        sn = np.zeros(nps, dtype=np.int32); a = C.c_double(0)
        self.L.ref_edge_area_nodes(t, _i(en), _d(ec), side, C.byref(a), _i(sn))
        return a.value, sn

    # ---- cohesive helper routines of the reference (m_elems.F90:658-720, m_seplaw.F90) ----
    def coh_values(self, ecoords):
        self.init(2)
        ec = np.asfortranarray(np.asarray(ecoords, dtype=np.float64).reshape(4, 2))
        t, n = np.zeros(2), np.zeros(2); det = C.c_double(0)
        self.L.ref_coh_values(_d(ec), _d(t), _d(n), C.byref(det))
        return t, n, det.value

    def coh_rels(self, u8, gp, dt):
        self.init(2)
        u = np.ascontiguousarray(u8, dtype=np.float64); ur, vr = np.zeros(2), np.zeros(2)
        self.L.ref_coh_rels(_d(u), int(gp) + 1, C.c_double(dt), _d(ur), _d(vr))
        return ur, vr

    def coh_gaps(self, t, n, urel, vrel):
        self.init(2)
        g, vg = np.zeros(2), np.zeros(2)
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (t, n, urel, vrel)]
        self.L.ref_coh_gaps(_d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(g), _d(vg))
        return g, vg

    def seplaw1(self, props, gap, vgap, dtime):
        p, g, vg = (np.ascontiguousarray(x, dtype=np.float64) for x in (props, gap, vgap))
        tr, st = np.zeros(2), np.zeros(4)
        self.L.ref_seplaw1(_d(p), _d(g), _d(vg), C.c_double(dtime), _d(tr), _d(st))
        return tr, st.reshape(2, 2, order="F")
