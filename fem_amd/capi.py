"""ctypes binding of libfem_amd.so (include/fem_amd.h) and thin PETSc-shaped
wrappers (Mat / Vec / KSP objects with the method names the reference calls
from src/main.F90 and src/m_global.F90).  Plumbing only: every number is
computed by the HIP library.  Import fails loudly if the library is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfem_amd.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_h = C.c_void_p


class FemError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libfem_amd error {code}: {msg}")
        self.code = code


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP library is required; there is no CPU path)")
    return C.CDLL(LIB_PATH)


_lib = _load()
_lib.fem_last_error.restype = C.c_char_p

# (name, argtypes) for every symbol of include/fem_amd.h
_SIGS = {
    "fem_version": [],
    "fem_device_count": [_ip],
    "fem_ctx_create": [C.c_int, C.POINTER(_h)],
    "fem_ctx_destroy": [_h],
    "fem_ctx_synchronize": [_h],
    "fem_ctx_timer_start": [_h],
    "fem_ctx_timer_stop": [_h, _dp],
    "fem_el_counts": [C.c_int, _ip, _ip, _ip, _ip, _ip],
    "fem_el_tables": [C.c_int, _dp, _dp, _dp, _dp, _ip],
    "fem_form_el_ke": [_h, C.c_int, C.c_int, _dp, _dp, _dp, _dp],
    "fem_calc_el_stress": [_h, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp],
    "fem_form_el_indx": [C.c_int, C.c_int, _ip, _ip],
    "fem_shape_func": [C.c_int, C.c_int, _dp],
    "fem_shape_funcd": [_h, C.c_int, C.c_int, _dp, _dp],
    "fem_form_dn_detj": [_h, C.c_int, C.c_int, _dp, _dp, _dp, _dp],
    "fem_bmat": [_h, C.c_int, C.c_int, C.c_int, _dp, _dp],
    "fem_dmat": [_h, C.c_int, C.c_int, _dp, _dp, _dp],
    "fem_coh_apply_1": [_h, C.c_int, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp],
    "fem_rcm_order": [C.c_int, _ip, _ip, _ip],
    "fem_mesh_create": [_h, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _dp, C.c_int, _dp, C.POINTER(_h)],
    "fem_mesh_destroy": [_h],
    "fem_mesh_set_halo": [_h, _h, C.c_int, _ip, _ip, _ip],
    "fem_mat_create_aij": [_h, C.POINTER(_h)],
    "fem_mat_destroy": [_h],
    "fem_mat_zero_entries": [_h],
    "fem_mat_assemble_elements": [_h],
    "fem_mat_apply_kbc": [_h, C.c_int, _ip],
    "fem_mat_get_info": [_h, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)],
    "fem_mat_get_csr": [_h, _ip, _ip, _dp],
    "fem_mat_mult": [_h, _h, _h],
    "fem_mat_set_mult_type": [_h, C.c_char_p],
    "fem_mat_set_profiling": [_h, C.c_int],
    "fem_mat_get_profile": [_h, C.c_int, _ip, _dp, _dp, _dp],
    "fem_mat_get_mult_info": [_h, _ip, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                              C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), _dp],
    "fem_mat_get_tile_geometry_info": [_h, C.POINTER(C.c_longlong), _ip, _ip],
    "fem_mesh_tile_plan_digest": [_h, C.c_char_p, C.POINTER(C.c_ulonglong), _dp],
    "fem_mat_get_diagonal": [_h, _h],
    "fem_mat_set_values": [_h, C.c_int, _ip, C.c_int, _ip, _dp, C.c_int],
    "fem_mat_mult_timed": [_h, _h, _h, C.c_int, _dp],
    "fem_vec_create": [_h, C.c_longlong, C.POINTER(_h)],
    "fem_vec_destroy": [_h],
    "fem_vec_get_size": [_h, C.POINTER(C.c_longlong)],
    "fem_vec_set": [_h, C.c_double],
    "fem_vec_set_array": [_h, _dp],
    "fem_vec_get_array": [_h, _dp],
    "fem_vec_device_ptr": [_h, C.POINTER(C.c_void_p)],
    "fem_vec_copy": [_h, _h],
    "fem_vec_axpy": [_h, C.c_double, _h],
    "fem_vec_dot": [_h, _h, _dp],
    "fem_vec_norm2": [_h, _dp],
    "fem_form_rhs": [_h, C.c_int, _ip, _dp, C.c_int, _ip, _dp, C.c_int, _ip, _dp, C.c_double, C.c_double,
                     C.c_double, _h],
    "fem_ksp_create": [_h, C.POINTER(_h)],
    "fem_ksp_destroy": [_h],
    "fem_ksp_set_operators": [_h, _h],
    "fem_ksp_set_type": [_h, C.c_char_p],
    "fem_ksp_set_pc_type": [_h, C.c_char_p],
    "fem_ksp_set_tolerances": [_h, C.c_double, C.c_double, C.c_double, C.c_int],
    "fem_ksp_set_from_options": [_h, C.c_char_p],
    "fem_ksp_set_up": [_h],
    "fem_ksp_solve": [_h, _h, _h],
    "fem_ksp_get_iteration_number": [_h, _ip],
    "fem_ksp_get_residual_norm": [_h, _dp],
    "fem_ksp_get_converged_reason": [_h, _ip],
    "fem_ksp_get_residual_history": [_h, _dp, C.c_int, _ip],
    "fem_ksp_set_profiling": [_h, C.c_int],
    "fem_ksp_get_profile": [_h, _ip, _dp, _dp],
    "fem_ksp_get_reused_solves": [_h, _ip],
    "fem_ksp_get_resident_solves": [_h, _ip],
    "fem_recover_nodal_stress": [_h, _h, _dp],
    "fem_recover_nodal_stress_vec": [_h, _h, _h],
    "fem_mesh_set_cohesive": [_h, _ip, C.c_int, _dp],
    "fem_mat_duplicate": [_h, C.POINTER(_h)],
    "fem_mat_copy": [_h, _h],
    "fem_mat_add_cohesive": [_h, _h, C.c_double, C.c_int, _ip],
    "fem_vec_add_cohesive_traction": [_h, _h, C.c_double, C.c_int, _ip, _h],
    "fem_enforce_bc_force": [_h, C.c_int, _ip, _dp, C.c_double, C.c_double, _h],
    "fem_mat_mult_add": [_h, _h, _h, _h],
    "fem_vec_scale": [_h, C.c_double],
    "fem_snes_create": [_h, C.POINTER(_h)],
    "fem_snes_destroy": [_h],
    "fem_snes_get_ksp": [_h, C.POINTER(_h)],
    "fem_snes_set_function": [_h, _h, C.c_void_p, C.c_void_p],
    "fem_snes_set_jacobian": [_h, _h, _h, C.c_void_p, C.c_void_p],
    "fem_snes_set_defmod_callbacks": [_h, _h, _h, _h, _h, _h, _h, _h, C.c_int, _ip, _dp, C.c_double, C.c_double, C.c_int, _ip, C.POINTER(C.c_void_p)],
    "fem_snes_calc_defmod_jacobian": [C.c_void_p, _h],
    "fem_snes_free_defmod_callbacks": [C.c_void_p],
    "fem_snes_set_from_options": [_h, C.c_char_p],
    "fem_snes_solve": [_h, _h],
    "fem_snes_get_iteration_number": [_h, _ip],
    "fem_snes_get_converged_reason": [_h, _ip],
    "fem_snes_get_function_norm": [_h, _dp],
    "fem_snes_get_linear_solve_iterations": [_h, _ip],
    "fem_snes_get_number_function_evals": [_h, _ip],
    "fem_comm_create_self": [_h, C.POINTER(_h)],
    "fem_comm_get_unique_id": [C.c_char_p],
    "fem_comm_create_rccl": [_h, C.c_int, C.c_int, C.c_char_p, C.POINTER(_h)],
    "fem_comm_create_local_group": [C.c_int, C.POINTER(_h), C.POINTER(_h)],
    "fem_comm_create_callbacks": [_h, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_h)],
    "fem_comm_get_stats": [_h, C.POINTER(C.c_longlong)],
    "fem_comm_destroy": [_h],
    "fem_comm_rank": [_h, _ip, _ip],
    "fem_comm_exchange": [_h, C.c_int, _ip, C.POINTER(C.c_longlong), _h, _h],
    "fem_comm_exchange_allreduce": [_h, C.c_int, _ip, C.POINTER(C.c_longlong), _h, _h, _h, C.c_int],
}
EXPORTED = sorted(_SIGS) + ["fem_last_error"]
for _n, _a in _SIGS.items():
    try:
        _f = getattr(_lib, _n)
    except AttributeError:
        # (A/B runs against an older build of the library, tools/ab_libs.sh: FEM_AMD_ALLOW_OLDER_LIB=1; otherwise a missing
        # entry point is a broken build and fails loudly here)
        if os.environ.get("FEM_AMD_ALLOW_OLDER_LIB") == "1":
            continue
        raise
    _f.argtypes = _a
    _f.restype = C.c_int


def _ck(rc: int):
    if rc != 0:
        raise FemError(rc, (_lib.fem_last_error() or b"").decode())


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


def _i32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a.reshape(shape) if shape is not None else a


PENALTY_PARAM = float(np.float32(1.0e30))   # src/m_global.F90:15
REFERENCE_RTOL = float(np.float32(1.0e-9))  # tol = 1.e-9, src/main.F90:349
DEFAULT = -2.0


def device_count() -> int:
    n = C.c_int(0)
    _ck(_lib.fem_device_count(C.byref(n)))
    return n.value


class Context:
    """PetscInitialize / PetscFinalize (src/main.F90:26,333): device + stream."""

    def __init__(self, device: int = 0):
        self.h = _h()
        _ck(_lib.fem_ctx_create(device, C.byref(self.h)))
        self.device = device

    def synchronize(self):
        _ck(_lib.fem_ctx_synchronize(self.h))

    def timer_start(self):
        _ck(_lib.fem_ctx_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_double(0)
        _ck(_lib.fem_ctx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def destroy(self):
        if self.h:
            _ck(_lib.fem_ctx_destroy(self.h))
            self.h = _h()

    # ---- module `local` / `elems` batched element API ----
    def form_el_ke(self, eltype: int, ecoords, E, nu) -> np.ndarray:
        """FormElKE (src/m_local.F90:21) for a batch: ecoords [nel, n, pdim] -> k [nel, nd, nd]."""
        n, _, dim, _, _ = el_counts(eltype)
        ec = _f64(ecoords, (-1, n, dim))
        nel = ec.shape[0]
        ecf = np.ascontiguousarray(ec.transpose(0, 2, 1))  # per element column-major (n, pdim)
        Ea = np.ascontiguousarray(np.broadcast_to(np.asarray(E, dtype=np.float64), (nel,)))
        na = np.ascontiguousarray(np.broadcast_to(np.asarray(nu, dtype=np.float64), (nel,)))
        nd = n * dim
        k = np.zeros((nel, nd, nd))
        _ck(_lib.fem_form_el_ke(self.h, eltype, nel, _d(ecf), _d(Ea), _d(na), _d(k)))
        return np.ascontiguousarray(k.transpose(0, 2, 1))  # column-major blocks -> k[e][i][j]

    def calc_el_stress(self, eltype: int, ecoords, edisp, E, nu) -> np.ndarray:
        """CalcElStress (src/m_local.F90:93): -> estress [nel, nip, cpdim]."""
        n, nip, dim, _, _ = el_counts(eltype)
        cp = 3 if dim == 2 else 6
        ec = _f64(ecoords, (-1, n, dim))
        nel = ec.shape[0]
        ecf = np.ascontiguousarray(ec.transpose(0, 2, 1))
        ed = _f64(edisp, (nel, n * dim))
        Ea = np.ascontiguousarray(np.broadcast_to(np.asarray(E, dtype=np.float64), (nel,)))
        na = np.ascontiguousarray(np.broadcast_to(np.asarray(nu, dtype=np.float64), (nel,)))
        es = np.zeros((nel, cp, nip))
        _ck(_lib.fem_calc_el_stress(self.h, eltype, nel, _d(ecf), _d(ed), _d(Ea), _d(na), _d(es)))
        return np.ascontiguousarray(es.transpose(0, 2, 1))


def el_counts(eltype: int):
    v = [C.c_int(0) for _ in range(5)]
    _ck(_lib.fem_el_counts(eltype, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def el_tables(eltype: int):
    """Initialize (src/m_elems.F90:124): ipoints(nip,pdim), weights, shape(nip,n), n2inv(nip,n) or None."""
    n, nip, dim, _, _ = el_counts(eltype)
    ip = np.zeros(nip * dim); w = np.zeros(nip); shp = np.zeros(nip * n); inv = np.zeros(nip * n)
    has = C.c_int(0)
    _ck(_lib.fem_el_tables(eltype, _d(ip), _d(w), _d(shp), _d(inv), C.byref(has)))
    return (ip.reshape(nip, dim, order="F"), w, shp.reshape(nip, n, order="F"),
            inv.reshape(nip, n, order="F") if has.value else None)


def form_el_indx(pdim: int, enodes) -> np.ndarray:
    en = _i32(enodes)
    out = np.zeros(len(en) * pdim, dtype=np.int32)
    _ck(_lib.fem_form_el_indx(pdim, len(en), _i(en), _i(out)))
    return out


def coh_apply_1(ctx: "Context", ecoords, u8, props, dt: float):
    """Cohesive elements, element level (applyTract_1 / applyStiff_1 of the reference, batched): returns
    tract[ncoh, 8], stiff[ncoh, 8, 8] and the per-integration-point record gp[ncoh, 2, 14]."""
    ec = np.ascontiguousarray(ecoords, dtype=np.float64).reshape(-1, 8)
    u = np.ascontiguousarray(u8, dtype=np.float64).reshape(-1, 8)
    p = np.ascontiguousarray(props, dtype=np.float64).reshape(-1, 6)
    n = ec.shape[0]
    tr, st, gp = np.zeros((n, 8)), np.zeros((n, 64)), np.zeros((n, 2, 14))
    _ck(_lib.fem_coh_apply_1(ctx.h, n, _d(ec), _d(u), _d(p), float(dt), _d(tr), _d(st), _d(gp)))
    return tr, st.reshape(n, 8, 8).transpose(0, 2, 1), gp


def rcm_order(ptr, adj) -> np.ndarray:
    """Reverse Cuthill-McKee numbering of an undirected graph (CSR adjacency): perm[old] = new."""
    ptr, adj = _i32(ptr), _i32(adj)
    perm = np.zeros(len(ptr) - 1, dtype=np.int32)
    _ck(_lib.fem_rcm_order(len(ptr) - 1, _i(ptr), _i(adj), _i(perm)))
    return perm


class Comm:
    def __init__(self, h):
        self.h = h

    @staticmethod
    def self_(ctx: Context) -> "Comm":
        h = _h()
        _ck(_lib.fem_comm_create_self(ctx.h, C.byref(h)))
        return Comm(h)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _ck(_lib.fem_comm_get_unique_id(buf))
        return buf.raw

    @staticmethod
    def rccl(ctx: Context, nranks: int, rank: int, uid: bytes) -> "Comm":
        assert len(uid) == 128
        h = _h()
        _ck(_lib.fem_comm_create_rccl(ctx.h, nranks, rank, uid, C.byref(h)))
        return Comm(h)

    @staticmethod
    def local_group(ctxs: list[Context]) -> list["Comm"]:
        n = len(ctxs)
        ca = (_h * n)(*[c.h for c in ctxs])
        out = (_h * n)()
        _ck(_lib.fem_comm_create_local_group(n, ca, out))
        return [Comm(_h(out[i])) for i in range(n)]

    _AR = C.CFUNCTYPE(None, C.c_void_p, _dp, C.c_int)
    _EX = C.CFUNCTYPE(None, C.c_void_p, C.c_int, _ip, C.POINTER(C.c_longlong), _dp, _dp)

    @staticmethod
    def callbacks(ctx: Context, nranks: int, rank: int, allreduce, exchange) -> "Comm":
        """Host-staged communicator (fem_comm_create_callbacks): `allreduce(buf)` sums the float64 array `buf` over all ranks in
        place; `exchange(neigh_rank, off, send, recv)` sends send[off[i]:off[i+1]] to neigh_rank[i] and fills the same range of
        recv from it.  The arrays are views of the library's pinned buffers, valid during the call only.  An exception in a
        callback cannot cross the C frames: it is kept, the process's peers would wait for ever, so the process is ended."""
        def _guard(fn):
            def run(*a):
                try:
                    fn(*a)
                except BaseException:  # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    os._exit(70)
            return run

        @_guard
        def ar(_user, buf, n):
            allreduce(np.ctypeslib.as_array(buf, shape=(n,)))

        @_guard
        def ex(_user, nneigh, nr, off, send, recv):
            o = np.ctypeslib.as_array(off, shape=(nneigh + 1,))
            tot = int(o[nneigh])
            exchange(np.ctypeslib.as_array(nr, shape=(nneigh,)), o,
                     np.ctypeslib.as_array(send, shape=(max(tot, 1),))[:tot], np.ctypeslib.as_array(recv, shape=(max(tot, 1),))[:tot])
        car, cex = Comm._AR(ar), Comm._EX(ex)
        h = _h()
        _ck(_lib.fem_comm_create_callbacks(ctx.h, nranks, rank, C.cast(car, C.c_void_p), C.cast(cex, C.c_void_p), None, C.byref(h)))
        c = Comm(h)
        c._keep = (car, cex)          # the C side holds the function pointers for the communicator's lifetime
        return c

    @staticmethod
    def gloo(ctx: Context, group=None) -> "Comm":
        """The host-staged communicator over torch.distributed's gloo back end (process group already initialised): what
        bench.py --rehearse and the multi-process tests use to run the N > 1 paths as real processes on one GPU."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def allreduce(buf):
            dist.all_reduce(torch.from_numpy(buf), group=group)

        def exchange(nr, off, send, recv):
            ts, tr = torch.from_numpy(send), torch.from_numpy(recv)
            ops = []
            for i, q in enumerate(nr):
                a, b = int(off[i]), int(off[i + 1])
                if b > a:
                    ops.append(dist.P2POp(dist.isend, ts[a:b], int(q), group))
                    ops.append(dist.P2POp(dist.irecv, tr[a:b], int(q), group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        return Comm.callbacks(ctx, world, rank, allreduce, exchange)

    def stats(self) -> dict:
        """fem_comm_get_stats: calls, largest and total count (doubles) of the all-reduces and of the neighbour exchanges."""
        st = (C.c_longlong * 6)()
        _ck(_lib.fem_comm_get_stats(self.h, st))
        return {"allreduce_calls": st[0], "allreduce_max_doubles": st[1], "allreduce_total_doubles": st[2],
                "exchange_calls": st[3], "exchange_max_doubles": st[4], "exchange_total_doubles": st[5]}

    def exchange(self, neigh_rank, off, send: "Vec", recv: "Vec"):
        """Neighbour i gets send[off[i]:off[i+1]]; recv[off[i]:off[i+1]] is filled from neighbour i."""
        nr = _i32(neigh_rank)
        o = np.ascontiguousarray(off, dtype=np.int64)
        _ck(_lib.fem_comm_exchange(self.h, len(nr), _i(nr), o.ctypes.data_as(C.POINTER(C.c_longlong)), send.h, recv.h))

    def exchange_allreduce(self, neigh_rank, off, send: "Vec", recv: "Vec", red: "Vec", nred: int):
        """exchange() together with the in-place sum of red[:nred] over all domains (one RCCL group on that back end)."""
        nr = _i32(neigh_rank)
        o = np.ascontiguousarray(off, dtype=np.int64)
        _ck(_lib.fem_comm_exchange_allreduce(self.h, len(nr), _i(nr), o.ctypes.data_as(C.POINTER(C.c_longlong)), send.h, recv.h, red.h, int(nred)))

    def rank_size(self):
        r, s = C.c_int(0), C.c_int(0)
        _ck(_lib.fem_comm_rank(self.h, C.byref(r), C.byref(s)))
        return r.value, s.value

    def destroy(self):
        if self.h:
            _ck(_lib.fem_comm_destroy(self.h))
            self.h = _h()


class DeviceMesh:
    """One domain's mesh on the device (local_elements/coords/mat of src/m_global.F90)."""

    def __init__(self, ctx: Context, mesh):
        self.ctx = ctx
        self.mesh = mesh
        self.h = _h()
        et, ep, cn, mi = _i32(mesh.eltype), _i32(mesh.eptr), _i32(mesh.conn), _i32(mesh.mat_id)
        co, ma = _f64(mesh.coords), _f64(mesh.mats)
        _ck(_lib.fem_mesh_create(ctx.h, mesh.pdim, mesh.nels, mesh.nnds, _i(et), _i(ep), _i(cn), _i(mi), _d(co),
                                 ma.shape[0], _d(ma), C.byref(self.h)))

    def set_cohesive(self, mesh=None):
        """Cohesive elements of the deck (el%nlMat, cohMats): before the first Mat is created."""
        m = mesh or self.mesh
        nl = _i32(m.nlmat if m.nlmat is not None else np.zeros(m.nels))
        props = _f64(np.asarray([pr for _, pr in m.cohmats], dtype=np.float64).reshape(-1, 6))
        _ck(_lib.fem_mesh_set_cohesive(self.h, _i(nl), props.shape[0], _d(props)))

    def add_cohesive_traction(self, u_total: "Vec", dt: float, R: "Vec", rels_mode: int = 0, rels_nodes=None):
        """The element loop of CalcResidual (src/m_global.F90:186-205)."""
        rn = None if rels_nodes is None else _i32(rels_nodes)
        _ck(_lib.fem_vec_add_cohesive_traction(self.h, u_total.h, dt, int(rels_mode), None if rn is None else _i(rn), R.h))

    def enforce_bc_force(self, R: "Vec", dt: float, mesh=None):
        """EnforceBCForce(Residual, dt) (src/m_global.F90:439-455)."""
        m = mesh or self.mesh
        bn, bv = _i32(m.bcnode), _f64(m.bcval)
        _ck(_lib.fem_enforce_bc_force(self.h, bn.shape[0], _i(bn), _d(bv), dt, m.t, R.h))

    def set_halo(self, comm: Comm, neigh_rank, neigh_ptr, neigh_nodes):
        nr, npx, nn = _i32(neigh_rank), _i32(neigh_ptr), _i32(neigh_nodes)
        _ck(_lib.fem_mesh_set_halo(self.h, comm.h, len(nr), _i(nr), _i(npx), _i(nn)))

    def tile_plan_digest(self, builder: str = "device") -> dict:
        """Digest of the tile plan of `-mat_mult_type ebe` built afresh by "device" (the product's builder) or "host" (the
        cross-check); a Mat must have been created from this mesh."""
        d = (C.c_ulonglong * 12)()
        ms = C.c_double(0)
        _ck(_lib.fem_mesh_tile_plan_digest(self.h, builder.encode(), d, C.byref(ms)))
        names = ("tiles", "lane_slots", "tile_nodes", "element_visits", "hdr", "steps", "nodes", "elems", "conn16", "flags",
                 "lists", "geometry")
        out = {k: int(v) for k, v in zip(names, d)}
        out["setup_ms"] = ms.value
        return out

    def form_rhs(self, F: "Vec", t_init: float, dt: float, mesh=None):
        """FormRHS(t_init, dt) (src/m_global.F90:373)."""
        m = mesh or self.mesh
        bn, bv = _i32(m.bcnode), _f64(m.bcval)
        fn, fv = _i32(m.fnode), _f64(m.fval)
        te, tv = _i32(m.telsd), _f64(m.tval)
        _ck(_lib.fem_form_rhs(self.h, bn.shape[0], _i(bn), _d(bv), fn.shape[0], _i(fn), _d(fv), te.shape[0], _i(te),
                              _d(tv), m.t, t_init, dt, F.h))

    def recover_nodal_stress(self, U: "Vec") -> np.ndarray:
        """src/main.F90:253-292 -> stress_node [nnds, cpdim]."""
        m = self.mesh
        out = np.zeros(m.nnds * m.cpdim)
        _ck(_lib.fem_recover_nodal_stress(self.h, U.h, _d(out)))
        return out.reshape(m.nnds, m.cpdim, order="F")

    def recover_nodal_stress_vec(self, U: "Vec", S: "Vec"):
        _ck(_lib.fem_recover_nodal_stress_vec(self.h, U.h, S.h))

    def destroy(self):
        if self.h:
            _ck(_lib.fem_mesh_destroy(self.h))
            self.h = _h()


class Vec:
    def __init__(self, ctx: Context, n: int):
        self.ctx = ctx
        self.n = int(n)
        self.h = _h()
        _ck(_lib.fem_vec_create(ctx.h, self.n, C.byref(self.h)))

    def set(self, alpha: float):
        _ck(_lib.fem_vec_set(self.h, alpha))

    def zero_entries(self):
        self.set(0.0)

    def set_array(self, a):
        a = _f64(a)
        assert a.size == self.n
        _ck(_lib.fem_vec_set_array(self.h, _d(a)))

    def get_array(self) -> np.ndarray:
        out = np.zeros(self.n)
        _ck(_lib.fem_vec_get_array(self.h, _d(out)))
        return out

    def device_ptr(self) -> int:
        p = C.c_void_p()
        _ck(_lib.fem_vec_device_ptr(self.h, C.byref(p)))
        return p.value or 0

    def copy_to(self, other: "Vec"):
        _ck(_lib.fem_vec_copy(self.h, other.h))

    def axpy(self, a: float, x: "Vec"):
        _ck(_lib.fem_vec_axpy(self.h, a, x.h))

    def scale(self, a: float):
        _ck(_lib.fem_vec_scale(self.h, a))

    def dot(self, y: "Vec") -> float:
        d = C.c_double(0)
        _ck(_lib.fem_vec_dot(self.h, y.h, C.byref(d)))
        return d.value

    def norm(self) -> float:
        d = C.c_double(0)
        _ck(_lib.fem_vec_norm2(self.h, C.byref(d)))
        return d.value

    def destroy(self):
        if self.h:
            _ck(_lib.fem_vec_destroy(self.h))
            self.h = _h()


class Mat:
    """Mat_K: MatCreateAIJ (src/main.F90:154) + the assembly loop (:157-171)."""

    def __init__(self, dmesh: DeviceMesh, handle=None):
        self.dmesh = dmesh
        self.h = _h()
        if handle is not None:
            self.h = handle
        else:
            _ck(_lib.fem_mat_create_aij(dmesh.h, C.byref(self.h)))

    def duplicate(self) -> "Mat":
        """MatDuplicate(..., MAT_COPY_VALUES, ...) (src/main.F90:197)."""
        h = _h()
        _ck(_lib.fem_mat_duplicate(self.h, C.byref(h)))
        return Mat(self.dmesh, handle=h)

    def copy_to(self, other: "Mat"):
        """MatCopy(self, other, SAME_NONZERO_PATTERN) (src/m_global.F90:123)."""
        _ck(_lib.fem_mat_copy(self.h, other.h))

    def add_cohesive(self, u_total: "Vec", dt: float, rels_mode: int = 0, rels_nodes=None):
        """The element loop of CalcJacobian (src/m_global.F90:127-147)."""
        rn = None if rels_nodes is None else _i32(rels_nodes)
        _ck(_lib.fem_mat_add_cohesive(self.h, u_total.h, dt, int(rels_mode), None if rn is None else _i(rn)))

    def mult_add(self, x: "Vec", y: "Vec", z: "Vec"):
        """MatMultAdd: z = y + A x (src/m_global.F90:224)."""
        _ck(_lib.fem_mat_mult_add(self.h, x.h, y.h, z.h))

    def zero_entries(self):
        _ck(_lib.fem_mat_zero_entries(self.h))

    def assemble_elements(self):
        _ck(_lib.fem_mat_assemble_elements(self.h))

    def apply_kbc(self, bcnode=None):
        bn = _i32(self.dmesh.mesh.bcnode if bcnode is None else bcnode)
        _ck(_lib.fem_mat_apply_kbc(self.h, bn.shape[0], _i(bn)))

    def info(self):
        a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        _ck(_lib.fem_mat_get_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def get_csr(self):
        nrows, nnz, _ = self.info()
        rp = np.zeros(nrows + 1, dtype=np.int32); ci = np.zeros(nnz, dtype=np.int32); v = np.zeros(nnz)
        _ck(_lib.fem_mat_get_csr(self.h, _i(rp), _i(ci), _d(v)))
        return rp, ci, v

    def mult(self, x: Vec, y: Vec):
        _ck(_lib.fem_mat_mult(self.h, x.h, y.h))

    def set_profiling(self, on):
        """False/0 off; True/1 every product; p > 1 every p-th product."""
        _ck(_lib.fem_mat_set_profiling(self.h, int(on)))

    def get_profile(self, max_products: int):
        """(n, element_ms, gather_ms, penalty_ms) summed over the n sampled ones of the first max_products EBE products."""
        n = C.c_int(0)
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        _ck(_lib.fem_mat_get_profile(self.h, int(max_products), C.byref(n), C.byref(a), C.byref(b), C.byref(c)))
        return n.value, a.value, b.value, c.value

    def get_mult_info(self) -> dict:
        """kind 0 assembled / 1 element buffer + gather / 2 one-pass tiles, and the tile plan's figures."""
        k = C.c_int(0)
        a, b, c, d, e = (C.c_longlong(0) for _ in range(5))
        ms = C.c_double(0)
        _ck(_lib.fem_mat_get_mult_info(self.h, C.byref(k), C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e),
                                       C.byref(ms)))
        u, pt, st = C.c_longlong(0), C.c_int(0), C.c_int(0)
        _ck(_lib.fem_mat_get_tile_geometry_info(self.h, C.byref(u), C.byref(pt), C.byref(st)))
        return {"kind": k.value, "tiles": a.value, "affine_tiles": b.value, "element_visits": c.value, "lane_slots": d.value,
                "tile_nodes": e.value, "setup_ms": ms.value, "uniform_tiles": u.value, "per_tile_records": int(pt.value),
                "streaming_lists": bool(st.value)}

    def set_mult_type(self, kind: str):
        """'assembled' (sliced block-ELL SpMV), 'ebe' (element-by-element recomputation: one-pass tile kernel on a mesh of
        hex8 elements alone, element buffer + node gather otherwise) or 'ebe_gather' (always the latter)."""
        _ck(_lib.fem_mat_set_mult_type(self.h, kind.encode()))

    def mult_timed(self, x: Vec, y: Vec, reps: int) -> float:
        ms = C.c_double(0)
        _ck(_lib.fem_mat_mult_timed(self.h, x.h, y.h, reps, C.byref(ms)))
        return ms.value

    def set_values(self, rows, cols, v, mode="insert"):
        """MatSetValues (src/m_global.F90:257-260,296): dense len(rows) x len(cols) block, row-major, 0-based dofs."""
        rows = np.ascontiguousarray(rows, dtype=np.int32); cols = np.ascontiguousarray(cols, dtype=np.int32)
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(len(rows), len(cols))
        _ck(_lib.fem_mat_set_values(self.h, len(rows), _i(rows), len(cols), _i(cols), _d(v),
                                    {"insert": 1, "add": 2}[mode]))

    def get_diagonal(self, d: Vec):
        _ck(_lib.fem_mat_get_diagonal(self.h, d.h))

    def destroy(self):
        if self.h:
            _ck(_lib.fem_mat_destroy(self.h))
            self.h = _h()


class KSP:
    """Krylov (SetupKSPSolver src/main.F90:339-403; KSPSolve :247)."""

    def __init__(self, ctx: Context, handle=None):
        self.ctx = ctx
        self.h = _h()
        self.owned = handle is None
        if handle is not None:
            self.h = handle
        else:
            _ck(_lib.fem_ksp_create(ctx.h, C.byref(self.h)))

    def set_operators(self, A: Mat):
        _ck(_lib.fem_ksp_set_operators(self.h, A.h))

    def set_type(self, t: str):
        _ck(_lib.fem_ksp_set_type(self.h, t.encode()))

    def set_pc_type(self, t: str):
        _ck(_lib.fem_ksp_set_pc_type(self.h, t.encode()))

    def set_tolerances(self, rtol=DEFAULT, abstol=DEFAULT, dtol=DEFAULT, max_it=-2):
        _ck(_lib.fem_ksp_set_tolerances(self.h, rtol, abstol, dtol, max_it))

    def set_from_options(self, options: str):
        _ck(_lib.fem_ksp_set_from_options(self.h, options.encode()))

    def set_up(self):
        _ck(_lib.fem_ksp_set_up(self.h))

    def solve(self, b: Vec, x: Vec):
        _ck(_lib.fem_ksp_solve(self.h, b.h, x.h))

    def get_iteration_number(self) -> int:
        v = C.c_int(0)
        _ck(_lib.fem_ksp_get_iteration_number(self.h, C.byref(v)))
        return v.value

    def get_residual_norm(self) -> float:
        v = C.c_double(0)
        _ck(_lib.fem_ksp_get_residual_norm(self.h, C.byref(v)))
        return v.value

    def get_converged_reason(self) -> int:
        v = C.c_int(0)
        _ck(_lib.fem_ksp_get_converged_reason(self.h, C.byref(v)))
        return v.value

    def get_residual_history(self, maxlen: int = 100001) -> np.ndarray:
        out = np.zeros(maxlen); n = C.c_int(0)
        _ck(_lib.fem_ksp_get_residual_history(self.h, _d(out), maxlen, C.byref(n)))
        return out[:n.value]

    def set_profiling(self, on: bool):
        _ck(_lib.fem_ksp_set_profiling(self.h, int(on)))

    def get_profile(self):
        n = C.c_int(0); a = C.c_double(0); b = C.c_double(0)
        _ck(_lib.fem_ksp_get_profile(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value, b.value

    def get_reused_solves(self) -> int:
        """Solves answered with the kept solution of an identical earlier solve (-ksp_reuse_solution, fem_amd.h)."""
        n = C.c_int(0)
        _ck(_lib.fem_ksp_get_reused_solves(self.h, C.byref(n)))
        return n.value

    def get_resident_solves(self) -> int:
        """Solves that ran as one resident launch (-ksp_cg_resident, fem_amd.h)."""
        n = C.c_int(0)
        _ck(_lib.fem_ksp_get_resident_solves(self.h, C.byref(n)))
        return n.value

    def destroy(self):
        if self.h and self.owned:
            _ck(_lib.fem_ksp_destroy(self.h))
        self.h = _h()


SNES_FUNCTION_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
SNES_JACOBIAN_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


def _borrow_vec(ctx, h) -> "Vec":
    """A Vec the library passes to a callback (not owned here: destroy() is never called on it)."""
    v = Vec.__new__(Vec)
    v.ctx = ctx
    v.h = _h(h)
    n = C.c_longlong(0)
    _ck(_lib.fem_vec_get_size(v.h, C.byref(n)))
    v.n = n.value
    return v


class SNES:
    """Solver (src/main.F90:200-206, :242-243): Newton line search with the host's CalcResidual / CalcJacobian."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.h = _h()
        _ck(_lib.fem_snes_create(ctx.h, C.byref(self.h)))
        self._keep = []

    def get_ksp(self) -> KSP:
        h = _h()
        _ck(_lib.fem_snes_get_ksp(self.h, C.byref(h)))
        return KSP(self.ctx, handle=h)

    def set_function(self, r: Vec, fn):
        """fn(x, f) fills the residual f for the iterate x (both borrowed Vec handles)."""
        def tramp(_snes, x, f, _ctx):
            try:
                fn(_borrow_vec(self.ctx, x), _borrow_vec(self.ctx, f))
                return 0
            except Exception as e:   # noqa: BLE001 -- reported through the C return code
                self.error = e
                return 1
        cb = SNES_FUNCTION_CB(tramp)
        self._keep.append(cb)
        _ck(_lib.fem_snes_set_function(self.h, r.h, C.cast(cb, C.c_void_p), None))

    def set_jacobian(self, J: Mat, fn):
        """fn(x, J) fills the Jacobian for the iterate x."""
        def tramp(_snes, x, j, _p, _ctx):
            try:
                fn(_borrow_vec(self.ctx, x), J)
                return 0
            except Exception as e:   # noqa: BLE001
                self.error = e
                return 1
        cb = SNES_JACOBIAN_CB(tramp)
        self._keep.append(cb)
        _ck(_lib.fem_snes_set_jacobian(self.h, J.h, J.h, C.cast(cb, C.c_void_p), None))

    def set_defmod_callbacks(self, dmesh, K: Mat, J: Mat, R: Vec, F: Vec, agg: Vec, work: Vec, dt: float, rels_mode: int = 0, rels_nodes=None):
        """The reference's CalcJacobian / CalcResidual as the library's own callbacks (fem_snes_set_defmod_callbacks): the same
        primitives as driver.NonlinearProblem's hand-written pair, without the interpreter between them."""
        m = dmesh.mesh
        bn, bv = _i32(m.bcnode), _f64(m.bcval)
        rn = _i32(rels_nodes) if rels_nodes is not None else None
        st = C.c_void_p(0)
        _ck(_lib.fem_snes_set_defmod_callbacks(self.h, dmesh.h, K.h, J.h, R.h, F.h, agg.h, work.h, bn.shape[0], _i(bn), _d(bv), float(dt),
                                               float(m.t), int(rels_mode), _i(rn) if rn is not None else None, C.byref(st)))
        self._defmod = st

    def calc_defmod_jacobian(self, du: Vec):
        """CalcJacobian(Solver, Vec_U, Jacobian, Jacobian) in front of SNESSolve (src/main.F90:241)."""
        _ck(_lib.fem_snes_calc_defmod_jacobian(self._defmod, du.h))

    def set_from_options(self, options: str):
        _ck(_lib.fem_snes_set_from_options(self.h, options.encode()))

    def solve(self, x: Vec):
        self.error = None
        rc = _lib.fem_snes_solve(self.h, x.h)
        if rc and self.error is not None:
            raise self.error
        _ck(rc)

    def _geti(self, f):
        v = C.c_int(0)
        _ck(f(self.h, C.byref(v)))
        return v.value

    def get_iteration_number(self) -> int:
        return self._geti(_lib.fem_snes_get_iteration_number)

    def get_converged_reason(self) -> int:
        return self._geti(_lib.fem_snes_get_converged_reason)

    def get_linear_solve_iterations(self) -> int:
        return self._geti(_lib.fem_snes_get_linear_solve_iterations)

    def get_number_function_evals(self) -> int:
        return self._geti(_lib.fem_snes_get_number_function_evals)

    def get_function_norm(self) -> float:
        v = C.c_double(0)
        _ck(_lib.fem_snes_get_function_norm(self.h, C.byref(v)))
        return v.value

    def destroy(self):
        if self.h:
            _ck(_lib.fem_snes_destroy(self.h))
            self.h = _h()
            if getattr(self, "_defmod", None):
                _ck(_lib.fem_snes_free_defmod_callbacks(self._defmod))
                self._defmod = None
