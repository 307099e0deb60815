"""Host-side mesh / problem description and the synthetic-mesh generators.

The record mirrors what the reference holds after its input phase
(`local_elements`, `coords`, `mat`, `bcnode/bcval`, `fnode/fval`, `telsd/tval`,
`t`, `dt`: /root/reference/src/m_global.F90:14-87, main.F90:39-103), flattened
to contiguous SoA arrays because allocatable-component derived types
(`type element`, m_elems.F90:6-12) cannot cross a C ABI.  Ids are 0-based here
(the reference is 1-based in Fortran and 0-based at its PETSc boundary).

Generators follow SURVEY.md section 8(d): node numbering x-fastest as in
make_example.F90:64-85, hex connectivity bottom face CCW then top face CCW
(m_elems.F90:557-564 ordering).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

TRI, QUA, TET, HEX, COH = 1, 2, 3, 4, 5
TYPE_NAMES = {"tri": TRI, "qua": QUA, "tet": TET, "hex": HEX, "coh": COH}
NODE_COUNT = {TRI: 3, QUA: 4, TET: 4, HEX: 8, COH: 4}
NIP = {TRI: 1, QUA: 4, TET: 1, HEX: 8, COH: 2}
ELDIM = {TRI: 2, QUA: 2, TET: 3, HEX: 3, COH: 2}


def _e(shape, dt):
    return np.zeros(shape, dtype=dt)


@dataclass
class Mesh:
    pdim: int
    eltype: np.ndarray  # [nels] int32
    eptr: np.ndarray    # [nels+1] int32
    conn: np.ndarray    # int32 node ids
    mat_id: np.ndarray  # [nels] int32, 0-based row of mats
    coords: np.ndarray  # [nnds, pdim] float64
    mats: np.ndarray    # [nmats, 5]  E nu visc expn density
    bcnode: np.ndarray = field(default_factory=lambda: _e((0, 1), np.int32))   # [nbcs, 1+pdim]
    bcval: np.ndarray = field(default_factory=lambda: _e((0, 1), np.float64))  # [nbcs, pdim]
    fnode: np.ndarray = field(default_factory=lambda: _e((0,), np.int32))
    fval: np.ndarray = field(default_factory=lambda: _e((0, 1), np.float64))   # [nfrcs, pdim+2]
    telsd: np.ndarray = field(default_factory=lambda: _e((0, 2), np.int32))    # element (0-based), side (1-based)
    tval: np.ndarray = field(default_factory=lambda: _e((0, 1), np.float64))
    t: float = 1.0
    dt: float = 1.0
    stype: str = "implicit"
    nodal_bw: int = 10
    nlmat: np.ndarray | None = None
    cohmats: list = field(default_factory=list)

    def __post_init__(self):
        pd = self.pdim
        self.eltype = np.ascontiguousarray(self.eltype, dtype=np.int32)
        self.eptr = np.ascontiguousarray(self.eptr, dtype=np.int32)
        self.conn = np.ascontiguousarray(self.conn, dtype=np.int32)
        self.mat_id = np.ascontiguousarray(self.mat_id, dtype=np.int32)
        self.coords = np.ascontiguousarray(self.coords, dtype=np.float64).reshape(-1, pd)
        self.mats = np.ascontiguousarray(self.mats, dtype=np.float64).reshape(-1, 5)
        self.bcnode = np.ascontiguousarray(self.bcnode, dtype=np.int32).reshape(-1, 1 + pd)
        self.bcval = np.ascontiguousarray(self.bcval, dtype=np.float64).reshape(-1, pd)
        self.fnode = np.ascontiguousarray(self.fnode, dtype=np.int32).reshape(-1)
        self.fval = np.ascontiguousarray(self.fval, dtype=np.float64).reshape(-1, pd + 2)
        self.telsd = np.ascontiguousarray(self.telsd, dtype=np.int32).reshape(-1, 2)
        self.tval = np.ascontiguousarray(self.tval, dtype=np.float64).reshape(-1, pd + 2)

    @property
    def nels(self) -> int:
        return int(self.eltype.shape[0])

    @property
    def nnds(self) -> int:
        return int(self.coords.shape[0])

    @property
    def ndof(self) -> int:
        return self.nnds * self.pdim

    @property
    def cpdim(self) -> int:
        return 3 if self.pdim == 2 else 6

    @property
    def nsteps(self) -> int:
        """Increment count of main.F90:216-219 (exit when dt*(k-1) >= t)."""
        k = 0
        while self.dt * k < self.t:
            k += 1
        return k

    def element_nodes(self, e: int) -> np.ndarray:
        return self.conn[self.eptr[e]:self.eptr[e + 1]]

    def uniform_type(self):
        """(type, npe) if every element has the same type, else None."""
        if self.nels and np.all(self.eltype == self.eltype[0]):
            return int(self.eltype[0]), NODE_COUNT[int(self.eltype[0])]
        return None


def _hash01(ids: np.ndarray, seed: int) -> np.ndarray:
    """Deterministic integer hash -> [-0.5, 0.5) (splitmix64 finaliser)."""
    off = np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        k = ids.astype(np.uint64) + off
        k ^= k >> np.uint64(33); k *= np.uint64(0xFF51AFD7ED558CCD)
        k ^= k >> np.uint64(33); k *= np.uint64(0xC4CEB9FE1A85EC53)
        k ^= k >> np.uint64(33)
    return (k >> np.uint64(11)).astype(np.float64) / 9007199254740992.0 - 0.5


def structured_hex(nx: int, ny: int | None = None, nz: int | None = None, h: float = 1.0,
                   E: float = 3.0e10, nu: float = 0.25, jitter: float = 0.0, seed: int = 12345,
                   top_disp: float | None = None, origin=(0, 0, 0), global_n=None) -> Mesh:
    """Config 3: nx*ny*nz hex8 brick, h=1, face z=0 clamped,
    face z=top prescribed u_z = -0.01*nz_global through the penalty, 1 increment.

    ``origin``/``global_n`` place the brick inside a larger global cube (element
    offsets / global element counts) so that a partition's BCs and the node
    hash of the distorted variant follow the *global* mesh.
    """
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    gx, gy, gz = global_n if global_n is not None else (nx, ny, nz)
    ox, oy, oz = origin
    nxn, nyn, nzn = nx + 1, ny + 1, nz + 1
    k, j, i = np.meshgrid(np.arange(nzn), np.arange(nyn), np.arange(nxn), indexing="ij")
    gi, gj, gk = i + ox, j + oy, k + oz
    coords = np.stack([gi.ravel() * h, gj.ravel() * h, gk.ravel() * h], axis=1).astype(np.float64)
    gid = (gi + (gx + 1) * (gj + (gy + 1) * gk)).ravel().astype(np.int64)
    if jitter:
        for d in range(3):
            coords[:, d] += jitter * h * _hash01(gid * 3 + d, seed)
    ke, je, ie = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    n0 = (ie + nxn * (je + nyn * ke)).ravel()
    conn = np.stack([n0, n0 + 1, n0 + 1 + nxn, n0 + nxn,
                     n0 + nxn * nyn, n0 + 1 + nxn * nyn, n0 + 1 + nxn + nxn * nyn, n0 + nxn + nxn * nyn], axis=1)
    nels = conn.shape[0]
    bot = np.nonzero(gk.ravel() == 0)[0]
    top = np.nonzero(gk.ravel() == gz)[0]
    disp = -0.01 * gz * h if top_disp is None else top_disp
    bcnode = np.concatenate([
        np.stack([bot, np.zeros_like(bot), np.zeros_like(bot), np.zeros_like(bot)], axis=1),
        np.stack([top, np.ones_like(top), np.ones_like(top), np.zeros_like(top)], axis=1)])
    bcval = np.concatenate([np.zeros((len(bot), 3)),
                            np.stack([np.zeros(len(top)), np.zeros(len(top)), np.full(len(top), disp)], axis=1)])
    return Mesh(pdim=3, eltype=np.full(nels, HEX), eptr=np.arange(nels + 1) * 8, conn=conn.ravel(),
                mat_id=np.zeros(nels), coords=coords, mats=[[E, nu, 1.0e18, 1.0, 3000.0]],
                bcnode=bcnode, bcval=bcval, t=1.0, dt=1.0)


def structured_quad(nx: int, ny: int | None = None, h: float = 1.0, E: float = 3.0e10, nu: float = 0.25,
                    jitter: float = 0.0, seed: int = 12345, fx: float = 1.0e6) -> Mesh:
    """Config 2: nx*ny quad4 plate, element (j, j+1, j+1+nxn, j+nxn)
    CCW, left edge clamped, nodal force f_x on every right-edge node over [0,t]."""
    ny = nx if ny is None else ny
    nxn, nyn = nx + 1, ny + 1
    j, i = np.meshgrid(np.arange(nyn), np.arange(nxn), indexing="ij")
    coords = np.stack([i.ravel() * h, j.ravel() * h], axis=1).astype(np.float64)
    if jitter:
        gid = (i + nxn * j).ravel().astype(np.int64)
        for d in range(2):
            coords[:, d] += jitter * h * _hash01(gid * 2 + d, seed)
    je, ie = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    n0 = (ie + nxn * je).ravel()
    conn = np.stack([n0, n0 + 1, n0 + 1 + nxn, n0 + nxn], axis=1)
    nels = conn.shape[0]
    left = np.nonzero(i.ravel() == 0)[0]
    right = np.nonzero(i.ravel() == nx)[0]
    bcnode = np.stack([left, np.zeros_like(left), np.zeros_like(left)], axis=1)
    bcval = np.zeros((len(left), 2))
    fval = np.stack([np.full(len(right), fx), np.zeros(len(right)), np.zeros(len(right)), np.ones(len(right))], axis=1)
    return Mesh(pdim=2, eltype=np.full(nels, QUA), eptr=np.arange(nels + 1) * 4, conn=conn.ravel(),
                mat_id=np.zeros(nels), coords=coords, mats=[[E, nu, 1.0e18, 1.0, 3000.0]],
                bcnode=bcnode, bcval=bcval, fnode=right, fval=fval, t=1.0, dt=1.0)


def hex_to_tets(m: Mesh) -> Mesh:
    """Split every hex8 of ``m`` into 6 tet4 (the ElVolHex split, m_elems.F90:512-535)
    -- used by tests to exercise the tet path on 3-D meshes."""
    split = np.array([[0, 2, 7, 6], [0, 2, 6, 3], [0, 4, 5, 6], [0, 4, 6, 7], [0, 1, 4, 7], [0, 1, 7, 2]])
    hexes = m.conn.reshape(-1, 8)
    tets = hexes[:, split].reshape(-1, 4)
    # make every tet positively oriented (signed detJ is kept as is by the reference)
    c = m.coords[tets]
    v = np.einsum("ij,ij->i", np.cross(c[:, 1] - c[:, 0], c[:, 2] - c[:, 0]), c[:, 3] - c[:, 0])
    flip = v < 0
    tets[flip, 2], tets[flip, 3] = tets[flip, 3].copy(), tets[flip, 2].copy()
    n = tets.shape[0]
    return Mesh(pdim=3, eltype=np.full(n, TET), eptr=np.arange(n + 1) * 4, conn=tets.ravel(),
                mat_id=np.repeat(m.mat_id, 6), coords=m.coords, mats=m.mats, bcnode=m.bcnode, bcval=m.bcval,
                fnode=m.fnode, fval=m.fval, t=m.t, dt=m.dt)


def quad_to_mixed(m: Mesh) -> Mesh:
    """Replace every second quad4 by two tri3 -- a tri+qua mix like
    examples/lin_two_quads_qs.inp, for the mixed-type assembly path."""
    quads = m.conn.reshape(-1, 4)
    eltype, eptr, conn, mat = [], [0], [], []
    for e, q in enumerate(quads):
        if e % 2 == 0:
            eltype.append(QUA); conn += list(q); eptr.append(len(conn)); mat.append(m.mat_id[e])
        else:
            for tri in ((q[0], q[1], q[2]), (q[0], q[2], q[3])):
                eltype.append(TRI); conn += list(tri); eptr.append(len(conn)); mat.append(m.mat_id[e])
    return Mesh(pdim=2, eltype=eltype, eptr=eptr, conn=conn, mat_id=mat, coords=m.coords, mats=m.mats,
                bcnode=m.bcnode, bcval=m.bcval, fnode=m.fnode, fval=m.fval, t=m.t, dt=m.dt)


def tile_deck(m: Mesh, copies: int, shift=None) -> Mesh:
    """`copies` disjoint copies of a deck side by side (BASELINE config 5: examples/cohesive_test_2.inp "scaled x1000"):
    every copy keeps its own nodes, elements, constraints and loads, shifted by `shift` (default: the deck's extent
    in x plus one), node and element numbers offset copy by copy."""
    ext = m.coords.max(axis=0) - m.coords.min(axis=0)
    sh = np.zeros(m.pdim) if shift is None else np.asarray(shift, dtype=np.float64)
    if shift is None:
        sh[0] = ext[0] + 1.0
    nn = m.nnds
    coords = np.concatenate([m.coords + c * sh for c in range(copies)])
    conn = np.concatenate([m.conn + c * nn for c in range(copies)])
    eptr = np.concatenate([[0]] + [m.eptr[1:] + c * m.eptr[-1] for c in range(copies)])
    bcnode = np.concatenate([m.bcnode + np.array([c * nn] + [0] * m.pdim) for c in range(copies)]) if len(m.bcnode) else m.bcnode
    telsd = np.concatenate([m.telsd + np.array([c * m.nels, 0]) for c in range(copies)]) if len(m.telsd) else m.telsd
    return Mesh(pdim=m.pdim, eltype=np.tile(m.eltype, copies), eptr=eptr, conn=conn, mat_id=np.tile(m.mat_id, copies),
                coords=coords, mats=m.mats, bcnode=bcnode, bcval=np.tile(m.bcval, (copies, 1)),
                fnode=np.concatenate([m.fnode + c * nn for c in range(copies)]) if len(m.fnode) else m.fnode,
                fval=np.tile(m.fval, (copies, 1)), telsd=telsd, tval=np.tile(m.tval, (copies, 1)), t=m.t, dt=m.dt,
                stype=m.stype, nodal_bw=m.nodal_bw, nlmat=None if m.nlmat is None else np.tile(m.nlmat, copies),
                cohmats=m.cohmats)


def tile_deck_connected(m: Mesh, copies: int) -> Mesh:
    """`copies` copies of a 2-D deck side by side along x as ONE connected strip (BASELINE config 5 as SURVEY 8(d) words it:
    examples/cohesive_test_2.inp "tiled x1000 along x"): the nodes on the right edge (x = max) of a copy are the nodes on
    the left edge (x = min) of the next one, matched by their y in the order of their numbers (the deck's duplicated
    interface nodes pair up in that order), so that the strip holds `copies` cohesive elements on one continuous interface.
    Constraints and loads of every copy are kept (records that land on one node are merged by ApplyKBC as the reference's
    sequential INSERT does)."""
    assert m.pdim == 2
    x = m.coords[:, 0]
    ext = x.max() - x.min()
    left = np.nonzero(x == x.min())[0]
    right = np.nonzero(x == x.max())[0]
    left = left[np.lexsort((left, m.coords[left, 1]))]
    right = right[np.lexsort((right, m.coords[right, 1]))]
    assert len(left) == len(right) and np.array_equal(m.coords[left, 1], m.coords[right, 1]), "edges of the deck do not match"
    nn = m.nnds
    inner = np.setdiff1d(np.arange(nn), left)               # nodes a later copy adds (its left edge is the previous right edge)
    newid = np.empty((copies, nn), dtype=np.int64)
    newid[0] = np.arange(nn)
    coords = [m.coords]
    for c in range(1, copies):
        base = nn + (c - 1) * len(inner)
        newid[c, inner] = base + np.arange(len(inner))
        newid[c, left] = newid[c - 1, right]
        coords.append(m.coords[inner] + np.array([c * ext, 0.0]))
    conn = np.concatenate([newid[c][m.conn] for c in range(copies)]).astype(np.int32)
    eptr = np.concatenate([[0]] + [m.eptr[1:] + c * m.eptr[-1] for c in range(copies)])
    bcnode = m.bcnode
    if len(m.bcnode):
        bcnode = np.concatenate([np.column_stack([newid[c][m.bcnode[:, 0]], m.bcnode[:, 1:]]) for c in range(copies)]).astype(np.int32)
    telsd = np.concatenate([m.telsd + np.array([c * m.nels, 0]) for c in range(copies)]) if len(m.telsd) else m.telsd
    return Mesh(pdim=m.pdim, eltype=np.tile(m.eltype, copies), eptr=eptr, conn=conn, mat_id=np.tile(m.mat_id, copies),
                coords=np.ascontiguousarray(np.concatenate(coords)), mats=m.mats, bcnode=bcnode, bcval=np.tile(m.bcval, (copies, 1)),
                fnode=np.concatenate([newid[c][m.fnode] for c in range(copies)]).astype(np.int32) if len(m.fnode) else m.fnode,
                fval=np.tile(m.fval, (copies, 1)), telsd=telsd, tval=np.tile(m.tval, (copies, 1)), t=m.t, dt=m.dt,
                stype=m.stype, nodal_bw=m.nodal_bw, nlmat=None if m.nlmat is None else np.tile(m.nlmat, copies),
                cohmats=m.cohmats)
