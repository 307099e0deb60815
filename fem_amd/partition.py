"""Element-partition domain decomposition for structured bricks.

The reference partitions the element graph with METIS on rank 0 and ships every
element with three blocking sends (src/m_io.F90:107-237); shared nodes are
duplicated per rank (src/main.F90:49-73).  Here each rank *generates* its own
brick of the global cube (no METIS, no element traffic) and only needs, per
neighbouring brick, the list of shared local node ids in an order both sides
agree on (ascending global node id) -- the input of fem_mesh_set_halo.
"""
from __future__ import annotations

import itertools

import numpy as np

from .mesh import Mesh, structured_hex


def brick_grid(nranks: int) -> tuple[int, int, int]:
    """Near-cubic px*py*pz = nranks (1, 2x1x1, 2x2x1, 2x2x2, ...), x fastest."""
    dims = [1, 1, 1]
    n, f, k = nranks, 2, 0
    factors = []
    while n > 1:
        while n % f == 0:
            factors.append(f); n //= f
        f += 1
    for f in sorted(factors, reverse=True):
        dims[int(np.argmin(dims))] *= f
        k += 1
    dims.sort(reverse=True)
    return dims[0], dims[1], dims[2]


def rank_coords(rank: int, grid) -> tuple[int, int, int]:
    px, py, _ = grid
    return rank % px, (rank // px) % py, rank // (px * py)


def brick_mesh(rank: int, grid, n_per_rank, **kw) -> Mesh:
    """This rank's brick (n_per_rank elements per edge) of the global cube."""
    nx, ny, nz = n_per_rank
    bx, by, bz = rank_coords(rank, grid)
    return structured_hex(nx, ny, nz, origin=(bx * nx, by * ny, bz * nz),
                          global_n=(grid[0] * nx, grid[1] * ny, grid[2] * nz), **kw)


def brick_halo(rank: int, grid, n_per_rank):
    """(neigh_rank, neigh_ptr, neigh_nodes) for fem_mesh_set_halo: for every
    adjacent brick (face, edge or corner) the shared local node ids, ordered by
    global node id (= local lexicographic (k, j, i) order on both sides)."""
    nx, ny, nz = n_per_rank
    px, py, pz = grid
    bx, by, bz = rank_coords(rank, grid)
    nxn, nyn = nx + 1, ny + 1
    entries = []
    for dz, dy, dx in itertools.product((-1, 0, 1), repeat=3):
        if dx == 0 and dy == 0 and dz == 0:
            continue
        qx, qy, qz = bx + dx, by + dy, bz + dz
        if not (0 <= qx < px and 0 <= qy < py and 0 <= qz < pz):
            continue
        q = qx + px * (qy + py * qz)
        ri = np.arange(nx + 1) if dx == 0 else np.array([0 if dx < 0 else nx])
        rj = np.arange(ny + 1) if dy == 0 else np.array([0 if dy < 0 else ny])
        rk = np.arange(nz + 1) if dz == 0 else np.array([0 if dz < 0 else nz])
        k, j, i = np.meshgrid(rk, rj, ri, indexing="ij")
        entries.append((q, (i + nxn * (j + nyn * k)).ravel().astype(np.int32)))
    entries.sort(key=lambda e: e[0])
    neigh_rank = np.array([e[0] for e in entries], dtype=np.int32)
    neigh_ptr = np.zeros(len(entries) + 1, dtype=np.int32)
    for n, e in enumerate(entries):
        neigh_ptr[n + 1] = neigh_ptr[n] + len(e[1])
    neigh_nodes = np.concatenate([e[1] for e in entries]) if entries else np.zeros(0, dtype=np.int32)
    return neigh_rank, neigh_ptr, neigh_nodes


def global_node_ids(rank: int, grid, n_per_rank) -> np.ndarray:
    """Global node id of every local node (nl2g of src/main.F90:78-96)."""
    nx, ny, nz = n_per_rank
    bx, by, bz = rank_coords(rank, grid)
    gx, gy = grid[0] * nx, grid[1] * ny
    k, j, i = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    return ((i + bx * nx) + (gx + 1) * ((j + by * ny) + (gy + 1) * (k + bz * nz))).ravel().astype(np.int64)


def general_halo(rank: int, nl2g_all: list[np.ndarray]):
    """Interface lists for an arbitrary partition given every rank's
    local->global node map: shared nodes with each other rank, sorted by global id."""
    mine = nl2g_all[rank]
    order = np.argsort(mine)
    sorted_g = mine[order]
    ranks, ptr, nodes = [], [0], []
    for q, other in enumerate(nl2g_all):
        if q == rank:
            continue
        common = np.intersect1d(sorted_g, other)
        if common.size == 0:
            continue
        loc = order[np.searchsorted(sorted_g, common)]
        ranks.append(q); nodes.append(loc.astype(np.int32)); ptr.append(ptr[-1] + len(loc))
    return (np.array(ranks, dtype=np.int32), np.array(ptr, dtype=np.int32),
            np.concatenate(nodes) if nodes else np.zeros(0, dtype=np.int32))


# ---- general (unstructured) element partition ------------------------------------------------------
def rcb_partition(mesh: Mesh, nparts: int) -> np.ndarray:
    """Element -> part by recursive coordinate bisection of the element centroids
    (stands in for METIS_PartMeshNodal, src/m_io.F90:137, which is not in the image).
    Deterministic; parts differ in size by at most one element per split."""
    cent = np.zeros((mesh.nels, mesh.pdim))
    for e in range(mesh.nels):
        cent[e] = mesh.coords[mesh.element_nodes(e)].mean(axis=0)
    epart = np.zeros(mesh.nels, dtype=np.int32)

    def split(idx, p0, np_):
        if np_ == 1 or len(idx) == 0:
            epart[idx] = p0
            return
        nl = np_ // 2
        c = cent[idx]
        ax = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
        order = idx[np.lexsort((idx, c[:, ax]))]
        cut = (len(idx) * nl) // np_
        split(order[:cut], p0, nl)
        split(order[cut:], p0 + nl, np_ - nl)

    split(np.arange(mesh.nels), 0, nparts)
    return epart


def split_mesh(mesh: Mesh, epart: np.ndarray, rank: int):
    """The local mesh of one part, numbered as the reference does after
    DistributeElements (src/main.F90:49-96): local nodes in ascending global id.
    Returns (local Mesh, nl2g).  BCs and nodal forces are kept on every part that
    holds the node (consistent RHS); tractions follow their element."""
    els = np.nonzero(epart == rank)[0]
    conn_g = np.concatenate([mesh.element_nodes(e) for e in els]) if len(els) else np.zeros(0, dtype=np.int32)
    nl2g = np.unique(conn_g)
    g2l = -np.ones(mesh.nnds, dtype=np.int64)
    g2l[nl2g] = np.arange(len(nl2g))
    eptr = np.zeros(len(els) + 1, dtype=np.int32)
    for k, e in enumerate(els):
        eptr[k + 1] = eptr[k] + (mesh.eptr[e + 1] - mesh.eptr[e])
    keep_bc = g2l[mesh.bcnode[:, 0]] >= 0 if mesh.bcnode.shape[0] else np.zeros(0, dtype=bool)
    keep_f = g2l[mesh.fnode] >= 0 if mesh.fnode.shape[0] else np.zeros(0, dtype=bool)
    e2l = -np.ones(mesh.nels, dtype=np.int64)
    e2l[els] = np.arange(len(els))
    keep_t = e2l[mesh.telsd[:, 0]] >= 0 if mesh.telsd.shape[0] else np.zeros(0, dtype=bool)
    bcnode = mesh.bcnode[keep_bc].copy()
    if bcnode.shape[0]:
        bcnode[:, 0] = g2l[bcnode[:, 0]]
    telsd = mesh.telsd[keep_t].copy()
    if telsd.shape[0]:
        telsd[:, 0] = e2l[telsd[:, 0]]
    loc = Mesh(pdim=mesh.pdim, eltype=mesh.eltype[els], eptr=eptr, conn=g2l[conn_g], mat_id=mesh.mat_id[els],
               coords=mesh.coords[nl2g], mats=mesh.mats, bcnode=bcnode, bcval=mesh.bcval[keep_bc],
               fnode=g2l[mesh.fnode[keep_f]], fval=mesh.fval[keep_f], telsd=telsd, tval=mesh.tval[keep_t],
               t=mesh.t, dt=mesh.dt)
    return loc, nl2g.astype(np.int64)
