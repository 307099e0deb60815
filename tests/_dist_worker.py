"""Worker for tests/test_distributed_gloo.py (launched once per rank).

Runs the multi-domain algorithm the HIP library implements -- sub-assembled
local matrices, neighbour halo sums in ascending-rank order, owner-masked dot
products, penalty held by the owning domain only -- with the oracle's local
kernels and torch.distributed (gloo) as the transport, and checks the result
against the serial oracle solution of the global mesh.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd import partition  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402
from oracle.oracle import Oracle, PENALTY_PARAM  # noqa: E402


def halo_sum(vec, ncomp, plan, rank):
    """vec [nnodes, ncomp]; contributions added in ascending rank order (canonical)."""
    nr, ptr, nodes = plan
    sends = [torch.from_numpy(np.ascontiguousarray(vec[nodes[ptr[i]:ptr[i + 1]]])) for i in range(len(nr))]
    recvs = [torch.zeros_like(s) for s in sends]
    ops = []
    for i, q in enumerate(nr):
        ops.append(dist.P2POp(dist.isend, sends[i], int(q)))
        ops.append(dist.P2POp(dist.irecv, recvs[i], int(q)))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    contrib = {}
    for i, q in enumerate(nr):
        for k, v in enumerate(nodes[ptr[i]:ptr[i + 1]]):
            contrib.setdefault(int(v), []).append((int(q), recvs[i][k].numpy()))
    for v, lst in contrib.items():
        lst.append((rank, vec[v].copy()))
        lst.sort(key=lambda e: e[0])
        acc = np.zeros(ncomp)
        for _, c in lst:
            acc = acc + c
        vec[v] = acc


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    o = Oracle()
    npr = (3, 3, 2)
    grid = partition.brick_grid(world)
    m = partition.brick_mesh(rank, grid, npr, jitter=0.15)
    plan = partition.brick_halo(rank, grid, npr)
    nr, ptr, nodes = plan
    owned = np.ones(m.nnds, dtype=bool)
    for i, q in enumerate(nr):
        if q < rank:
            owned[nodes[ptr[i]:ptr[i + 1]]] = False
    # local (sub-assembled) matrix; penalty only on the owner
    rp, ci = o.csr_pattern(m)
    val = o.assemble(m, rp, ci, apply_bc=True)
    for i in range(m.bcnode.shape[0]):
        v = m.bcnode[i, 0]
        for d in range(3):
            if m.bcnode[i, 1 + d] == 0 and not owned[v]:
                r = 3 * v + d
                k = rp[r] + np.searchsorted(ci[rp[r]:rp[r + 1]], r)
                assert val[k] == PENALTY_PARAM
                val[k] = 0.0
    F = o.form_rhs(m, 0.0, m.dt)                      # consistent: every sharer holds the full value
    diag = np.array([val[rp[r] + np.searchsorted(ci[rp[r]:rp[r + 1]], r)] for r in range(m.ndof)]).reshape(-1, 3)
    halo_sum(diag, 3, plan, rank)
    dinv = np.where(diag != 0, 1.0 / np.where(diag != 0, diag, 1.0), 1.0).ravel()
    mask = np.repeat(owned, 3)

    def gdot(a, b):
        t = torch.tensor([float(np.dot(a[mask], b[mask]))], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    # KSPCG, preconditioned norm (same flow as the library)
    x = np.zeros(m.ndof); r = F.copy(); z = dinv * r; p = np.zeros(m.ndof)
    rtol = 1e-13
    dp = np.sqrt(gdot(z, z)); ttol = rtol * dp
    beta = gdot(z, r); betaold = 1.0
    its = 0
    for i in range(2000):
        if dp <= ttol:
            break
        p = z + (beta / betaold if i else 0.0) * p
        w = o.spmv(rp, ci, val, p)
        t = torch.tensor([float(np.dot(p, w))], dtype=torch.float64)   # un-assembled p.Kp: no mask
        dist.all_reduce(t)
        w2 = w.reshape(-1, 3)
        halo_sum(w2, 3, plan, rank)
        w = w2.ravel()
        a = beta / float(t.item())
        x += a * p; r -= a * w; z = dinv * r
        dp = np.sqrt(gdot(z, z)); betaold = beta; beta = gdot(z, r)
        its = i + 1
    # nodal stress: local sums + halo sums of stress and valence
    sn, cnt = o.recover_nodal_stress(m, x)
    s_sum = sn * cnt[:, None]
    halo_sum(s_sum, 6, plan, rank)
    val_d = cnt.astype(np.float64).reshape(-1, 1)
    halo_sum(val_d, 1, plan, rank)
    stress = s_sum / val_d

    # serial oracle on the global mesh
    gn = tuple(grid[d] * npr[d] for d in range(3))
    g = structured_hex(*gn, jitter=0.15)
    ug, sg, gits = o.run(g, "cg", rtol=rtol)
    gid = partition.global_node_ids(rank, grid, npr)
    eu = np.abs(x.reshape(-1, 3) - ug.reshape(-1, 3)[gid]).max() / np.abs(ug).max()
    es = np.abs(stress - sg[gid]).max() / np.abs(sg).max()
    ok = eu < 1e-10 and es < 1e-9 and abs(its - gits[0]) <= 3
    print(f"rank {rank}: its {its} (serial {gits[0]}) err_u {eu:.2e} err_s {es:.2e} ok={ok}", flush=True)
    t = torch.tensor([0 if ok else 1])
    dist.all_reduce(t)
    dist.destroy_process_group()
    sys.exit(int(t.item()))


if __name__ == "__main__":
    main()
