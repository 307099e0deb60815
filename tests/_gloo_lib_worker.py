"""Worker of tests/test_gpu_rehearsal.py: one REAL process per domain (torch.distributed.run), all on GPU 0, the library's
collectives through the host-staged communicator over gloo (fem_comm_create_callbacks).  Every rank runs the library's
partitioned path -- fem_mesh_set_halo, sub-assembled Mat_K, KSPCG with owner-masked dots, the halo sums of the diagonal, of w
and of the nodal stress -- on its brick; rank 0 gathers the fields and checks them against the oracle's solve of the
undivided mesh (the reference's own mode is `mpiexec -n 2`, Makefile:6-10)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from fem_amd import capi, driver, partition
    from fem_amd.mesh import structured_hex
    npr = tuple(int(x) for x in sys.argv[1].split(","))
    options = sys.argv[2]
    jitter = float(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = capi.Context(0)
    comm = capi.Comm.gloo(ctx)
    assert comm.rank_size() == (rank, world)
    grid = partition.brick_grid(world)
    m = partition.brick_mesh(rank, grid, npr, jitter=jitter)
    prob = driver.Problem(ctx, m, options, comm=comm, halo=partition.brick_halo(rank, grid, npr))
    prob.assemble()
    prob.ksp.set_up()
    prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
    prob.ksp.solve(prob.F, prob.U)
    u = prob.U.get_array()
    s = prob.dmesh.recover_nodal_stress(prob.U)
    gid = partition.global_node_ids(rank, grid, npr)
    mine = (u, s, prob.ksp.get_iteration_number(), prob.ksp.get_converged_reason(), gid, comm.stats())
    out = [None] * world if rank == 0 else None
    dist.gather_object(mine, out, dst=0)
    ok = True
    if rank == 0:
        from oracle.oracle import Oracle, build
        build(ref=False)
        gn = tuple(grid[d] * npr[d] for d in range(3))
        g = structured_hex(*gn, jitter=jitter)
        ug, sg, gits = Oracle().run(g, "cg", rtol=1e-13)
        rep = {"world": world, "its_oracle": int(gits[0]), "ranks": []}
        for r, (ur, sr, its, reason, gr, st) in enumerate(out):
            eu = float(np.abs(ur.reshape(-1, 3) - ug.reshape(-1, 3)[gr]).max() / np.abs(ug).max())
            es = float(np.abs(sr - sg[gr]).max() / np.abs(sg).max())
            rep["ranks"].append({"rank": r, "its": int(its), "reason": int(reason), "err_u": eu, "err_s": es, "stats": st})
            ok = ok and reason > 0 and eu <= 1e-10 and es <= 1e-10
            if "-pc_type mg" not in options:
                ok = ok and abs(its - gits[0]) <= 3
        # all sharers of an interface node hold bitwise the same displacement (canonical-order halo sums)
        for r in range(world):
            for q in range(r + 1, world):
                common, ir, iq = np.intersect1d(out[r][4], out[q][4], return_indices=True)
                if common.size:
                    ok = ok and np.array_equal(out[r][0].reshape(-1, 3)[ir], out[q][0].reshape(-1, 3)[iq])
        rep["ok"] = bool(ok)
        print("GLOO_LIB " + json.dumps(rep), flush=True)
    prob.destroy(); comm.destroy(); ctx.destroy()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
