"""GPU: the multi-domain (halo) path of the library on ONE GPU -- n brick
domains driven by n host threads through the in-process communicator
(fem_comm_create_local_group).  Everything except the RCCL transport itself is
the code the 8-GPU run executes: interface packing, canonical-order unpack,
owner masks, owner-held penalties, fused un-assembled p.Kp, halo-summed Jacobi
diagonal and nodal stress/valence."""
import threading

import numpy as np
import pytest

from fem_amd import capi, driver, partition
from fem_amd.mesh import structured_hex

pytestmark = pytest.mark.gpu


def _run_domains(world, npr, jitter, options):
    grid = partition.brick_grid(world)
    ctxs = [capi.Context(0) for _ in range(world)]
    comms = capi.Comm.local_group(ctxs)
    out = [None] * world
    err = []

    def work(r):
        try:
            m = partition.brick_mesh(r, grid, npr, jitter=jitter)
            prob = driver.Problem(ctxs[r], m, options, comm=comms[r], halo=partition.brick_halo(r, grid, npr))
            prob.assemble()
            prob.ksp.set_up()
            prob.dmesh.form_rhs(prob.F, 0.0, m.dt)
            prob.ksp.solve(prob.F, prob.U)
            u = prob.U.get_array()
            s = prob.dmesh.recover_nodal_stress(prob.U)
            # MatMult through the public API (sub-assembled product + halo sum)
            X, Y = capi.Vec(ctxs[r], m.ndof), capi.Vec(ctxs[r], m.ndof)
            gid = partition.global_node_ids(r, grid, npr)
            xg = np.sin(0.37 * np.arange(3 * (gid.max() + 1) + 3))
            X.set_array(np.stack([xg[3 * gid], xg[3 * gid + 1], xg[3 * gid + 2]], axis=1).ravel())
            prob.K.mult(X, Y)
            out[r] = (u, s, prob.ksp.get_iteration_number(), prob.ksp.get_converged_reason(), Y.get_array(), gid)
            X.destroy(); Y.destroy(); prob.destroy()
        except Exception as e:  # noqa: BLE001
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not err, err
    assert all(o is not None for o in out), "a domain thread hung"
    for c in comms:
        c.destroy()
    for c in ctxs:
        c.destroy()
    return grid, out


@pytest.mark.parametrize("world,npr", [(2, (5, 4, 3)), (4, (3, 3, 4)), (6, (2, 3, 3))])
def test_multidomain_matches_serial_oracle(oracle, world, npr):
    grid, out = _run_domains(world, npr, 0.15, "-ksp_rtol 1e-13")
    gn = tuple(grid[d] * npr[d] for d in range(3))
    g = structured_hex(*gn, jitter=0.15)
    ug, sg, gits = oracle.run(g, "cg", rtol=1e-13)
    rp, ci = oracle.csr_pattern(g)
    val = oracle.assemble(g, rp, ci, apply_bc=True)
    xg = np.sin(0.37 * np.arange(g.ndof + 3))[:g.ndof]
    yg = oracle.spmv(rp, ci, val, xg).reshape(-1, 3)
    for r, (u, s, its, reason, y, gid) in enumerate(out):
        assert reason > 0
        assert np.abs(u.reshape(-1, 3) - ug.reshape(-1, 3)[gid]).max() <= 1e-10 * np.abs(ug).max()
        assert np.abs(s - sg[gid]).max() <= 1e-9 * np.abs(sg).max()
        assert abs(its - gits[0]) <= 3
        assert np.abs(y.reshape(-1, 3) - yg[gid]).max() <= 1e-12 * np.abs(yg).max()
    # all sharers hold bitwise-identical interface values (canonical-order halo sums)
    for r in range(world):
        for q in range(r + 1, world):
            common, ir, iq = np.intersect1d(out[r][5], out[q][5], return_indices=True)
            if common.size:
                assert np.array_equal(out[r][0].reshape(-1, 3)[ir], out[q][0].reshape(-1, 3)[iq])
