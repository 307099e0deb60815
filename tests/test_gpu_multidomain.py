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


LAST_STATS = []      # per domain of the last _run_domains call: fem_comm_get_stats + the number of interface node slots


def _run_domains(world, npr, jitter, options, extras=True):
    grid = partition.brick_grid(world)
    ctxs = [capi.Context(0) for _ in range(world)]
    comms = capi.Comm.local_group(ctxs)
    out = [None] * world
    stats = [None] * world
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
            gid = partition.global_node_ids(r, grid, npr)
            y = None
            if extras:
                # MatMult through the public API (sub-assembled product + halo sum)
                X, Y = capi.Vec(ctxs[r], m.ndof), capi.Vec(ctxs[r], m.ndof)
                xg = np.sin(0.37 * np.arange(3 * (gid.max() + 1) + 3))
                X.set_array(np.stack([xg[3 * gid], xg[3 * gid + 1], xg[3 * gid + 2]], axis=1).ravel())
                prob.K.mult(X, Y)
                y = Y.get_array()
                X.destroy(); Y.destroy()
            out[r] = (u, s, prob.ksp.get_iteration_number(), prob.ksp.get_converged_reason(), y, gid)
            stats[r] = dict(comms[r].stats(), interface_slots=int(len(partition.brick_halo(r, grid, npr)[2])))
            prob.destroy()
        except Exception as e:  # noqa: BLE001
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=1200)
    assert not err, err
    assert all(o is not None for o in out), "a domain thread hung"
    for c in comms:
        c.destroy()
    for c in ctxs:
        c.destroy()
    LAST_STATS[:] = stats
    return grid, out


def assert_no_collective_grows_with_the_mesh(max_components=6):
    """On the communicators of the last _run_domains call: no all-reduce beyond the replicated coarsest multigrid level
    (27 slots x 36 entries x 125 boxes at set-up, 750 numbers per application) and no neighbour exchange beyond the domain's
    interface (its shared node slots x 6 components, the nodal-stress halo sum) -- SURVEY 2a: the reference's whole-mesh
    MPI_AllReduce (src/main.F90:274) is what the partitioned paths replace."""
    assert LAST_STATS and all(st is not None for st in LAST_STATS)
    for st in LAST_STATS:
        assert st["allreduce_max_doubles"] <= 27 * 36 * 125, st
        assert st["exchange_max_doubles"] <= max_components * st["interface_slots"], st


@pytest.mark.parametrize("world,npr", [(2, (5, 4, 3)), (4, (3, 3, 4)), (6, (2, 3, 3)), (8, (3, 2, 3))])   # 8 = 2x2x2: the layout of config 4
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
        assert np.abs(s - sg[gid]).max() <= 1e-10 * np.abs(sg).max()
        assert abs(its - gits[0]) <= 3
        assert np.abs(y.reshape(-1, 3) - yg[gid]).max() <= 1e-12 * np.abs(yg).max()
    # all sharers hold bitwise-identical interface values (canonical-order halo sums)
    for r in range(world):
        for q in range(r + 1, world):
            common, ir, iq = np.intersect1d(out[r][5], out[q][5], return_indices=True)
            if common.size:
                assert np.array_equal(out[r][0].reshape(-1, 3)[ir], out[q][0].reshape(-1, 3)[iq])


def test_rccl_single_rank_collectives(oracle):
    """RCCL back end on the one GPU of this box: librccl is dlopen'ed, a 1-rank communicator is
    created from a unique id, and (FEM_AMD_FORCE_COLLECTIVES) every CG scalar goes through
    ncclAllReduce on the solve's stream.  Only ncclSend/ncclRecv need a second GPU."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from fem_amd import capi, driver
from fem_amd.mesh import structured_hex
from oracle.oracle import Oracle
ctx = capi.Context(0)
comm = capi.Comm.rccl(ctx, 1, 0, capi.Comm.unique_id())
assert comm.rank_size() == (0, 1)
m = structured_hex(6, 5, 4, jitter=0.2)
empty = (np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32))
prob = driver.Problem(ctx, m, "-ksp_rtol 1e-13", comm=comm, halo=empty)
prob.assemble(); prob.ksp.set_up(); prob.dmesh.form_rhs(prob.F, 0.0, m.dt); prob.ksp.solve(prob.F, prob.U)
u = prob.U.get_array(); s = prob.dmesh.recover_nodal_stress(prob.U)
uo, so, its = Oracle().run(m, "cg", rtol=1e-13)
assert prob.ksp.get_converged_reason() > 0 and abs(prob.ksp.get_iteration_number() - its[0]) <= 2
assert np.abs(u - uo).max() <= 1e-10 * np.abs(uo).max() and np.abs(s - so).max() <= 1e-10 * np.abs(so).max()
prob.destroy(); comm.destroy(); ctx.destroy()
print("RCCL_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEM_AMD_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_rccl_send_recv_primitive_on_one_rank():
    """ncclSend / ncclRecv as the halo exchange issues them (one group, fp64 counts, the solve's
    stream), here between rank 0 and itself -- the only peer a 1-GPU box has.  Two "neighbours" (both
    rank 0) with different segment lengths check offsets and the group semantics."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from fem_amd import capi
ctx = capi.Context(0)
comm = capi.Comm.rccl(ctx, 1, 0, capi.Comm.unique_id())
n = 1000
send, recv = capi.Vec(ctx, n), capi.Vec(ctx, n)
x = np.arange(n, dtype=np.float64) * 0.5 + 1.0
send.set_array(x); recv.set_array(np.zeros(n))
comm.exchange([0], [0, 700], send, recv)
got = recv.get_array()
assert np.array_equal(got[:700], x[:700]) and not got[700:].any(), got[:8]
send.destroy(); recv.destroy(); comm.destroy(); ctx.destroy()
print("SENDRECV_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "SENDRECV_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_rccl_grouped_allreduce_and_send_recv_on_one_rank():
    """RcclComm::exchange_allreduce -- ncclAllReduce inside the ncclGroup of the halo exchange's sends and receives, what
    `-ksp_fuse_allreduce 1` issues once per CG iteration -- on real RCCL with the one rank (itself as the peer) a 1-GPU box has
    (round-3 advice: the grouped call had only ever run through the in-process fallback, which does the two one after the other)."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
from fem_amd import capi
ctx = capi.Context(0)
comm = capi.Comm.rccl(ctx, 1, 0, capi.Comm.unique_id())
n = 1000
send, recv, red = capi.Vec(ctx, n), capi.Vec(ctx, n), capi.Vec(ctx, 4)
x = np.arange(n, dtype=np.float64) * 0.25 - 3.0
send.set_array(x); recv.set_array(np.zeros(n)); red.set_array(np.array([1.5, -2.0, 7.0, 9.0]))
for rep in range(3):
    comm.exchange_allreduce([0], [0, 600], send, recv, red, 2)
got, r = recv.get_array(), red.get_array()
assert np.array_equal(got[:600], x[:600]) and not got[600:].any(), got[:8]
assert np.array_equal(r, [1.5, -2.0, 7.0, 9.0]), r            # one rank: the sum over all ranks is the value itself
st = comm.stats()
assert st["allreduce_calls"] == 3 and st["exchange_calls"] == 3 and st["exchange_max_doubles"] == 600
comm.exchange_allreduce([], [0], send, recv, red, 4)            # a domain without neighbours still takes part in the reduction
send.destroy(); recv.destroy(); red.destroy(); comm.destroy(); ctx.destroy()
print("GROUPED_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "GROUPED_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_exchange_primitive_local_group():
    """The same primitive through the in-process communicator: three domains, everybody a neighbour of
    everybody; domain r must receive, in its segment for q, what q put into its segment for r."""
    world, seg = 3, 6
    ctxs = [capi.Context(0) for _ in range(world)]
    comms = capi.Comm.local_group(ctxs)
    out, err = [None] * world, []
    nbs = [[q for q in range(world) if q != r] for r in range(world)]

    def work(r):
        try:
            send, recv = capi.Vec(ctxs[r], 2 * seg), capi.Vec(ctxs[r], 2 * seg)
            send.set_array(100.0 * r + np.arange(2 * seg)); recv.set_array(np.zeros(2 * seg))
            comms[r].exchange(nbs[r], [0, seg, 2 * seg], send, recv)
            out[r] = recv.get_array()
            send.destroy(); recv.destroy()
        except Exception as e:  # noqa: BLE001
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not err, err
    for r in range(world):
        for i, q in enumerate(nbs[r]):
            j = nbs[q].index(r)
            assert np.array_equal(out[r][i * seg:(i + 1) * seg], 100.0 * q + np.arange(j * seg, (j + 1) * seg))
    for c in comms:
        c.destroy()
    for c in ctxs:
        c.destroy()


def _unstructured_case(kind):
    """A deck-like mesh with every load type: clamped edge, nodal forces over a time window, tractions."""
    from fem_amd.mesh import hex_to_tets, quad_to_mixed, structured_quad
    if kind == "mixed2d":
        m = quad_to_mixed(structured_quad(9, 6, jitter=0.15))
        pd = 2
    else:
        m = hex_to_tets(structured_hex(4, 3, 3, jitter=0.1))
        m.fnode = np.array([m.nnds - 1, m.nnds - 2], dtype=np.int32)
        m.fval = np.array([[1.0e6, -2.0e6, 5.0e5, 0.0, 1.0], [0.0, 3.0e6, 0.0, 0.0, 0.5]])
        pd = 3
    # tractions on one side of several elements (side numbering src/m_elems.F90:286,370,473)
    els = np.arange(0, m.nels, max(1, m.nels // 7))[:6]
    m.telsd = np.stack([els, 1 + (els % 3)], axis=1).astype(np.int32)
    tv = np.zeros((len(els), pd + 2)); tv[:, 0] = 2.0e5; tv[:, 1] = -1.0e5; tv[:, pd] = 0.0; tv[:, pd + 1] = 1.0
    m.tval = tv
    m.t, m.dt = 1.0, 0.5        # two increments
    return m


@pytest.mark.parametrize("kind,world", [("mixed2d", 3), ("tet3d", 4)])
def test_general_partition_matches_serial_oracle(oracle, kind, world):
    """RCB element partition of an unstructured mixed mesh (the reference's METIS step), interface
    lists from the local->global maps, all load types, two increments; vs the serial oracle."""
    m = _unstructured_case(kind)
    epart = partition.rcb_partition(m, world)
    parts = [partition.split_mesh(m, epart, r) for r in range(world)]
    nl2g = [p[1] for p in parts]
    ctxs = [capi.Context(0) for _ in range(world)]
    comms = capi.Comm.local_group(ctxs)
    out, err = [None] * world, []

    def work(r):
        try:
            lm = parts[r][0]
            prob = driver.Problem(ctxs[r], lm, "-ksp_rtol 1e-13", comm=comms[r], halo=partition.general_halo(r, nl2g))
            prob.assemble(); prob.ksp.set_up()
            u = np.zeros(lm.ndof); s = np.zeros((lm.nnds, lm.cpdim))
            for step in range(2):
                prob.dmesh.form_rhs(prob.F, lm.dt * step, lm.dt)
                prob.ksp.solve(prob.F, prob.U)
                assert prob.ksp.get_converged_reason() > 0
                u += prob.U.get_array()
                s += prob.dmesh.recover_nodal_stress(prob.U)
            out[r] = (u, s)
            prob.destroy()
        except Exception as e:  # noqa: BLE001
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not err, err
    assert all(o is not None for o in out)
    ug, sg, its = oracle.run(m, "cg", rtol=1e-13)
    assert len(its) == 2
    pd = m.pdim
    for r in range(world):
        u, s = out[r]
        assert np.abs(u.reshape(-1, pd) - ug.reshape(-1, pd)[nl2g[r]]).max() <= 1e-10 * np.abs(ug).max()
        assert np.abs(s - sg[nl2g[r]]).max() <= 1e-10 * np.abs(sg).max()
    for c in comms:
        c.destroy()
    for c in ctxs:
        c.destroy()


def test_allreduce_inside_the_halo_group_gives_the_same_fields():
    """-ksp_fuse_allreduce 1 (the CG's p.Kp all-reduce and the halo exchange of w through Comm::exchange_allreduce: one RCCL
    group on real peers, one after the other in the in-process communicator) against the default schedule: same iteration
    counts, bitwise the same fields."""
    a = _run_domains(4, (5, 4, 3), 0.1, "-mat_mult_type ebe -ksp_rtol 1e-12")[1]
    b = _run_domains(4, (5, 4, 3), 0.1, "-mat_mult_type ebe -ksp_rtol 1e-12 -ksp_fuse_allreduce 1")[1]
    for (ua, sa, ia, ra, ya, ga), (ub, sb, ib, rb, yb, gb) in zip(a, b):
        assert ia == ib and ra == rb > 0
        assert np.array_equal(ua, ub) and np.array_equal(sa, sb)


def test_streaming_cg_on_domains_gives_the_same_fields():
    """-ksp_stream_min_dof 0 forces what problems beyond the Infinity Cache get (x and K p with the streaming hint) on four small
    domains: bitwise the fields of the default kernels."""
    a = _run_domains(4, (5, 4, 3), 0.1, "-mat_mult_type ebe -ksp_rtol 1e-12")[1]
    b = _run_domains(4, (5, 4, 3), 0.1, "-mat_mult_type ebe -ksp_rtol 1e-12 -ksp_stream_min_dof 0")[1]
    for (ua, sa, ia, ra, ya, ga), (ub, sb, ib, rb, yb, gb) in zip(a, b):
        assert ia == ib and ra == rb > 0
        assert np.array_equal(ua, ub) and np.array_equal(sa, sb)
