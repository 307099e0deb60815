"""GPU: the tile plan of `-mat_mult_type ebe` built on the device (ebe_plan.hip, what the product uses) against the host
builder kept as its cross-check (ebe_tile.hip): same plan, array by array, on the meshes of test_gpu_ebe.py, on seeded
random bricks (straight / sheared / turned / jittered / half-jittered / shuffled / graded), on partitioned bricks (tiles
that own shared nodes first) and on config 2 / config 3; the device plan is the same from one build to the next."""
import numpy as np
import pytest

from fem_amd import capi, partition
from fem_amd.mesh import structured_hex, structured_quad

from test_gpu_ebe import _graded, _meshes, _shuffled

pytestmark = pytest.mark.gpu

ARRAYS = ("tiles", "lane_slots", "tile_nodes", "element_visits", "hdr", "steps", "nodes", "elems", "conn16", "flags", "lists",
          "geometry")


def _digests(ctx, m, halo=None, comm=None):
    dm = capi.DeviceMesh(ctx, m)
    if halo is not None:
        dm.set_halo(comm, *halo)
    K = capi.Mat(dm)                      # the symbolic phase supplies the node -> element lists
    dev = dm.tile_plan_digest("device")
    dev2 = dm.tile_plan_digest("device")
    host = dm.tile_plan_digest("host")
    K.destroy(); dm.destroy()
    return dev, dev2, host


def _same(a, b):
    return [k for k in ARRAYS if a[k] != b[k]]


@pytest.mark.parametrize("name", [n for n, m in _meshes().items() if n.startswith("hex") or n.startswith("quad")])
def test_device_plan_equals_host_plan(gpu_ctx, name):
    dev, dev2, host = _digests(gpu_ctx, _meshes()[name])
    assert dev["tiles"] > 0
    assert _same(dev, dev2) == []         # reproducible
    assert _same(dev, host) == []


def _soak_mesh(seed):
    rng = np.random.default_rng(1000 + seed)
    dims = tuple(int(v) for v in rng.integers(2, [48, 30, 22]))
    kind = ["straight", "sheared", "turned", "jitter", "half_jitter", "shuffled", "graded"][seed % 7]
    m = structured_hex(*dims, jitter=0.15 if kind == "jitter" else 0.0)
    if kind == "sheared":
        M = np.eye(3) + 0.4 * rng.standard_normal((3, 3))
        if abs(np.linalg.det(M)) < 0.3:
            M = np.eye(3) + np.diag([0.2, -0.1, 0.3])
        m.coords = np.ascontiguousarray(m.coords @ M.T + rng.standard_normal(3))
    elif kind == "turned":
        Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        m.coords = np.ascontiguousarray((m.coords * np.array([1.0, 0.7, 1.3])) @ Q.T)
    elif kind == "half_jitter":
        j = structured_hex(*dims, jitter=0.2)
        keep = m.coords[:, 0] < 0.5 * dims[0]
        m.coords = np.where(keep[:, None], m.coords, j.coords)
    elif kind == "shuffled":
        m = _shuffled(m, seed=seed)
    elif kind == "graded":
        c = m.coords.copy()
        c[:, 0] = dims[0] * (np.expm1(c[:, 0] / max(c[:, 0].max(), 1.0) * 2.0) / np.expm1(2.0))
        c[:, 2] = dims[2] * (c[:, 2] / max(c[:, 2].max(), 1.0)) ** 2
        m.coords = np.ascontiguousarray(c)
    return kind, dims, m


@pytest.mark.parametrize("seed", range(28))
def test_device_plan_equals_host_plan_on_seeded_bricks(gpu_ctx, seed):
    kind, dims, m = _soak_mesh(seed)
    dev, dev2, host = _digests(gpu_ctx, m)
    assert dev["tiles"] > 0, (kind, dims)
    assert _same(dev, dev2) == [], (kind, dims)
    assert _same(dev, host) == [], (kind, dims)


def test_device_plan_of_a_partitioned_brick_lists_interface_tiles_first(gpu_ctx):
    """four in-process domains: every domain's plan (tiles that own shared nodes first) from both builders"""
    import threading
    grid, npr, world = (2, 2, 1), (20, 17, 9), 4
    ctxs = [capi.Context(0) for _ in range(world)]
    comms = capi.Comm.local_group(ctxs)
    out, err = [None] * world, []

    def work(r):
        try:
            m = partition.brick_mesh(r, grid, npr)
            out[r] = _digests(ctxs[r], m, halo=partition.brick_halo(r, grid, npr), comm=comms[r])
        except Exception as e:  # noqa: BLE001
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not err, err
    for c in comms:
        c.destroy()
    for c in ctxs:
        c.destroy()
    plain = _digests(gpu_ctx, partition.brick_mesh(0, grid, npr))[0]
    for r, (dev, dev2, host) in enumerate(out):
        assert _same(dev, dev2) == [] and _same(dev, host) == [], r
    assert out[0][0]["lists"] != plain["lists"]      # (the interface tiles moved to the front)
    assert out[0][0]["hdr"] == plain["hdr"]


def test_device_plan_at_config_2_and_3(gpu_ctx):
    for m in (structured_quad(707), structured_hex(149, 149, 149)):
        dev, dev2, host = _digests(gpu_ctx, m)
        assert _same(dev, dev2) == []
        assert _same(dev, host) == []
        assert dev2["setup_ms"] < 25.0         # (the host builder: 120-340 ms at config 3; the first build also pays the allocations)
        print(f"{m.ndof} dof: {dev['tiles']} tiles, device builder {dev['setup_ms']:.1f} ms (again: {dev2['setup_ms']:.1f} ms), host builder {host['setup_ms']:.0f} ms")
