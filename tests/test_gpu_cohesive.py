"""GPU: cohesive elements at element level -- fem_coh_apply_1 against the oracle, which is pinned
to the reference's compiled helper routines (tests/golden/cohesive_golden.json)."""
import json
import os

import numpy as np
import pytest

from fem_amd import capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cohesive_batch_matches_oracle_and_golden(gpu_ctx, oracle):
    d = json.load(open(os.path.join(HERE, "golden", "cohesive_golden.json")))
    cases = d["cases"]
    # a ragged batch: the golden cases, then 300 random ones (more than one 128-thread workgroup)
    rng = np.random.default_rng(11)
    ec = [np.asarray(c["ecoords"]) for c in cases]
    u8 = [np.asarray(c["u8"]) for c in cases]
    props = [np.asarray(c["props"]) for c in cases]
    for _ in range(300):
        a = rng.uniform(-1, 1, 2); ang = rng.uniform(0, 2 * np.pi); L = rng.uniform(0.5, 2.0)
        td = np.array([np.cos(ang), np.sin(ang)])
        ec.append(np.stack([a, a + L * td, a + L * td, a]) + rng.uniform(-1e-3, 1e-3, (4, 2)))
        p = np.array([rng.uniform(10, 1e3), rng.uniform(1e-3, 5e-2), rng.uniform(1e-3, 5e-2), rng.uniform(0.3, 1.5),
                      rng.uniform(-0.5, 0.5), rng.uniform(0, 0.1)])
        props.append(p)
        u8.append(rng.uniform(-1, 1, 8) * p[1] * 2)
    dt = 0.1
    ecf = np.stack([np.asfortranarray(e).ravel(order="F") for e in ec])      # ecoords(4,2) column-major per element
    tr, K, gp = capi.coh_apply_1(gpu_ctx, ecf, np.stack(u8), np.stack(props), dt)
    for k in range(len(ec)):
        to, Ko = oracle.coh_apply_1(ec[k], u8[k], props[k], dt)
        assert np.abs(tr[k] - to).max() <= 1e-13 * max(np.abs(to).max(), 1e-300)
        assert np.abs(K[k] - Ko).max() <= 1e-13 * np.abs(Ko).max()
    # per-integration-point record against the reference's own numbers (golden cases were made with their own dt)
    for k, c in enumerate(cases):
        tr1, K1, gp1 = capi.coh_apply_1(gpu_ctx, ecf[k:k + 1], np.stack(u8[k:k + 1]), np.stack(props[k:k + 1]), c["dt"])
        for q, g in enumerate(c["gp"]):
            rec = gp1[0, q]
            ref = np.concatenate([g["urel"], g["vrel"], g["gap"], g["vgap"], g["tract"], np.asarray(g["stiff"]).ravel(order="F")])
            assert np.abs(rec - ref).max() <= 1e-13 * np.abs(ref).max()


def test_cohesive_empty_batch_and_bad_dt(gpu_ctx):
    tr, K, gp = capi.coh_apply_1(gpu_ctx, np.zeros((0, 8)), np.zeros((0, 8)), np.zeros((0, 6)), 0.1)
    assert tr.shape == (0, 8) and K.shape == (0, 8, 8)
    with pytest.raises(capi.FemError, match="dt must be positive"):
        capi.coh_apply_1(gpu_ctx, np.ones((1, 8)), np.zeros((1, 8)), np.ones((1, 6)), 0.0)
