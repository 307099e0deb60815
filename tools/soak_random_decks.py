#!/usr/bin/env python3
"""tools/soak_random_decks.py -- 300 further seeded decks of tests/test_gpu_random_decks.py through every solver
option against the oracle (round 1: 0 failures in 2,100 solves)."""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_random_decks.py"))
t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
from fem_amd import capi, driver
from oracle.oracle import Oracle
o = Oracle(); ctx = capi.Context(0)
bad = []
for seed in range(40, 340):
    m = t.random_deck(seed)
    u, s, its = o.run(m, "lu")
    u2, _, _ = o.run(m, "cg", rtol=1e-14)
    if np.abs(u - u2).max() > 1e-10 * np.abs(u).max():
        continue   # ill-conditioned deck: the oracle's own two solvers disagree
    for opts in t.SOLVERS:
        try:
            res = driver.run(m, opts, ctx=ctx)
            eu = np.abs(res.aggregate_u - u).max() / max(np.abs(u).max(), 1e-300)
            es = np.abs(res.aggregate_stress - s).max() / max(np.abs(s).max(), 1e-300)
            ok = all(r > 0 for r in res.reasons) and eu <= 1e-9 and es <= 1e-8
        except Exception as e:  # noqa: BLE001
            ok = False; eu = es = -1; print("EXC", seed, opts, repr(e)[:200])
        if not ok:
            bad.append((seed, opts, eu, es))
print("decks 40..339 x", len(t.SOLVERS), "options; failures:", len(bad))
for b in bad[:20]:
    print(b)
