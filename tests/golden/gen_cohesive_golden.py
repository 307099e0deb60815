#!/usr/bin/env python3
"""tests/golden/gen_cohesive_golden.py -- golden vectors of the reference's cohesive helper routines
(getCohValues, getCohRels, getCohGaps src/m_elems.F90:658-720; Seplaw_1_Tract/Stiff src/m_seplaw.F90),
produced by the reference's own modules as compiled by oracle/Makefile (`make ref`).  Run in the build
container (needs /root/reference); writes tests/golden/cohesive_golden.json: inputs and outputs only."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import RefLib  # noqa: E402

ref = RefLib()
rng = np.random.default_rng(20261003)
cases = []
# the property sets the reference decks use (examples/cohesive_test_2.inp:26, cohesive_test_microns.inp) and random ones
prop_sets = [[100.0, 0.01, 0.01, 1.0, 0.0, 0.0], [100.0, 0.01, 0.01, 1.0, 0.0, 0.05]]
for k in range(24):
    if k < 2:
        props = prop_sets[k]
    else:
        props = [float(rng.uniform(10, 1e3)), float(rng.uniform(1e-3, 5e-2)), float(rng.uniform(1e-3, 5e-2)),
                 float(rng.uniform(0.3, 1.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.0, 0.1))]
    ang = rng.uniform(0, 2 * np.pi)
    L = rng.uniform(0.5, 2.0)
    a = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1)])
    tdir = np.array([np.cos(ang), np.sin(ang)])
    # nodes 1,2 on the lower face, 3,4 on the upper (3 over 2, 4 over 1), zero thickness + small perturbation
    ec = np.stack([a, a + L * tdir, a + L * tdir, a]) + rng.uniform(-1e-3, 1e-3, (4, 2))
    u8 = rng.uniform(-1, 1, 8) * props[1] * 2.0
    dt = float(rng.choice([0.01, 0.1, 1.0]))
    t, n, det = ref.coh_values(ec)
    gps = []
    for gp in range(2):
        ur, vr = ref.coh_rels(u8, gp, dt)
        g, vg = ref.coh_gaps(t, n, ur, vr)
        tr, st = ref.seplaw1(props, g, vg, dt)
        gps.append({"urel": ur.tolist(), "vrel": vr.tolist(), "gap": g.tolist(), "vgap": vg.tolist(),
                    "tract": tr.tolist(), "stiff": st.tolist()})
    cases.append({"ecoords": ec.tolist(), "u8": u8.tolist(), "props": props, "dt": dt,
                  "tangent": t.tolist(), "normal": n.tolist(), "det": det, "gp": gps})
out = {"generator": "tests/golden/gen_cohesive_golden.py",
       "source": "reference modules m_utils/m_elems/m_seplaw compiled by oracle/Makefile (make ref), AMD flang",
       "cases": cases}
with open(os.path.join(ROOT, "tests", "golden", "cohesive_golden.json"), "w") as f:
    json.dump(out, f, indent=0)
print("wrote", len(cases), "cases")
