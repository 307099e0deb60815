#!/usr/bin/env python3
"""tools/snes_log_variants.py -- the audit behind "parity unpinned beyond one Newton step" (DESIGN.md section 2).

The reference holds one log of its Newton loop: examples/SNES_test/cohesive_test/tmp (a two-rank run of an OLDER revision: 7-integer
header, cohesive properties compiled in), with Vec_U printed after each of five increments and the SNES iteration counts 1, 1, 0, 0, 0,
and the two rank files {0,1}_output_000000.vtk beside it.  Increment 1 is reproduced by the oracle (and by the HIP path) to the six
printed digits with the legacy penalty scaling and rank 0's node set in getCohRels.  This script runs the oracle's restatement of the
loop under every combination of the readings that could distinguish that revision from the current sources --

  getCohRels nodes   rank 0's four (1,2,4,5) | the rank's first four (0,1,2,3) | the element's own
  penalty force      dt (legacy) | dt / t (current)
  variant bits       1 no aggregate_u in CalcJacobian, 2 none in CalcResidual, 4 Vec_U zeroed per increment,
                     8 prescribed values total instead of incremental, 16 CalcJacobian before SNESSolve skipped after the first increment

-- and prints, per combination, which increments' Vec_U (6 digits) and SNES counts match the log.  Run in the build container (CPU
only); the numbers of the log are read from tests/golden/reference_examples.json (data copied from the log), the deck is
tests/golden/decks/cohesive_test_legacy.inp."""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd.inp import read_inp  # noqa: E402
from oracle.oracle import Oracle, build  # noqa: E402

build(ref=False)
o = Oracle()
m = read_inp(os.path.join(ROOT, "tests", "golden", "decks", "cohesive_test_legacy.inp"))
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_examples.json")))["snes_log"]
log_u = np.array(ref["vec_u"])            # [5][16]
log_its = ref["snes_iterations"]          # [1, 1, 0, 0, 0]


def digits_match(a, b):
    """six printed digits: relative 5e-6 on the entries that are not round-off"""
    big = np.abs(b) > 1e-10
    return bool(np.all(np.abs(a[big] / b[big] - 1.0) < 5e-6) and np.all(np.abs(a[~big]) < 1e-10))


rows = []
for (rname, rmode, rnodes), legacy, var in itertools.product(
        (("rank 0's nodes", 0, (1, 2, 4, 5)), ("first four", 0, (0, 1, 2, 3)), ("element's own", 1, (0, 1, 2, 3))), (True, False), range(32)):
    try:
        r = o.snes_run(m, rels_mode=rmode, rels_nodes=rnodes, bc_legacy=legacy, variant=var)
    except Exception as e:  # noqa: BLE001
        rows.append((rname, legacy, var, f"error {e}"))
        continue
    scale = 1.0 if legacy else m.t      # (with the current dt / t scaling Vec_U comes out 1 / t times the log's: the same field)
    ok_u = [k < len(r["U_hist"]) and digits_match(r["U_hist"][k] * scale, log_u[k]) for k in range(5)]
    ok_i = [k < len(r["its"]) and int(r["its"][k]) == log_its[k] for k in range(5)]
    rows.append((rname, legacy, var, "".join("U" if a else "." for a in ok_u), "".join("i" if a else "." for a in ok_i),
                 list(map(int, r["its"])), list(map(int, r["reasons"]))))
print("getCohRels nodes | penalty force | variant bits | Vec_U of increments 1-5 matching the log | SNES counts matching | counts | reasons")
best = 0
for row in rows:
    if len(row) == 4:
        print(f"{row[0]:15s} | {'dt' if row[1] else 'dt/t':5s} | {row[2]:2d} | {row[3]}")
        continue
    n = 0
    while n < 5 and row[3][n] == "U":
        n += 1
    best = max(best, n)
    print(f"{row[0]:15s} | {'dt' if row[1] else 'dt/t':5s} | {row[2]:2d} | {row[3]} | {row[4]} | {row[5]} | {row[6]}")
print(f"\n{len(rows)} combinations; the longest run of leading increments any of them reproduces: {best} "
      f"(the log has 5; increment 1 alone = what tests/test_oracle_golden.py pins)")
