#!/usr/bin/env python3
"""tools/config5_strip_profile.py [copies] -- cProfile of the Python driver loop of config 5's connected strip (where the host side of a
Newton increment spends its time: callbacks, C-ABI calls, read-backs)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd import capi, driver  # noqa: E402
from fem_amd.inp import read_inp  # noqa: E402
from fem_amd.mesh import tile_deck_connected  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = tile_deck_connected(read_inp(os.path.join(ROOT, "tests", "golden", "decks", "cohesive_test_2.inp")), copies)
ctx = capi.Context(0)
for _ in range(2):
    driver.run_nonlinear(m, "-ksp_type preonly -pc_type lu", ctx=ctx, rels_mode=1)
pr = cProfile.Profile()
pr.enable()
driver.run_nonlinear(m, "-ksp_type preonly -pc_type lu", ctx=ctx, rels_mode=1)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
