#!/usr/bin/env python3
"""tools/config5_profile.py [copies] -- where the time of a config-5 run goes: cProfile of the Python driver loop over the C-ABI calls."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fem_amd import capi, driver  # noqa: E402
from fem_amd.inp import read_inp  # noqa: E402
from fem_amd.mesh import tile_deck  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = tile_deck(read_inp(os.path.join(ROOT, "tests", "golden", "decks", "cohesive_test_2.inp")), copies)
ctx = capi.Context(0)
driver.run_nonlinear(m, "-ksp_type preonly -pc_type lu", ctx=ctx)
pr = cProfile.Profile()
pr.enable()
driver.run_nonlinear(m, "-ksp_type preonly -pc_type lu", ctx=ctx)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
