#!/usr/bin/env python3
"""tools/assembly_timing.py [n] [jitter] [reps] -- times the element loop (k_assemble_rows + ApplyKBC) of config 3 and prints a
digest of the assembled values, so that two builds of the kernel can be compared bit for bit (round 3 used it for the
two-pass variant, profiles/r03_experiment_assembly_two_pass.txt)."""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
jitter = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = capi.Context(0)
m = structured_hex(n, jitter=jitter) if jitter else structured_hex(n)
prob = driver.Problem(ctx, m, "")
prob.create_matrix()
prob.assemble()
ctx.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    prob.assemble()
    ctx.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"n {n} jitter {jitter}: element loop + ApplyKBC "
      f"min {min(ts):.3f} ms, median {sorted(ts)[len(ts) // 2]:.3f} ms", flush=True)
if n <= 60:
    ptr, col, val = prob.K.get_csr()
    print("values sha1", hashlib.sha1(np.ascontiguousarray(val).tobytes()).hexdigest(), "nnz", len(val))
prob.destroy(); ctx.destroy()
