#!/usr/bin/env python3
"""tools/asm_only.py [n] [reps] -- config-3 mesh, symbolic phase once, then `reps` numeric assemblies (zero + element
loop + ApplyKBC) and one nodal-stress recovery: the kernels rocprofv3 passes over k_assemble_rows / k_stress look at."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi  # noqa: E402
from fem_amd.mesh import structured_hex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = capi.Context(0)
m = structured_hex(n)
dm = capi.DeviceMesh(ctx, m)
t0 = time.perf_counter()
K = capi.Mat(dm)
ctx.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    K.zero_entries(); K.assemble_elements(); K.apply_kbc()
ctx.synchronize()
t2 = time.perf_counter()
U = capi.Vec(ctx, m.ndof)
U.set_array(np.sin(0.001 * np.arange(m.ndof)))
S = capi.Vec(ctx, m.nnds * m.cpdim)
dm.recover_nodal_stress_vec(U, S) if hasattr(dm, "recover_nodal_stress_vec") else dm.recover_nodal_stress(U)
ctx.synchronize()
print(f"symbolic {1e3 * (t1 - t0):.2f} ms, assemble+bc {1e3 * (t2 - t1) / reps:.2f} ms per rep", flush=True)
for o in (U, S, K, dm):
    o.destroy()
ctx.destroy()
