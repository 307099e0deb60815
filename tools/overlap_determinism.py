#!/usr/bin/env python3
"""tools/overlap_determinism.py -- in-process multi-domain CG with the one-pass tile product: hash of the fields for
FEM_AMD_OVERLAP in {0, 1} x FEM_AMD_SELF_REDUCE in {0, 1}, each twice (the switches are read once per process: child
processes)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from tests.test_gpu_multidomain import _run_domains
grid, out = _run_domains(%d, %r, 0.1, "-mat_mult_type ebe -ksp_rtol 1e-12")
h = hashlib.sha256()
for (u, s, its, reason, y, gid) in out:
    assert reason > 0
    h.update(np.ascontiguousarray(u).tobytes()); h.update(np.ascontiguousarray(y).tobytes())
print("RESULT", [o[2] for o in out], h.hexdigest()[:16])
"""
for world, npr in ((4, (9, 8, 5)), (8, (5, 6, 5))):
    for sr in ("1", "0"):
        for ov in ("0", "1"):
            for rep in range(2):
                env = dict(os.environ, FEM_AMD_OVERLAP=ov, FEM_AMD_SELF_REDUCE=sr, HSA_ENABLE_IPC_MODE_LEGACY="0")
                p = subprocess.run([sys.executable, "-c", code % (ROOT, world, npr)], env=env, capture_output=True, text=True, timeout=600)
                line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
                print(f"world {world} self_reduce {sr} overlap {ov} run {rep}:", line[0] if line else p.stderr[-500:], flush=True)
