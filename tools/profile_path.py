#!/usr/bin/env python3
"""tools/profile_path.py -- runs every kernel of the hot path once per config so that
`rocprofv3 --kernel-trace --stats -- python3 tools/profile_path.py c3` has all of them
(bench.py times assemble+solve only; stress recovery and the 2-D kernels show up here)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fem_amd import capi, driver  # noqa: E402
from fem_amd.mesh import structured_hex, structured_quad  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    m = structured_hex(149) if which == "c3" else structured_quad(707)
    ctx = capi.Context(0)
    for rep in range(2):
        t0 = time.perf_counter()
        res = driver.run(m, "", ctx=ctx)
        print(which, "rep", rep, "dof", m.ndof, "its", res.its, "symbolic %.1f ms assemble %.1f ms solve %.1f ms stress %.1f ms total %.1f ms"
              % (res.t_symbolic * 1e3, res.t_assemble * 1e3, res.t_solve * 1e3, res.t_stress * 1e3,
                 (time.perf_counter() - t0) * 1e3), flush=True)
    ctx.destroy()


if __name__ == "__main__":
    main()
