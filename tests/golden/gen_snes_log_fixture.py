#!/usr/bin/env python3
"""tests/golden/gen_snes_log_fixture.py -- copies the NUMBERS of the reference's SNES log and of the two rank files beside it into
tests/golden/reference_examples.json ("snes_log"): Vec_U after each of the five increments, the SNES iteration counts, and the
points / displacements / stresses of {0,1}_output_000000.vtk.  Run in the build container (reads /root/reference)."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/examples/SNES_test/cohesive_test"
lines = open(os.path.join(REF, "tmp")).read().split("\n")
incs, cur = [], None
for ln in lines:
    if ln.strip().startswith("Interval:"):
        cur = {"vec": [], "its": None, "residual_evaluations": 0}
        incs.append(cur)
    elif cur is not None:
        if "Viewing the Residual" in ln:
            cur["residual_evaluations"] += 1
        m = re.match(r"\s*SNES Iteration Count:\s*(\d+)", ln)
        if m:
            cur["its"] = int(m.group(1))
        try:
            cur["vec"].append(float(ln.strip()))
        except ValueError:
            pass


def vtk(path):
    t = open(path).read().split()
    out = {}
    i = t.index("POINTS"); n = int(t[i + 1])
    out["points"] = [float(x) for x in t[i + 3:i + 3 + 3 * n]]
    i = t.index("LOOKUP_TABLE"); k = i + 2
    ncomp = int(t[t.index("STRESS") + 2])
    out["stress"] = [float(x) for x in t[k:k + ncomp * n]]
    i = t.index("displacements")
    out["displacements"] = [float(x) for x in t[i + 2:i + 2 + 3 * n]]
    return out


path = os.path.join(HERE, "reference_examples.json")
d = json.load(open(path))
d["snes_log"] = {"source": "examples/SNES_test/cohesive_test/tmp (Vec_U after every increment, 16 dofs in PETSc's global order; SNES "
                           "iteration counts; number of residual evaluations printed per increment) and {0,1}_output_000000.vtk",
                 "vec_u": [c["vec"] for c in incs], "snes_iterations": [c["its"] for c in incs],
                 "residual_evaluations": [c["residual_evaluations"] for c in incs],
                 "vtk_rank0": vtk(os.path.join(REF, "0_output_000000.vtk")), "vtk_rank1": vtk(os.path.join(REF, "1_output_000000.vtk"))}
json.dump(d, open(path, "w"), indent=1)
print("increments", len(incs), "its", d["snes_log"]["snes_iterations"], "evaluations", d["snes_log"]["residual_evaluations"])
