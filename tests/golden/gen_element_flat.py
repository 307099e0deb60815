#!/usr/bin/env python3
"""tests/golden/gen_element_flat.py -- flattens tests/golden/element_golden.json (made by gen_golden.py from the
reference's compiled element modules) into a list-directed text file a Fortran program can read:

    ncases
    per case:  type-name  n  pdim  nip
               E nu
               ecoords(n,pdim) column-major | edisp(n*pdim) | xi(pdim)

(the expected outputs stay in the JSON: the Fortran program prints what it computed, the test compares)."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
g = json.load(open(os.path.join(HERE, "element_golden.json")))
lines, n = [], 0
for name in ("tri", "qua", "tet", "hex"):
    t = g["types"][name]
    for c in t["cases"]:
        ec = c["ecoords"]
        nn, pd = len(ec), len(ec[0])
        lines.append(f"{name} {nn} {pd} {t['nip']}")
        lines.append(f"{c['E']!r} {c['nu']!r}")
        lines.append(" ".join(repr(ec[a][d]) for d in range(pd) for a in range(nn)))
        lines.append(" ".join(repr(x) for x in c["edisp"]))
        lines.append(" ".join(repr(x) for x in c["xi"]))
        n += 1
with open(os.path.join(HERE, "element_cases.txt"), "w") as f:
    f.write(f"{n}\n" + "\n".join(lines) + "\n")
print("wrote", n, "cases")
