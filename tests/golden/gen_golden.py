"""Generates tests/golden/element_golden.json from the REFERENCE's own element
routines (oracle/_ref/libdefmod_ref.so = /root/reference/src/m_{utils,elems,
local,seplaw}.F90 compiled by AMD flang, see oracle/Makefile).  Run in the build
container only (the reference does not travel):

    make -C oracle ref && python tests/golden/gen_golden.py

The JSON holds inputs and expected outputs only (data, no reference source).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import RefLib, TRI, QUA, TET, HEX  # noqa: E402

BASE = {
    TRI: [[0, 0], [1, 0], [0, 1.0]],
    QUA: [[0, 0], [1, 0], [1, 1], [0, 1.0]],
    TET: [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]],
    HEX: [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1.0]],
}
NAMES = {TRI: "tri", QUA: "qua", TET: "tet", HEX: "hex"}


def main():
    r = RefLib()
    out = {"generator": "tests/golden/gen_golden.py", "source": "reference element modules via oracle/_ref", "types": {}}
    rng = np.random.default_rng(20261003)
    for t in (TRI, QUA, TET, HEX):
        base = np.array(BASE[t], dtype=float)
        pd = base.shape[1]
        n, nip, dim, nps, vtk = r.counts(t)
        ip, w, shp, inv = r.tables(t, pd)
        rec = {"nodecount": n, "nip": nip, "dim": dim, "nps": nps, "vtkid": vtk,
               "ipoints": ip.tolist(), "weights": w.tolist(), "shape": shp.tolist(),
               "n2inv": None if inv is None else inv.tolist(), "cases": []}
        nsides = {TRI: 3, QUA: 4, TET: 4, HEX: 6}[t]
        for c in range(7):
            if c == 0:
                ec, E, nu = base.copy(), 3640.0, 0.3        # the reference's own example material
            else:
                ec = base * rng.uniform(0.5, 2.0) + rng.uniform(-0.2, 0.2, base.shape)
                E, nu = float(rng.uniform(1e3, 1e11)), float(rng.uniform(0.0, 0.45))
            u = rng.standard_normal(n * pd)
            k = r.form_el_ke(t, ec, E, nu, pd)
            s = r.calc_el_stress(t, ec, u, E, nu, pd)
            xi = ip[min(c, nip - 1)]
            dN, detj = r.form_dn_detj(t, xi, ec, pd)
            enodes = np.arange(10, 10 + n, dtype=np.int32)
            sides = []
            for side in range(1, nsides + 1):
                a, sn = r.edge_area_nodes(t, enodes, ec, side, pd)
                sides.append({"side": side, "area": a, "snodes": sn.tolist()})
            rec["cases"].append({"ecoords": ec.tolist(), "E": E, "nu": nu, "k": k.tolist(), "edisp": u.tolist(),
                                 "estress": s.tolist(), "xi": xi.tolist(), "dN": dN.tolist(), "detj": detj,
                                 "dmat": r.dmat(pd, E, nu).tolist(), "sides": sides,
                                 "indx": r.form_el_indx(enodes, pd).tolist(), "enodes": enodes.tolist()})
        out["types"][NAMES[t]] = rec
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "element_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
