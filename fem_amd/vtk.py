"""Legacy ASCII VTK output in the reference's layout (WriteOutput,
/root/reference/src/m_io.F90:480-555): POINTS as F0.3, CELLS `n id..` (0-based),
CELL_TYPES vtk ids (src/m_elems.F90:108-120), `SCALARS STRESS FLOAT cpdim` and
`VECTORS displacements double` as F0.6.  With ``elements`` given, the file holds
that element subset with nodes renumbered in ascending global id, i.e. what one
MPI rank of the reference writes after its local renumbering (src/main.F90:49-73).
"""
from __future__ import annotations

import numpy as np

from .mesh import Mesh, TRI, QUA, TET, HEX, COH

VTK_ID = {TRI: 5, QUA: 9, TET: 10, HEX: 12, COH: 9}


def _f0(v: float, d: int) -> str:
    """Fortran F0.d edit descriptor: no leading zero; a negative value that rounds to zero keeps its sign
    (`-.000000`, as in the reference's own rank files)."""
    s = f"{v:.{d}f}"
    if s.startswith("0."):
        return s[1:]
    if s.startswith("-0."):
        return "-" + s[2:]
    return s


def vtk_text(mesh: Mesh, u: np.ndarray, stress: np.ndarray, elements=None) -> str:
    pd, cp = mesh.pdim, mesh.cpdim
    els = np.arange(mesh.nels) if elements is None else np.asarray(elements)
    used = np.unique(np.concatenate([mesh.element_nodes(e) for e in els])) if len(els) else np.zeros(0, dtype=int)
    g2l = {int(g): i for i, g in enumerate(used)}
    u = np.asarray(u).reshape(-1, pd)
    out = ["# vtk DataFile Version 2.0", "File written by Defmod", "ASCII", "DATASET UNSTRUCTURED_GRID",
           f"POINTS {len(used)} double"]
    for g in used:
        p = list(mesh.coords[g]) + [0.0] * (3 - pd)
        out.append(" ".join(_f0(x, 3) for x in p))      # (3(F0.3,1X)): no blank after the last value
    ncol = sum(len(mesh.element_nodes(e)) + 1 for e in els)
    out.append(f"CELLS {len(els)} {ncol}")
    for e in els:
        nd = mesh.element_nodes(e)
        out.append(" ".join([str(len(nd))] + [str(g2l[int(v)]) for v in nd]))
    out.append(f"CELL_TYPES {len(els)}")
    out += [str(VTK_ID[int(mesh.eltype[e])]) for e in els]
    out.append(f"POINT_DATA {len(used)}")
    out.append(f"SCALARS STRESS FLOAT {cp}")
    out.append("LOOKUP_TABLE DEFAULT")
    for g in used:
        out.append(" ".join(_f0(x, 6) for x in stress[g]))
    out.append("VECTORS displacements double")
    for g in used:
        p = list(u[g]) + [0.0] * (3 - pd)
        out.append(" ".join(_f0(x, 6) for x in p))
    return "\n".join(out) + "\n"


def write_vtk(path: str, mesh: Mesh, u, stress, elements=None) -> None:
    with open(path, "w") as f:
        f.write(vtk_text(mesh, u, stress, elements))
