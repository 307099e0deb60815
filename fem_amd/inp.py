"""Reader for the reference's `.inp` deck format (current 8-integer header).

Follows /root/reference/src/m_io.F90: ReadParameters :12-20, ReadElementsCoords
:71-105, ReadDistMaterials :282-327, ReadDistForces :329-374 (t1,t2 clamped to
t at :357-358), ReadDistBcs :376-414, ReadDistTractions :416-475.  Fortran
list-directed reads: blank lines are skipped, tokens are separated by blanks or
commas, everything after the tokens a record needs (e.g. `! comment`) is
ignored, and a record may continue on following lines when short.
Node / element numbers are converted to 0-based.
"""
from __future__ import annotations

import numpy as np

from .mesh import Mesh, TYPE_NAMES, NODE_COUNT


class _Lines:
    def __init__(self, text: str):
        self.lines = []
        for ln in text.splitlines():
            ln = ln.split("!", 1)[0].replace(",", " ").strip()
            if ln:
                self.lines.append(ln.split())
        self.i = 0

    def record(self, n: int) -> list[str]:
        """Next record of at least n tokens (continuing over lines like a
        list-directed READ); surplus tokens on the last line are dropped."""
        toks: list[str] = []
        while len(toks) < n:
            if self.i >= len(self.lines):
                raise ValueError("unexpected end of deck")
            toks += self.lines[self.i]
            self.i += 1
        return toks[:n]

    def peek(self) -> str:
        return self.lines[self.i][0]


def _f(s: str) -> float:
    return float(s.replace("d", "e").replace("D", "e"))


def parse_inp(text: str) -> Mesh:
    L = _Lines(text)
    stype, pdim, nodal_bw = L.record(3)
    pdim = int(pdim)
    nels, nnds, nmts, ncohmats, nceqs, nfrcs, ntrcs, nbcs = (int(x) for x in L.record(8))
    if nceqs != 0:
        raise ValueError("nceqs must be 0 (cval is never allocated in the reference, m_global.F90:390-397)")
    t, dt = (_f(x) for x in L.record(2))
    eltype, eptr, conn, mat, nlmat = [], [0], [], [], []
    for _ in range(nels):
        name = L.peek().strip("'\"").lower()
        if name not in TYPE_NAMES:
            raise ValueError(f"unknown element type {name!r}")
        ty = TYPE_NAMES[name]
        n = NODE_COUNT[ty]
        rec = L.record(1 + n + 2)
        eltype.append(ty)
        conn += [int(x) - 1 for x in rec[1:1 + n]]
        eptr.append(len(conn))
        mat.append(int(rec[1 + n]) - 1)      # mat = 0 (-> -1) marks a cohesive element
        nlmat.append(int(rec[2 + n]))
    coords = np.array([[_f(x) for x in L.record(pdim)] for _ in range(nnds)], dtype=np.float64).reshape(nnds, pdim)
    mats = np.array([[_f(x) for x in L.record(5)] for _ in range(nmts)], dtype=np.float64).reshape(nmts, 5)
    cohmats = []
    for _ in range(ncohmats):
        seplaw = int(L.peek())
        nprops = {1: 6}[seplaw]                # PROP_COUNTS, m_seplaw.F90:8
        rec = L.record(1 + nprops)
        cohmats.append((seplaw, [_f(x) for x in rec[1:]]))
    bcnode = np.zeros((nbcs, 1 + pdim), dtype=np.int32)
    bcval = np.zeros((nbcs, pdim))
    for i in range(nbcs):
        rec = L.record(1 + 2 * pdim)
        bcnode[i, 0] = int(rec[0]) - 1
        bcnode[i, 1:] = [int(x) for x in rec[1:1 + pdim]]
        bcval[i] = [_f(x) for x in rec[1 + pdim:]]
    fnode = np.zeros(nfrcs, dtype=np.int32)
    fval = np.zeros((nfrcs, pdim + 2))
    for i in range(nfrcs):
        rec = L.record(1 + pdim + 2)
        fnode[i] = int(rec[0]) - 1
        fval[i] = [_f(x) for x in rec[1:]]
        fval[i, pdim] = min(fval[i, pdim], t)
        fval[i, pdim + 1] = min(fval[i, pdim + 1], t)
    telsd = np.zeros((ntrcs, 2), dtype=np.int32)
    tval = np.zeros((ntrcs, pdim + 2))
    for i in range(ntrcs):
        rec = L.record(2 + pdim + 2)
        telsd[i] = [int(rec[0]) - 1, int(rec[1])]
        tval[i] = [_f(x) for x in rec[2:]]
    return Mesh(pdim=pdim, eltype=eltype, eptr=eptr, conn=conn, mat_id=mat, coords=coords, mats=mats,
                bcnode=bcnode, bcval=bcval, fnode=fnode, fval=fval, telsd=telsd, tval=tval, t=t, dt=dt,
                stype=stype, nodal_bw=int(nodal_bw), nlmat=np.array(nlmat, dtype=np.int32), cohmats=cohmats)


def read_inp(path: str) -> Mesh:
    with open(path) as f:
        return parse_inp(f.read())


def write_inp(path: str, m: Mesh) -> None:
    """The deck of a Mesh in the format parse_inp reads (current 8-integer header; 17 significant digits, so that a
    deck written here and read back -- by this reader or by a Fortran list-directed READ -- carries the same numbers)."""
    names = {v: k for k, v in TYPE_NAMES.items()}

    def fl(x):
        return repr(float(x))

    out = [f"{m.stype} {m.pdim} {m.nodal_bw}",
           f"{m.nels} {m.nnds} {m.mats.shape[0]} {len(m.cohmats)} 0 {len(m.fnode)} {len(m.telsd)} {len(m.bcnode)}",
           f"{fl(m.t)} {fl(m.dt)}"]
    nlmat = m.nlmat if m.nlmat is not None else np.zeros(m.nels, dtype=np.int32)
    for e in range(m.nels):
        nodes = " ".join(str(int(v) + 1) for v in m.conn[m.eptr[e]:m.eptr[e + 1]])
        out.append(f"{names[int(m.eltype[e])]} {nodes} {int(m.mat_id[e]) + 1} {int(nlmat[e])}")
    for v in range(m.nnds):
        out.append(" ".join(fl(c) for c in m.coords[v]))
    for row in m.mats:
        out.append(" ".join(fl(c) for c in row))
    for law, props in m.cohmats:
        out.append(f"{law} " + " ".join(fl(c) for c in props))
    for i in range(len(m.bcnode)):
        out.append(f"{int(m.bcnode[i, 0]) + 1} " + " ".join(str(int(f)) for f in m.bcnode[i, 1:]) + " " + " ".join(fl(c) for c in m.bcval[i]))
    for i in range(len(m.fnode)):
        out.append(f"{int(m.fnode[i]) + 1} " + " ".join(fl(c) for c in m.fval[i]))
    for i in range(len(m.telsd)):
        out.append(f"{int(m.telsd[i, 0]) + 1} {int(m.telsd[i, 1])} " + " ".join(fl(c) for c in m.tval[i]))
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
