"""fem_amd -- MI355X-native assemble-and-solve hot path of the defmod FEM code.

Host-side mirror of the reference's element / Mat / Vec / KSP surface over the
C-ABI library `libfem_amd.so` (include/fem_amd.h).  The HIP library is
required: nothing here falls back to a CPU path.
"""
from .mesh import Mesh, structured_hex, structured_quad, TRI, QUA, TET, HEX  # noqa: F401
from .inp import read_inp, parse_inp  # noqa: F401

__all__ = ["Mesh", "structured_hex", "structured_quad", "read_inp", "parse_inp", "TRI", "QUA", "TET", "HEX"]
