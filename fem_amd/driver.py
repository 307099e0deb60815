"""`main.F90`-shaped driver over the C-ABI: assemble K once, then the increment
loop of src/main.F90:216-296 (FormRHS -> KSPSolve -> GetVec_U -> stress
recovery -> running totals).  Host logic only; all arithmetic runs in
libfem_amd.so.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import capi
from .mesh import Mesh


@dataclass
class RunResult:
    aggregate_u: np.ndarray            # [nnds*pdim]   (aggregate_u, src/m_global.F90:65)
    aggregate_stress: np.ndarray       # [nnds, cpdim] (aggregate_stress)
    its: list = field(default_factory=list)
    reasons: list = field(default_factory=list)
    rnorms: list = field(default_factory=list)
    t_symbolic: float = 0.0
    t_assemble: float = 0.0
    t_solve: float = 0.0
    t_stress: float = 0.0
    nnz: int = 0


class Problem:
    """Device objects of one domain: mesh, Mat_K, Vec_F, Vec_U, Krylov."""

    def __init__(self, ctx: capi.Context, mesh: Mesh, options: str = "", comm: capi.Comm | None = None, halo=None):
        self.ctx, self.mesh = ctx, mesh
        self.dmesh = capi.DeviceMesh(ctx, mesh)
        if comm is not None and halo is not None:
            self.dmesh.set_halo(comm, *halo)
        self.K: capi.Mat | None = None
        self.F = capi.Vec(ctx, mesh.ndof)
        self.U = capi.Vec(ctx, mesh.ndof)
        self.ksp = capi.KSP(ctx)
        # SetupKSPSolver: rtol = 1.e-9 (src/main.F90:349-351), other tolerances PETSc defaults,
        # then KSPSetFromOptions (:377)
        self.ksp.set_type("cg")
        self.ksp.set_pc_type("jacobi")
        self.ksp.set_tolerances(rtol=capi.REFERENCE_RTOL)
        if options:
            self.ksp.set_from_options(options)

    def create_matrix(self):
        """MatCreateAIJ (src/main.F90:154) -- symbolic phase."""
        self.K = capi.Mat(self.dmesh)

    def assemble(self):
        """src/main.F90:157-171: element loop + MatAssembly + ApplyKBC."""
        if self.K is None:
            self.create_matrix()
        self.K.zero_entries()
        self.K.assemble_elements()
        self.K.apply_kbc()
        self.ksp.set_operators(self.K)

    def destroy(self):
        self.ksp.destroy(); self.F.destroy(); self.U.destroy()
        if self.K is not None:
            self.K.destroy()
        self.dmesh.destroy()


def run(mesh: Mesh, options: str = "", ctx: capi.Context | None = None, recover_stress: bool = True) -> RunResult:
    own = ctx is None
    ctx = ctx or capi.Context(0)
    prob = Problem(ctx, mesh, options)
    res = RunResult(np.zeros(mesh.ndof), np.zeros((mesh.nnds, mesh.cpdim)))
    t0 = time.perf_counter()
    prob.create_matrix()
    ctx.synchronize()
    t1 = time.perf_counter()
    prob.assemble()
    prob.ksp.set_up()
    ctx.synchronize()
    t2 = time.perf_counter()
    res.t_symbolic, res.t_assemble = t1 - t0, t2 - t1
    res.nnz = prob.K.info()[1]
    step = 0
    while True:
        t_init = mesh.dt * step               # t_init = dt*(dtNo-1); exit when t_init >= t (src/main.F90:218-219)
        if t_init >= mesh.t:
            break
        ta = time.perf_counter()
        prob.dmesh.form_rhs(prob.F, t_init, mesh.dt)
        prob.ksp.solve(prob.F, prob.U)
        ctx.synchronize()
        tb = time.perf_counter()
        reason = prob.ksp.get_converged_reason()
        res.its.append(prob.ksp.get_iteration_number())
        res.reasons.append(reason)
        res.rnorms.append(prob.ksp.get_residual_norm())
        if reason < 0:
            raise RuntimeError(f"KSP diverged at increment {step + 1}: reason {reason}")
        res.aggregate_u += prob.U.get_array()                      # aggregate_u = aggregate_u + uu (:250)
        res.t_solve += tb - ta
        if recover_stress:
            tc = time.perf_counter()
            res.aggregate_stress += prob.dmesh.recover_nodal_stress(prob.U)   # (:291)
            res.t_stress += time.perf_counter() - tc
        step += 1
    prob.destroy()
    if own:
        ctx.destroy()
    return res
