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


class NonlinearProblem(Problem):
    """The reference's state for a deck with cohesive elements: Mat_K, its duplicate Jacobian, Residual, the SNES with
    its two callbacks (src/main.F90:197-206) and the running total aggregate_u the callbacks add to the iterate."""

    def __init__(self, ctx: capi.Context, mesh: Mesh, options: str = "", rels_mode: int = 0, rels_nodes=None, builtin_callbacks: bool = False):
        super().__init__(ctx, mesh, "")
        self.builtin_callbacks = builtin_callbacks
        self.rels_mode, self.rels_nodes = rels_mode, rels_nodes
        self.dmesh.set_cohesive()                          # before MatCreateAIJ: the sparsity holds the cohesive pairs
        self.create_matrix()
        self.assemble()                                    # Mat_K with ApplyKBC (src/main.F90:157-171)
        self.J = self.K.duplicate()                        # MatDuplicate(Mat_K, MAT_COPY_VALUES, Jacobian) :197
        self.R = capi.Vec(ctx, mesh.ndof)                  # VecDuplicate(Vec_U, Residual) :198
        self.agg = capi.Vec(ctx, mesh.ndof)                # aggregate_u (device copy for the callbacks)
        self.utot = capi.Vec(ctx, mesh.ndof)
        self.snes = capi.SNES(ctx)                         # SNESCreate :200
        self.snes_ksp = self.snes.get_ksp()                # SNESGetKSP + SetupKSPSolver :201-202
        self.snes_ksp.set_type("preonly"); self.snes_ksp.set_pc_type("lu")          # KSPPREONLY + PCLU (MUMPS), :355-358
        self.snes_ksp.set_tolerances(rtol=capi.REFERENCE_RTOL)                       # :349-351
        if builtin_callbacks:
            # the same two callbacks, issued by the library itself (fem_snes_set_defmod_callbacks): no interpreter between the primitives
            self.snes.set_defmod_callbacks(self.dmesh, self.K, self.J, self.R, self.F, self.agg, self.utot, mesh.dt, rels_mode, rels_nodes)
        else:
            self.snes.set_function(self.R, self.calc_residual)                       # :204
            self.snes.set_jacobian(self.J, self.calc_jacobian)                       # :205
        if options:
            self.snes.set_from_options(options)                                      # SNESSetFromOptions :206

    def calc_jacobian(self, du: capi.Vec, J: capi.Mat):
        """CalcJacobian, src/m_global.F90:98-158."""
        self.K.copy_to(J)                                  # MatCopy(Mat_K, Jacobian) :123
        du.copy_to(self.utot); self.utot.axpy(1.0, self.agg)          # uu = uu + aggregate_u :127
        J.add_cohesive(self.utot, self.mesh.dt, self.rels_mode, self.rels_nodes)    # applyStiff_1 + MatSetValue(ADD) :129-147
        J.apply_kbc()                                      # ApplyKBC(Jacobian) :153

    def calc_residual(self, du: capi.Vec, R: capi.Vec):
        """CalcResidual, src/m_global.F90:161-235: Residual = Jacobian*du - (Vec_F + cohesive tractions, BC rows)."""
        du.copy_to(self.utot); self.utot.axpy(1.0, self.agg)          # current_u = current_u + aggregate_u :182
        self.F.copy_to(R)                                  # VecCopy(Vec_F, Residual) :184
        self.dmesh.add_cohesive_traction(self.utot, self.mesh.dt, R, self.rels_mode, self.rels_nodes)   # :186-205
        self.dmesh.enforce_bc_force(R, self.mesh.dt)       # EnforceBCForce :210 (and again :215)
        R.scale(-1.0)                                      # VecScale(Residual, MINUS) :223
        self.J.mult_add(du, R, R)                          # MatMultAdd(Jacobian, du, Residual, Residual) :224

    def destroy(self):
        self.snes.destroy()
        for o in (self.R, self.agg, self.utot, self.J):
            o.destroy()
        super().destroy()


def run_nonlinear(mesh: Mesh, options: str = "", ctx: capi.Context | None = None, rels_mode: int = 0, rels_nodes=None,
                  recover_stress: bool = True, builtin_callbacks: bool = False) -> RunResult:
    """The increment loop of src/main.F90:216-296 on a deck with cohesive elements (the `ncohmats /= 0` branch)."""
    own = ctx is None
    ctx = ctx or capi.Context(0)
    prob = NonlinearProblem(ctx, mesh, options, rels_mode, rels_nodes, builtin_callbacks)
    res = RunResult(np.zeros(mesh.ndof), np.zeros((mesh.nnds, mesh.cpdim)))
    res.snes_its, res.snes_reasons, res.fnorms, res.U_hist = [], [], [], []
    step = 0
    while True:
        t_init = mesh.dt * step
        if t_init >= mesh.t:
            break
        ta = time.perf_counter()
        prob.dmesh.form_rhs(prob.F, t_init, mesh.dt)       # FormRHS :236
        if builtin_callbacks:
            prob.snes.calc_defmod_jacobian(prob.U)
        else:
            prob.calc_jacobian(prob.U, prob.J)             # CalcJacobian(.., Vec_U, Jacobian, Jacobian) :241
        prob.snes.solve(prob.U)                            # SNESSolve :242 (Vec_U keeps the previous increment)
        ctx.synchronize()
        res.t_solve += time.perf_counter() - ta
        res.snes_its.append(prob.snes.get_iteration_number())          # :243
        res.snes_reasons.append(prob.snes.get_converged_reason())
        res.fnorms.append(prob.snes.get_function_norm())
        res.its.append(prob.snes.get_linear_solve_iterations())
        u = prob.U.get_array()
        res.U_hist.append(u)
        res.aggregate_u += u                               # aggregate_u = aggregate_u + uu :250
        prob.agg.set_array(res.aggregate_u)
        if recover_stress:
            res.aggregate_stress += prob.dmesh.recover_nodal_stress(prob.U)
        step += 1
        if res.snes_reasons[-1] < 0:
            break
    prob.destroy()
    if own:
        ctx.destroy()
    return res


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
