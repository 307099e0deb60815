// fem_amd/csrc/cg_device.h -- the device-resident state of KSPSolve_CG and the scalar logic every workgroup of its kernels evaluates
// (ksp.hip), shared with the tile product that opens an iteration itself (ebe_tile.hip, k_ebe_tile_q4x<true>).
// Reference semantics: PETSc KSPSolve_CG + KSPConvergedDefault (the reference selects them through `-ksp_type cg -pc_type jacobi`,
// src/main.F90:354,377).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fem_amd.h"
#include "grid_barrier.h"

namespace fem {

struct KspState {
    double beta, betaold, a, b, dp, rnorm0, ttol;
    double dpi2[2];  // p.w of the last two iterations (ping-pong: see the field table in ksp.hip)
    double red[4];   // reduced scalars: [0] z.z  [1] z.r  [2] p.w
    int it, its, reason, done;
    int stop;        // `done` as the x / r update saw it: what the kernel that opens an iteration may end on (see k_cg_update_p)
};

__device__ __forceinline__ double block_sum(double v, double* sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// What opens iteration i + 1 of KSPSolve_CG (i = iterations completed): KSPConvergedDefault on the preconditioned residual
// norm sqrt(z.z), then beta = z.r and b = beta / beta_old.  Pure function of its arguments: every workgroup of a kernel
// evaluates it from the same inputs.
struct CgOpen { double dp, rnorm0, ttol, b; int reason, its; };
__device__ __forceinline__ CgOpen cg_open_logic(int i, double zz, double zr, double betaold, double rnorm0_in, double ttol_in,
                                                double rtol, double abstol, double dtol, int maxit)
{
    CgOpen o;
    o.dp = sqrt(zz);
    o.rnorm0 = (i == 0) ? o.dp : rnorm0_in;
    o.ttol = (i == 0) ? fmax(rtol * o.dp, abstol) : ttol_in;
    o.reason = 0;
    if (isnan(o.dp) || isinf(o.dp)) o.reason = FEM_KSP_DIVERGED_NANORINF;
    else if (o.dp <= o.ttol) o.reason = (o.dp < abstol) ? FEM_KSP_CONVERGED_ATOL : FEM_KSP_CONVERGED_RTOL;
    else if (o.dp >= dtol * o.rnorm0) o.reason = FEM_KSP_DIVERGED_DTOL;
    else if (i >= maxit) o.reason = FEM_KSP_DIVERGED_ITS;
    o.its = -1;
    if (!o.reason) {
        o.its = i + 1;   // PETSc sets ksp->its = i+1 before the beta checks
        if (zr == 0.0) o.reason = FEM_KSP_CONVERGED_ATOL;
        else if (i > 0 && zr * betaold < 0.0) o.reason = FEM_KSP_DIVERGED_INDEFINITE_PC;
    }
    o.b = (i == 0) ? 0.0 : zr / betaold;
    return o;
}
// z = M^-1 r and p = z + b p with one rounding each, written the same way in every kernel that forms them (the compiler is
// free to contract a * b + c either way round; the kernels of one solve have to agree to the bit)
__device__ __forceinline__ double cg_z(double dinv, double r) { return __dmul_rn(dinv, r); }
__device__ __forceinline__ double cg_p(double z, double b, double p) { return __fma_rn(b, p, z); }

// The scalars of the convergence test and where they go (arguments of the kernels that open an iteration)
struct CgOpenArgs {
    KspState* st;
    const double *pzz, *pzr;   // partials of z.z and z.r (nullptr: already reduced in st->red[0], [1])
    int npart;
    double rtol, abstol, dtol;
    int maxit;
    double* hist;
    int hist_len;
};
struct CgOpened { int i, reason; double b, aprev; };
// Head of the kernel that opens iteration i + 1: every workgroup sums the partials in the same order and evaluates the same logic;
// workgroup 0 / thread 0 (`writer`) records the verdict in the KspState.  (256-thread workgroups; sh: 4 doubles of LDS.)
__device__ __forceinline__ CgOpened cg_open_block(const CgOpenArgs& c, bool writer, double* sh)
{
    KspState* st = c.st;
    // (the state first: its loads are in flight while the partials are summed -- the barriers of the sums would hold them back)
    const int i = st->it;
    const double betaold = st->betaold, rnorm0 = st->rnorm0, ttol = st->ttol, aprev = st->a;
    double zz, zr;
    if (c.pzz) {
        double s0 = 0.0, s1 = 0.0;
        for (int k = threadIdx.x; k < c.npart; k += 256) { s0 += c.pzz[k]; s1 += c.pzr[k]; }
        zz = block_sum(s0, sh);
        zr = block_sum(s1, sh);
    } else { zz = st->red[0]; zr = st->red[1]; }
    const CgOpen o = cg_open_logic(i, zz, zr, betaold, rnorm0, ttol, c.rtol, c.abstol, c.dtol, c.maxit);
    CgOpened r;
    r.i = i; r.reason = o.reason; r.b = o.b;
    r.aprev = aprev;   // alpha of the iteration just completed (x += a p is still owed; 0 before the first)
    if (writer) {
        st->dp = o.dp;
        if (c.hist && i < c.hist_len) c.hist[i] = o.dp;
        if (i == 0) { st->rnorm0 = o.rnorm0; st->ttol = o.ttol; }
        if (o.its >= 0) st->its = o.its;
        if (o.reason) { st->reason = o.reason; st->done = 1; }
        else { st->b = o.b; st->beta = zr; }
    }
    return r;
}

// The p update moved into the tile product (problems that sit in the caches: the product's stage-in forms p = M^-1 r + b p for
// the nodes of its tile -- rim nodes again in every tile that reads them -- writes it for the nodes it owns and applies the
// x += a p the previous iteration owes them): p alternates between two buffers, because the neighbours' rims still read the old one.
struct CgFuseArgs {
    CgOpenArgs open;
    const double *dinv, *r, *p_old;
    double *p_new, *x;
};

// KSPCG / PCJACOBI as one resident launch (ebe_tile.hip: k_cg_resident_q4)
struct CgResidentArgs {
    KspState* st;
    double rtol, abstol, dtol;
    int maxit;
    double* hist;
    int hist_len;
    const double* b;       // right-hand side
    const double* dinv;    // M^-1 (1 for -pc_type none)
    double* x;             // solution (written at the end)
    double* r;             // residual: owned nodes written every iteration (the neighbours' rims read it), final residual
    double *ppw, *pzz, *pzr;   // [ntiles] partials
    GridBarrier bar;
    int test_leave;            // tests (FEM_AMD_CG_RESIDENT_TEST_LEAVE=1): workgroup 1 leaves before the census -- the barrier must give up
};

}  // namespace fem
