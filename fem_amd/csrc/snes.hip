// fem_amd/csrc/snes.hip -- Newton line-search solver behind the reference's SNES calls.
//
// The reference solves every increment of a deck with cohesive elements through PETSc's SNES:
//   SNESCreate / SNESGetKSP / SNESSetType(SNESNEWTONLS) / SNESSetFunction(CalcResidual) / SNESSetJacobian(CalcJacobian)
//   / SNESSetFromOptions (src/main.F90:200-206), SNESSolve + SNESGetIterationNumber (:242-243).
// PETSc is not part of the reference tree; what is built here is its published algorithm (PETSc 3.5:
// src/snes/impls/ls/ls.c SNESSolve_NEWTONLS, src/snes/interface/snesut.c SNESConvergedDefault,
// src/snes/linesearch/impls/bt/linesearchbt.c, cubic backtracking) with PETSc's defaults (the tests hold a CPU
// restatement of the same loop against it).  The two callbacks stay with the host (as in the reference, where
// they are Fortran routines of module `global`): they receive the iterate and fill the residual / the Jacobian with
// the library's device primitives (fem_mat_copy, fem_mat_add_cohesive, fem_vec_add_cohesive_traction, ...).
// All vectors live on the device; the scalars of the line search are reduced there and read back -- a Newton
// iteration costs a factorisation or a Krylov solve, the handful of round trips does not matter.
#include <chrono>
#include <cmath>
#include <sstream>

#include "objects.h"

using namespace fem;

extern "C" {
typedef int (*fem_snes_function_cb_t)(fem_snes, fem_vec, fem_vec, void*);
typedef int (*fem_snes_jacobian_cb_t)(fem_snes, fem_vec, fem_mat, fem_mat, void*);
}

struct fem_snes_s {
    fem_ctx ctx = nullptr;
    fem_ksp ksp = nullptr;          // owned (SNESGetKSP hands it out)
    fem_vec r = nullptr;            // the caller's residual vector (SNESSetFunction)
    fem_mat J = nullptr, P = nullptr;
    fem_snes_function_cb_t fcb = nullptr;
    fem_snes_jacobian_cb_t jcb = nullptr;
    void *fctx = nullptr, *jctx = nullptr;
    // SNESCreate defaults
    double rtol = 1.0e-8, abstol = 1.0e-50, stol = 1.0e-8;
    int max_its = 50, max_funcs = 10000;
    std::string linesearch = "bt";
    // SNESLineSearch (bt) defaults
    double ls_alpha = 1.0e-4, ls_maxstep = 1.0e8, ls_minlambda = 1.0e-12;
    int ls_max_its = 40;
    // results
    int iter = 0, reason = 0, nfuncs = 0, linear_its = 0;
    double fnorm = 0.0;
    fem_vec Y = nullptr, W = nullptr, G = nullptr;
    DBuf<double> red;               // partials and results of dots3 (kept: no block request per line-search trial)
    double* h_red = nullptr;        // pinned, coherent: 3 doubles and the ticket of the read-back that wrote them
    unsigned long long red_ticket = 0;
};

// Vec helpers of ksp.hip / capi.cpp
double vec_dot(fem_vec, fem_vec);
void vec_axpy(fem_vec, double, fem_vec);
void vec_waxpy(fem_vec, double, fem_vec, fem_vec);
void mat_mult_raw(fem_mat, const double*, double*, double*, const int*);

namespace {

// up to three dot products in one pass and one read-back (the scalars of a line-search trial): partials per workgroup, then one
// workgroup adds them in a fixed order
__global__ void __launch_bounds__(256) k_dots3_partial(size_t n, int npairs, const double* __restrict__ a0, const double* __restrict__ b0,
                                                       const double* __restrict__ a1, const double* __restrict__ b1,
                                                       const double* __restrict__ a2, const double* __restrict__ b2, double* __restrict__ part)
{
    __shared__ double sh[4];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        s0 += a0[i] * b0[i];
        if (npairs > 1) s1 += a1[i] * b1[i];
        if (npairs > 2) s2 += a2[i] * b2[i];
    }
    double v[3] = {s0, s1, s2};
    for (int q = 0; q < 3; q++) {
        double x = v[q];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
        __syncthreads();
        if (threadIdx.x == 0) part[(size_t)q * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    }
}
// host_out (pinned, coherent host memory: 3 doubles and a ticket): the sums go straight to the host, the ticket behind a system-scope
// fence -- the host polls the ticket instead of enqueueing a copy and waiting for the stream (a copy kernel, its completion signal and
// the wake-up: 20-25 us of idle device per read-back of a line search that has two of them per trial point)
__global__ void __launch_bounds__(256) k_dots3_final(int nblocks, const double* __restrict__ part, double* __restrict__ out,
                                                     double* __restrict__ host_out, unsigned long long ticket)
{
    __shared__ double sh[4];
    for (int q = 0; q < 3; q++) {
        double x = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += 256) x += part[(size_t)q * nblocks + i];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
            out[q] = v;
            if (host_out) host_out[q] = v;
        }
    }
    if (host_out && threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + 3), ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int check(int rc)
{
    if (rc) throw Error(rc, fem_last_error());
    return rc;
}

void ensure_vec(fem_snes sn, fem_vec& v, long long n)
{
    if (v && v->n == n) return;
    if (v) check(fem_vec_destroy(v));
    check(fem_vec_create(sn->ctx, n, &v));
}

double norm2(fem_vec v) { return std::sqrt(vec_dot(v, v)); }

// (a0.b0, a1.b1, a2.b2) with one read-back; pairs beyond npairs are not touched
void dots3(fem_snes sn, int npairs, fem_vec a0, fem_vec b0, fem_vec a1, fem_vec b1, fem_vec a2, fem_vec b2, double (&out)[3])
{
    sn->ctx->bind();
    hipStream_t s = sn->ctx->stream;
    const size_t n = (size_t)a0->n;
    const int nb = (int)std::min<size_t>(256, std::max<size_t>(1, (n + 255) / 256));
    if (sn->red.n < (size_t)3 * 256 + 3) sn->red.alloc((size_t)3 * 256 + 3);
    if (!sn->h_red) {
        FEM_HIP(hipHostMalloc((void**)&sn->h_red, 4 * sizeof(double)));
        sn->h_red[3] = 0.0;   // (the ticket's bits: 0)
    }
    // FEM_AMD_SNES_POLL=0: the read-back as a copy and a stream synchronisation
    static const bool poll_env = !(std::getenv("FEM_AMD_SNES_POLL") && std::atoi(std::getenv("FEM_AMD_SNES_POLL")) == 0);
    static bool poll_works = true;   // (a ticket that did not arrive once: the copy and the stream from then on, not 200 ms per read-back)
    const bool poll = poll_env && poll_works;
    hipLaunchKernelGGL(k_dots3_partial, dim3(nb), dim3(256), 0, s, n, npairs, a0->d.p, b0->d.p, (a1 ? a1 : a0)->d.p, (b1 ? b1 : b0)->d.p,
                       (a2 ? a2 : a0)->d.p, (b2 ? b2 : b0)->d.p, sn->red.p);
    const unsigned long long ticket = ++sn->red_ticket;
    hipLaunchKernelGGL(k_dots3_final, dim3(1), dim3(256), 0, s, nb, sn->red.p, sn->red.p + (size_t)3 * 256, poll ? sn->h_red : (double*)nullptr, ticket);
    FEM_HIP(hipGetLastError());
    bool got = false;
    if (poll) {
        const volatile unsigned long long* tk = reinterpret_cast<const volatile unsigned long long*>(sn->h_red + 3);
        const auto t0 = std::chrono::steady_clock::now();
        for (long spins = 0;; spins++) {
            if (__atomic_load_n(tk, __ATOMIC_ACQUIRE) == ticket) { got = true; break; }
            // (a device error, or a platform whose host memory the device does not write through: the stream tells)
            if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
    }
    if (!got) {
        if (poll) poll_works = false;
        FEM_HIP(hipMemcpyAsync(sn->h_red, sn->red.p + (size_t)3 * 256, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
        FEM_HIP(hipStreamSynchronize(s));
    }
    for (int q = 0; q < 3; q++) out[q] = sn->h_red[q];
}

void compute_function(fem_snes sn, fem_vec x, fem_vec f)
{
    sn->nfuncs++;
    const int rc = sn->fcb(sn, x, f, sn->fctx);
    FEM_CHECK(rc == 0, 20, "fem_snes_solve: the residual callback returned " + std::to_string(rc));
}

// SNESConvergedDefault
int converged(fem_snes sn, int it, double xnorm, double snorm, double fnorm, double& ttol)
{
    if (!it) ttol = fnorm * sn->rtol;
    if (std::isnan(fnorm) || std::isinf(fnorm)) return FEM_SNES_DIVERGED_FNORM_NAN;
    if (fnorm < sn->abstol) return FEM_SNES_CONVERGED_FNORM_ABS;
    if (sn->nfuncs >= sn->max_funcs) return FEM_SNES_DIVERGED_FUNCTION_COUNT;
    if (it) {
        if (fnorm <= ttol) return FEM_SNES_CONVERGED_FNORM_RELATIVE;
        if (snorm < sn->stol * xnorm) return FEM_SNES_CONVERGED_SNORM_RELATIVE;
    }
    return 0;
}

// w = x - lambda y
void step_to(fem_vec w, fem_vec x, double lambda, fem_vec y)
{
    vec_waxpy(w, -lambda, y, x);   // (one kernel, the roundings of the copy and the axpy it stands for)
}

// SNESLineSearchApply_BT, cubic.  X, F in/out; Y the direction with J Y = F.  Returns success.
bool linesearch_bt(fem_snes sn, fem_vec X, fem_vec F, fem_vec Y, double& xnorm, double& fnorm, double& ynorm)
{
    fem_vec W = sn->W, G = sn->G;
    double lambda = 1.0;
    // |Y|, |X| and the initial slope F . (J Y) in one pass and one read-back (J Y first: a step beyond the maximum -- 1e8 by
    // default -- scales Y and, the product being linear, the slope by the same factor)
    sn->J->ctx->bind();
    mat_mult_raw(sn->J, Y->d.p, W->d.p, nullptr, nullptr);
    double d3[3];
    dots3(sn, 3, Y, Y, X, X, F, W, d3);
    ynorm = std::sqrt(d3[0]); xnorm = std::sqrt(d3[1]);
    double initslope = d3[2];
    if (ynorm == 0.0) return false;
    if (ynorm > sn->ls_maxstep) { const double sc = sn->ls_maxstep / ynorm; check(fem_vec_scale(Y, sc)); initslope *= sc; ynorm = sn->ls_maxstep; }
    const double f = fnorm * fnorm;
    if (initslope > 0.0) initslope = -initslope;
    if (initslope == 0.0) initslope = -1.0;
    double wnorm2 = 0.0;     // |W|^2 of the trial point: |X| of the accepted step without another read-back
    auto trial_norms = [&](double& gn) { dots3(sn, 2, G, G, W, W, nullptr, nullptr, d3); gn = std::sqrt(d3[0]); wnorm2 = d3[1]; };
    step_to(W, X, lambda, Y);
    if (sn->nfuncs >= sn->max_funcs) return false;
    compute_function(sn, W, G);
    double gnorm;
    trial_norms(gnorm);
    double g = gnorm * gnorm;
    if (std::isnan(g) || std::isinf(g)) return false;
    if (!(.5 * g <= .5 * f + lambda * sn->ls_alpha * initslope)) {
        if (sn->stol * xnorm > ynorm) return false;   // the full step did not work and the step is tiny
        double lambdatemp = -initslope / (g - f - 2.0 * lambda * initslope);   // quadratic fit
        double lambdaprev = lambda, gprev = g;
        if (lambdatemp > .5 * lambda) lambdatemp = .5 * lambda;
        lambda = (lambdatemp <= .1 * lambda) ? .1 * lambda : lambdatemp;
        step_to(W, X, lambda, Y);
        if (sn->nfuncs >= sn->max_funcs) return false;
        compute_function(sn, W, G);
        trial_norms(gnorm); g = gnorm * gnorm;
        if (std::isnan(g) || std::isinf(g)) return false;
        if (!(.5 * g < .5 * f + lambda * sn->ls_alpha * initslope)) {
            int count = 0;
            for (; count < sn->ls_max_its; count++) {   // cubic fits
                if (lambda <= sn->ls_minlambda) return false;
                const double t1 = .5 * (g - f) - lambda * initslope;
                const double t2 = .5 * (gprev - f) - lambdaprev * initslope;
                const double a = (t1 / (lambda * lambda) - t2 / (lambdaprev * lambdaprev)) / (lambda - lambdaprev);
                const double b = (-lambdaprev * t1 / (lambda * lambda) + lambda * t2 / (lambdaprev * lambdaprev)) / (lambda - lambdaprev);
                double d = b * b - 3 * a * initslope;
                if (d < 0.0) d = 0.0;
                lambdatemp = (a == 0.0) ? -initslope / (2.0 * b) : (-b + std::sqrt(d)) / (3.0 * a);
                lambdaprev = lambda; gprev = g;
                if (lambdatemp > .5 * lambda) lambdatemp = .5 * lambda;
                lambda = (lambdatemp <= .1 * lambda) ? .1 * lambda : lambdatemp;
                step_to(W, X, lambda, Y);
                if (sn->nfuncs >= sn->max_funcs) return false;
                compute_function(sn, W, G);
                trial_norms(gnorm); g = gnorm * gnorm;
                if (std::isnan(g) || std::isinf(g)) return false;
                if (.5 * g < .5 * f + lambda * sn->ls_alpha * initslope) break;
            }
            if (count == sn->ls_max_its) return false;
        }
    }
    check(fem_vec_copy(W, X));
    check(fem_vec_copy(G, F));
    xnorm = std::sqrt(wnorm2);
    fnorm = gnorm;
    return true;
}

// -snes_linesearch_type basic: the full Newton step, no test
bool linesearch_basic(fem_snes sn, fem_vec X, fem_vec F, fem_vec Y, double& xnorm, double& fnorm, double& ynorm)
{
    ynorm = norm2(Y);
    vec_axpy(X, -1.0, Y);
    compute_function(sn, X, F);
    xnorm = norm2(X); fnorm = norm2(F);
    return !(std::isnan(fnorm) || std::isinf(fnorm));
}

}  // namespace

// SNESSolve_NEWTONLS
void snes_solve_impl(fem_snes sn, fem_vec X)
{
    FEM_CHECK(sn->fcb && sn->r, 7, "fem_snes_solve: no residual callback set (fem_snes_set_function)");
    FEM_CHECK(sn->jcb && sn->J, 7, "fem_snes_solve: no Jacobian callback set (fem_snes_set_jacobian)");
    FEM_CHECK(X && X->n == sn->r->n, 8, "fem_snes_solve: vector size mismatch");
    sn->ctx->bind();
    fem_vec F = sn->r;
    ensure_vec(sn, sn->Y, X->n); ensure_vec(sn, sn->W, X->n); ensure_vec(sn, sn->G, X->n);
    sn->iter = 0; sn->nfuncs = 0; sn->linear_its = 0; sn->reason = 0;
    compute_function(sn, X, F);
    double fnorm = norm2(F), xnorm = 0.0, ynorm = 0.0, ttol = 0.0;
    int reason = converged(sn, 0, 0.0, 0.0, fnorm, ttol);
    int failures = 0;
    for (int i = 0; i < sn->max_its && !reason; i++) {
        const int rc = sn->jcb(sn, X, sn->J, sn->P ? sn->P : sn->J, sn->jctx);   // SNESComputeJacobian
        FEM_CHECK(rc == 0, 20, "fem_snes_solve: the Jacobian callback returned " + std::to_string(rc));
        check(fem_ksp_set_operators(sn->ksp, sn->J));                            // KSPSetOperators(ksp, J, Jpre)
        check(fem_ksp_solve(sn->ksp, F, sn->Y));                                 // J Y = F
        int kreason = 0, lits = 0;
        check(fem_ksp_get_converged_reason(sn->ksp, &kreason));
        check(fem_ksp_get_iteration_number(sn->ksp, &lits));
        if (kreason < 0) { reason = FEM_SNES_DIVERGED_LINEAR_SOLVE; break; }     // maxLinearSolveFailures = 1
        sn->linear_its += lits;
        const bool ok = (sn->linesearch == "basic") ? linesearch_basic(sn, X, F, sn->Y, xnorm, fnorm, ynorm)
                                                    : linesearch_bt(sn, X, F, sn->Y, xnorm, fnorm, ynorm);
        if (sn->nfuncs >= sn->max_funcs) { reason = FEM_SNES_DIVERGED_FUNCTION_COUNT; break; }
        if (std::isnan(fnorm) || std::isinf(fnorm)) { reason = FEM_SNES_DIVERGED_FNORM_NAN; break; }
        if (!ok) {
            if (sn->stol * xnorm > ynorm) { reason = FEM_SNES_CONVERGED_SNORM_RELATIVE; break; }
            if (++failures >= 1) { reason = FEM_SNES_DIVERGED_LINE_SEARCH; break; }   // maxFailures = 1
        }
        sn->iter = i + 1;
        reason = converged(sn, i + 1, xnorm, ynorm, fnorm, ttol);
    }
    if (!reason) reason = FEM_SNES_DIVERGED_MAX_IT;
    sn->reason = reason;
    sn->fnorm = fnorm;
}

void snes_set_from_options_impl(fem_snes sn, const char* options)
{
    std::istringstream is(options ? options : "");
    std::string key, val, rest;
    while (is >> key) {
        if (key == "-snes_rtol") is >> sn->rtol;
        else if (key == "-snes_atol") is >> sn->abstol;
        else if (key == "-snes_stol") is >> sn->stol;
        else if (key == "-snes_max_it") is >> sn->max_its;
        else if (key == "-snes_max_funcs") is >> sn->max_funcs;
        else if (key == "-snes_type") { is >> val; FEM_CHECK(val == "newtonls" || val == "ls", 6, "unsupported -snes_type " + val + " (newtonls)"); }
        else if (key == "-snes_linesearch_type") {
            is >> val;
            FEM_CHECK(val == "bt" || val == "basic", 6, "unsupported -snes_linesearch_type " + val + " (bt|basic)");
            sn->linesearch = val;
        }
        else if (key == "-snes_linesearch_alpha") is >> sn->ls_alpha;
        else if (key == "-snes_linesearch_maxstep") is >> sn->ls_maxstep;
        else if (key == "-snes_linesearch_minlambda") is >> sn->ls_minlambda;
        else if (key == "-snes_linesearch_max_it") is >> sn->ls_max_its;
        else {   // everything else belongs to the inner KSP / PC (KSPSetFromOptions is called by SNESSetFromOptions)
            is >> val;
            rest += " " + key + " " + val;
        }
        FEM_CHECK(!is.fail(), 6, "fem_snes_set_from_options: bad value for " + key);
    }
    if (!rest.empty()) check(fem_ksp_set_from_options(sn->ksp, rest.c_str()));
}

fem_snes snes_new(fem_ctx ctx)
{
    std::unique_ptr<fem_snes_s> sn(new fem_snes_s);
    sn->ctx = ctx;
    check(fem_ksp_create(ctx, &sn->ksp));
    // SNES's KSP starts from KSPCreate defaults; the reference then runs SetupKSPSolver on it (src/main.F90:201-202):
    // KSPPREONLY + PCLU, rtol 1e-9 -- the host does the same through fem_snes_get_ksp
    return sn.release();
}

void snes_delete(fem_snes sn)
{
    if (!sn) return;
    if (sn->ctx) sn->ctx->bind();
    for (fem_vec* v : {&sn->Y, &sn->W, &sn->G})
        if (*v) (void)fem_vec_destroy(*v);
    if (sn->ksp) (void)fem_ksp_destroy(sn->ksp);
    if (sn->h_red) (void)hipHostFree(sn->h_red);
    delete sn;
}

void snes_set_function(fem_snes sn, fem_vec r, fem_snes_function_cb_t cb, void* c) { sn->r = r; sn->fcb = cb; sn->fctx = c; }
void snes_set_jacobian(fem_snes sn, fem_mat J, fem_mat P, fem_snes_jacobian_cb_t cb, void* c) { sn->J = J; sn->P = P; sn->jcb = cb; sn->jctx = c; }
fem_ksp snes_get_ksp(fem_snes sn) { return sn->ksp; }
void snes_get_result(fem_snes sn, int* its, int* reason, double* fnorm, int* lits, int* nfuncs)
{
    if (its) *its = sn->iter;
    if (reason) *reason = sn->reason;
    if (fnorm) *fnorm = sn->fnorm;
    if (lits) *lits = sn->linear_its;
    if (nfuncs) *nfuncs = sn->nfuncs;
}
