/* oracle/snes_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's nonlinear increment (BASELINE config 5): the SNES callbacks
 * CalcJacobian (src/m_global.F90:98-158) and CalcResidual (:161-235) around the cohesive element
 * (applyStiff_1 / applyTract_1, restated in defmod_oracle.c), and the driver lines src/main.F90:200-206,240-250.
 *
 * The Newton iteration itself lives in PETSc (SNESNEWTONLS with the default `bt` line search, KSPPREONLY + PCLU
 * through MUMPS for the linear solves), which is neither vendored in /root/reference nor installed here.  It is
 * restated from PETSc 3.5's published algorithm (src/snes/impls/ls/ls.c SNESSolve_NEWTONLS,
 * src/snes/interface/snesut.c SNESConvergedDefault, src/snes/linesearch/impls/bt/linesearchbt.c) with PETSc's
 * defaults: rtol 1e-8, abstol 1e-50, stol 1e-8, 50 iterations, 10000 function evaluations; line search: cubic
 * backtracking, alpha 1e-4, maximum step 1e8, minimum lambda 1e-12, 40 trial steps.  PARITY of this loop: the first
 * increment of the reference's own two-rank log examples/SNES_test/cohesive_test/tmp (:20-37: Vec_U to six digits,
 * round-off entries of 1e-32 included, SNES iteration count 1) is reproduced exactly (tests/test_oracle_golden.py) --
 * that pins Mat_K, the cohesive tangent at zero opening, ApplyKBC, the penalty right-hand side and the Newton step.
 * The later increments of that log come from an older revision of the reference (EnforceBCForce scaled by dt instead
 * of dt/t, cohesive properties compiled in; what else differed is not recorded) and could not be reproduced by any
 * reading of the current sources: beyond one Newton step the loop is "parity unpinned" -- see DESIGN.md.
 *
 * What the reference does and this file repeats on purpose:
 *   - the "residual" is J du - F with whatever Jacobian the last CalcJacobian call left behind (:224);
 *   - getCohRels reads four fixed local nodes (`rels_nodes`: the first four nodes of the rank) instead of the
 *     element's own (src/m_elems.F90:697-704); rels_mode 1 switches to the element's nodes;
 *   - cohesive elements contribute nothing to Mat_K (FormElKE("coh") integrates an uninitialised dN in the
 *     reference -- undefined behaviour that cannot be matched) and are left out of the stress recovery.
 * MUMPS is run by the reference with null-pivot detection (ICNTL(24)=1, CNTL(3)=1e-6, src/main.F90:368-371): the
 * dense LU stand-in fixes a pivot below 1e-6 * max|A| the same way (its unknown becomes 0).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "defmod_oracle.h"

typedef struct {
    const dm_problem *p;
    const dm_cohesive *c;
    int N;
    long long nnz;
    int *rowptr, *colind;
    double *K, *J;          /* Mat_K (with ApplyKBC) and the Jacobian */
    double *F;              /* Vec_F of the increment */
    double *agg;            /* aggregate_u */
    double dt;
    int nfuncs;
    int step;               /* increment number, 0-based (variant 8 of dm_cohesive) */
} snes_ctx;

static int find_col(const int *colind, int lo, int hi, int c)
{
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (colind[mid] < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static void coh_inputs(const snes_ctx *s, int e, const double *utot, double *ec, double *u8, const double **props)
{
    const dm_problem *p = s->p;
    const int *nd = p->conn + p->eptr[e];
    int a, d;
    for (a = 0; a < 4; a++)
        for (d = 0; d < 2; d++) ec[a + 4 * d] = p->coords[(size_t)nd[a] * 2 + d];
    for (a = 0; a < 4; a++) {
        const int v = s->c->rels_mode ? nd[a] : s->c->rels_nodes[a];
        u8[2 * a] = utot[2 * v]; u8[2 * a + 1] = utot[2 * v + 1];
    }
    *props = s->c->props + (size_t)(s->c->nlmat[e] - 1) * 6;   /* cohMats(el%nlMat) */
}

static void bc_force(const snes_ctx *s, double *R)
{   /* EnforceBCForce, src/m_global.F90:439-455 (bc_legacy: the older revision behind the SNES_test log scaled by dt only) */
    const dm_problem *p = s->p;
    int i, d;
    for (i = 0; i < p->nbcs; i++)
        for (d = 0; d < p->pdim; d++)
            if (p->bcnode[(size_t)i * (1 + p->pdim) + 1 + d] == 0) {
                const double v = p->bcval[(size_t)i * p->pdim + d];
                R[(size_t)p->pdim * p->bcnode[(size_t)i * (1 + p->pdim)] + d] =
                    (s->c->bc_legacy ? DM_PENALTY_PARAM * v * s->dt : DM_PENALTY_PARAM * v * s->dt / p->t) *
                    ((s->c->variant & 8) ? (double)(s->step + 1) : 1.0);
            }
}

/* CalcJacobian, src/m_global.F90:98-158 */
static void calc_jacobian(snes_ctx *s, const double *du)
{
    const dm_problem *p = s->p;
    double *utot = (double *)malloc(sizeof(double) * (size_t)s->N);
    int e, i;
    memcpy(s->J, s->K, sizeof(double) * (size_t)s->nnz);               /* MatCopy(Mat_K, Jacobian) */
    for (i = 0; i < s->N; i++) utot[i] = du[i] + ((s->c->variant & 1) ? 0.0 : s->agg[i]);   /* uu = uu + aggregate_u */
    for (e = 0; e < p->nels; e++) {
        double ec[8], u8[8], st[64];
        const double *props;
        int n1, n2, d1, d2;
        if (s->c->nlmat[e] == 0) continue;
        coh_inputs(s, e, utot, ec, u8, &props);
        dm_coh_apply_stiff_1(ec, u8, props, s->dt, st);
        for (n1 = 0; n1 < 4; n1++) for (n2 = 0; n2 < 4; n2++) for (d1 = 0; d1 < 2; d1++) for (d2 = 0; d2 < 2; d2++) {
            const int r = 2 * p->conn[p->eptr[e] + n1] + d1, cc = 2 * p->conn[p->eptr[e] + n2] + d2;
            const int pos = find_col(s->colind, s->rowptr[r], s->rowptr[r + 1], cc);
            s->J[pos] += st[(n1 * 2 + d1) + 8 * (n2 * 2 + d2)];      /* MatSetValue(..., ADD_VALUES) */
        }
    }
    dm_apply_kbc(p, s->rowptr, s->colind, s->J);                        /* ApplyKBC(Jacobian) */
    free(utot);
}

/* CalcResidual, src/m_global.F90:161-235 */
static void calc_residual(snes_ctx *s, const double *du, double *R)
{
    const dm_problem *p = s->p;
    double *cur = (double *)malloc(sizeof(double) * (size_t)s->N);
    int e, i, j, k;
    s->nfuncs++;
    for (i = 0; i < s->N; i++) cur[i] = du[i] + ((s->c->variant & 2) ? 0.0 : s->agg[i]);
    memcpy(R, s->F, sizeof(double) * (size_t)s->N);                     /* VecCopy(Vec_F, Residual) */
    for (e = 0; e < p->nels; e++) {
        double ec[8], u8[8], tr[8];
        const double *props;
        if (s->c->nlmat[e] == 0) continue;
        coh_inputs(s, e, cur, ec, u8, &props);
        dm_coh_apply_tract_1(ec, u8, props, s->dt, tr);
        for (j = 0; j < 4; j++)
            for (k = 0; k < 2; k++) R[2 * p->conn[p->eptr[e] + j] + k] += tr[2 * j + k];   /* ApplyNodalForce(.., .true.) */
    }
    bc_force(s, R);                                                     /* EnforceBCForce (twice in the reference) */
    for (i = 0; i < s->N; i++) {                                        /* VecScale(-1); MatMultAdd(Jacobian, du, R, R) */
        double acc = -R[i];
        for (k = s->rowptr[i]; k < s->rowptr[i + 1]; k++) acc += s->J[k] * du[s->colind[k]];
        R[i] = acc;
    }
    free(cur);
}

/* KSPPREONLY + PCLU (MUMPS with null-pivot detection) stand-in: dense LU, partial pivoting */
static int lu_solve(const snes_ctx *s, const double *b, double *x)
{
    const int n = s->N;
    double *A = (double *)calloc((size_t)n * n, sizeof(double)), amax = 0.0;
    int i, j, k;
    if (!A) return -1;
    for (i = 0; i < n; i++)
        for (k = s->rowptr[i]; k < s->rowptr[i + 1]; k++) A[(size_t)i * n + s->colind[k]] += s->J[k];
    for (i = 0; i < n; i++) x[i] = b[i];
    for (k = 0; k < n; k++) {
        int pv = k;
        double mx = fabs(A[(size_t)k * n + k]);
        for (i = k + 1; i < n; i++)
            if (fabs(A[(size_t)i * n + k]) > mx) { mx = fabs(A[(size_t)i * n + k]); pv = i; }
        if (k == 0) { for (i = 0; i < n * n; i++) if (fabs(A[i]) < 1e29 && fabs(A[i]) > amax) amax = fabs(A[i]); }
        if (mx <= 1e-6 * amax) {   /* null pivot (CNTL(3) relative to the largest non-penalty entry): the unknown is fixed to 0 */
            for (j = 0; j < n; j++) { A[(size_t)k * n + j] = 0.0; A[(size_t)j * n + k] = 0.0; }
            A[(size_t)k * n + k] = 1.0; x[k] = 0.0;
            continue;
        }
        if (pv != k) {
            double tmp;
            for (j = 0; j < n; j++) { tmp = A[(size_t)k * n + j]; A[(size_t)k * n + j] = A[(size_t)pv * n + j]; A[(size_t)pv * n + j] = tmp; }
            tmp = x[k]; x[k] = x[pv]; x[pv] = tmp;
        }
        for (i = k + 1; i < n; i++) {
            const double f = A[(size_t)i * n + k] / A[(size_t)k * n + k];
            if (f == 0.0) continue;
            for (j = k; j < n; j++) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
            x[i] -= f * x[k];
        }
    }
    for (i = n - 1; i >= 0; i--) {
        double sum = x[i];
        for (j = i + 1; j < n; j++) sum -= A[(size_t)i * n + j] * x[j];
        x[i] = sum / A[(size_t)i * n + i];
    }
    free(A);
    return 0;
}

/* KSPBCGS + PCJACOBI (left preconditioning, preconditioned residual norm, zero initial guess): what the HIP library
 * offers for the unsymmetric Jacobian; PETSc 3.5 src/ksp/ksp/impls/bcgs/bcgs.c restated.  Returns iterations or <0. */
static int bcgs_solve(const snes_ctx *s, const double *b, double *x, double rtol, double abstol, int maxit)
{
    const int n = s->N;
    double *R = (double *)malloc(sizeof(double) * 7 * (size_t)n);
    double *RP = R + n, *V = RP + n, *T = V + n, *S = T + n, *P = S + n, *dinv = P + n;
    double dp, rho, rhoold = 1.0, alpha = 1.0, omega = 1.0, omegaold = 1.0, ttol;
    int i, k, it;
#define MULT(xin, yout)                                                                   \
    for (i = 0; i < n; i++) { double a_ = 0.0; for (k = s->rowptr[i]; k < s->rowptr[i + 1]; k++) a_ += s->J[k] * (xin)[s->colind[k]]; (yout)[i] = dinv[i] * a_; }
    for (i = 0; i < n; i++) {
        double d = 0.0;
        for (k = s->rowptr[i]; k < s->rowptr[i + 1]; k++) if (s->colind[k] == i) { d = s->J[k]; break; }
        dinv[i] = (d != 0.0) ? 1.0 / d : 1.0;
    }
    for (i = 0; i < n; i++) { x[i] = 0.0; R[i] = dinv[i] * b[i]; RP[i] = R[i]; P[i] = 0.0; V[i] = 0.0; }
    dp = 0.0; for (i = 0; i < n; i++) dp += R[i] * R[i]; dp = sqrt(dp);
    ttol = fmax(rtol * dp, abstol);
    if (dp <= ttol) { free(R); return 0; }
    for (it = 0; it < maxit; it++) {
        double beta, d1 = 0.0, d2 = 0.0;
        rho = 0.0; for (i = 0; i < n; i++) rho += R[i] * RP[i];
        if (rho == 0.0) { free(R); return -5; }                        /* KSP_DIVERGED_BREAKDOWN */
        beta = (rho / rhoold) * (alpha / omegaold);
        for (i = 0; i < n; i++) P[i] = R[i] + beta * (P[i] - omegaold * V[i]);
        MULT(P, V);
        for (i = 0; i < n; i++) d1 += V[i] * RP[i];
        if (d1 == 0.0) { free(R); return -5; }
        alpha = rho / d1;
        for (i = 0; i < n; i++) S[i] = R[i] - alpha * V[i];
        MULT(S, T);
        d1 = 0.0; for (i = 0; i < n; i++) { d1 += S[i] * T[i]; d2 += T[i] * T[i]; }
        if (d2 == 0.0) {
            /* t is 0: if s is 0 too, the iterate is exact */
            for (i = 0; i < n; i++) x[i] += alpha * P[i];
            free(R); return it + 1;
        }
        omega = d1 / d2;
        for (i = 0; i < n; i++) { x[i] += alpha * P[i] + omega * S[i]; R[i] = S[i] - omega * T[i]; }
        dp = 0.0; for (i = 0; i < n; i++) dp += R[i] * R[i]; dp = sqrt(dp);
        rhoold = rho; omegaold = omega;
        if (dp <= ttol) { free(R); return it + 1; }
        if (isnan(dp) || isinf(dp)) { free(R); return -9; }
    }
#undef MULT
    free(R);
    return -3;
}

static double vnorm(int n, const double *a) { double s = 0.0; int i; for (i = 0; i < n; i++) s += a[i] * a[i]; return sqrt(s); }

/* SNESConvergedDefault */
static int snes_converged(int it, double xnorm, double snorm, double fnorm, double *ttol, const snes_ctx *s)
{
    const double rtol = 1e-8, abstol = 1e-50, stol = 1e-8;
    if (!it) *ttol = fnorm * rtol;
    if (isnan(fnorm) || isinf(fnorm)) return DM_SNES_DIVERGED_FNORM_NAN;
    if (fnorm < abstol) return DM_SNES_CONVERGED_FNORM_ABS;
    if (s->nfuncs >= 10000) return DM_SNES_DIVERGED_FUNCTION_COUNT;
    if (it) {
        if (fnorm <= *ttol) return DM_SNES_CONVERGED_FNORM_RELATIVE;
        if (snorm < stol * xnorm) return DM_SNES_CONVERGED_SNORM_RELATIVE;
    }
    return 0;
}

/* SNESLineSearchApply_BT (cubic).  X, F in/out; Y the Newton direction (J Y = F, step X - lambda Y).
 * Returns 1 on success, 0 on failure. */
static int linesearch_bt(snes_ctx *s, double *X, double *F, double *Y, double *W, double *G, double *xnorm, double *fnorm,
                         double *ynorm)
{
    const int n = s->N;
    const double alpha = 1e-4, maxstep = 1e8, minlambda = 1e-12, stol = 1e-8;
    const int max_its = 40;
    double lambda = 1.0, initslope = 0.0, f, g, gnorm, lambdatemp, lambdaprev, gprev;
    int i, k, count;
    *ynorm = vnorm(n, Y); *xnorm = vnorm(n, X);
    if (*ynorm == 0.0) return 0;
    if (*ynorm > maxstep) { for (i = 0; i < n; i++) Y[i] *= maxstep / (*ynorm); *ynorm = maxstep; }
    f = (*fnorm) * (*fnorm);
    for (i = 0; i < n; i++) {                                /* MatMult(jac, Y, W); initslope = F . W */
        double a = 0.0;
        for (k = s->rowptr[i]; k < s->rowptr[i + 1]; k++) a += s->J[k] * Y[s->colind[k]];
        initslope += F[i] * a;
    }
    if (initslope > 0.0) initslope = -initslope;
    if (initslope == 0.0) initslope = -1.0;
    for (i = 0; i < n; i++) W[i] = X[i] - lambda * Y[i];
    calc_residual(s, W, G);
    gnorm = vnorm(n, G); g = gnorm * gnorm;
    if (isnan(g) || isinf(g)) return 0;
    if (.5 * g <= .5 * f + lambda * alpha * initslope) {
        /* sufficient reduction with the full step */
    } else {
        if (stol * (*xnorm) > *ynorm) return 0;
        lambdatemp = -initslope / (g - f - 2.0 * lambda * initslope);
        lambdaprev = lambda; gprev = g;
        if (lambdatemp > .5 * lambda) lambdatemp = .5 * lambda;
        if (lambdatemp <= .1 * lambda) lambda = .1 * lambda; else lambda = lambdatemp;
        for (i = 0; i < n; i++) W[i] = X[i] - lambda * Y[i];
        calc_residual(s, W, G);
        gnorm = vnorm(n, G); g = gnorm * gnorm;
        if (isnan(g) || isinf(g)) return 0;
        if (.5 * g < .5 * f + lambda * alpha * initslope) {
            /* quadratically determined step */
        } else {
            for (count = 0; count < max_its; count++) {
                double t1, t2, a, b, d;
                if (lambda <= minlambda) return 0;
                t1 = .5 * (g - f) - lambda * initslope;
                t2 = .5 * (gprev - f) - lambdaprev * initslope;
                a = (t1 / (lambda * lambda) - t2 / (lambdaprev * lambdaprev)) / (lambda - lambdaprev);
                b = (-lambdaprev * t1 / (lambda * lambda) + lambda * t2 / (lambdaprev * lambdaprev)) / (lambda - lambdaprev);
                d = b * b - 3 * a * initslope;
                if (d < 0.0) d = 0.0;
                if (a == 0.0) lambdatemp = -initslope / (2.0 * b);
                else lambdatemp = (-b + sqrt(d)) / (3.0 * a);
                lambdaprev = lambda; gprev = g;
                if (lambdatemp > .5 * lambda) lambdatemp = .5 * lambda;
                if (lambdatemp <= .1 * lambda) lambda = .1 * lambda; else lambda = lambdatemp;
                for (i = 0; i < n; i++) W[i] = X[i] - lambda * Y[i];
                calc_residual(s, W, G);
                gnorm = vnorm(n, G); g = gnorm * gnorm;
                if (isnan(g) || isinf(g)) return 0;
                if (.5 * g < .5 * f + lambda * alpha * initslope) break;
            }
            if (count == max_its) return 0;
        }
    }
    memcpy(X, W, sizeof(double) * (size_t)n);
    memcpy(F, G, sizeof(double) * (size_t)n);
    *xnorm = vnorm(n, X);
    *fnorm = gnorm;
    return 1;
}

/* SNESSolve_NEWTONLS.  inner: 0 = KSPPREONLY + PCLU stand-in, 1 = KSPBCGS + PCJACOBI (rtol ksp_rtol). */
static int snes_solve(snes_ctx *s, double *X, int inner, double ksp_rtol, double ksp_abstol, int *its_out, double *fnorm_out,
                      int *lin_its)
{
    const int n = s->N, maxits = 50;
    double *F = (double *)malloc(sizeof(double) * 4 * (size_t)n), *Y = F + n, *W = Y + n, *G = W + n;
    double fnorm, xnorm = 0.0, ynorm = 0.0, ttol = 0.0;
    int i, reason, failures = 0;
    s->nfuncs = 0;
    *its_out = 0;
    calc_residual(s, X, F);
    fnorm = vnorm(n, F);
    reason = snes_converged(0, 0.0, 0.0, fnorm, &ttol, s);
    for (i = 0; i < maxits && !reason; i++) {
        int ok, l;
        calc_jacobian(s, X);                                            /* SNESComputeJacobian */
        if (inner == 0) { l = lu_solve(s, F, Y) == 0 ? 1 : -1; }
        else l = bcgs_solve(s, F, Y, ksp_rtol, ksp_abstol, 10000);
        if (l < 0) { reason = DM_SNES_DIVERGED_LINEAR_SOLVE; break; }
        if (lin_its) *lin_its += l;
        ok = linesearch_bt(s, X, F, Y, W, G, &xnorm, &fnorm, &ynorm);
        if (s->nfuncs >= 10000) { reason = DM_SNES_DIVERGED_FUNCTION_COUNT; break; }
        if (isnan(fnorm) || isinf(fnorm)) { reason = DM_SNES_DIVERGED_FNORM_NAN; break; }
        if (!ok) {
            if (1e-8 * xnorm > ynorm) { reason = DM_SNES_CONVERGED_SNORM_RELATIVE; break; }
            if (++failures >= 1) { reason = DM_SNES_DIVERGED_LINE_SEARCH; break; }
        }
        *its_out = i + 1;
        reason = snes_converged(i + 1, xnorm, ynorm, fnorm, &ttol, s);
    }
    if (!reason) reason = DM_SNES_DIVERGED_MAX_IT;
    *fnorm_out = fnorm;
    free(F);
    return reason;
}

/* The increment loop src/main.F90:216-296 on a deck with cohesive elements (ncohmats /= 0 branch, :240-245).
 * U_hist[nsteps*N] (may be NULL) receives Vec_U after every increment; its / reasons per increment.
 * aggregate_stress(nnds,cpdim) as in dm_run (cohesive elements left out).  Returns the number of increments. */
int dm_snes_run(const dm_problem *p, const dm_cohesive *c, int inner, double ksp_rtol, double ksp_abstol, double *aggregate_u,
                double *aggregate_stress, double *U_hist, int *its, int *reasons, double *fnorms, int maxsteps)
{
    snes_ctx s;
    const int N = p->nnds * p->pdim, cp = (p->pdim == 2) ? 3 : 6;
    double *U, *sn;
    int step = 0, i;
    if (p->pdim != 2) return -1;
    memset(&s, 0, sizeof(s));
    s.p = p; s.c = c; s.N = N; s.dt = p->dt;
    s.nnz = dm_csr_pattern(p, NULL, NULL);
    s.rowptr = (int *)malloc(sizeof(int) * ((size_t)N + 1));
    s.colind = (int *)malloc(sizeof(int) * (size_t)s.nnz);
    s.K = (double *)malloc(sizeof(double) * (size_t)s.nnz);
    s.J = (double *)malloc(sizeof(double) * (size_t)s.nnz);
    s.F = (double *)malloc(sizeof(double) * (size_t)N);
    s.agg = aggregate_u;
    dm_csr_pattern(p, s.rowptr, s.colind);
    dm_assemble(p, s.rowptr, s.colind, s.K, 1);                          /* main.F90:157-168 (solid elements) */
    dm_apply_kbc(p, s.rowptr, s.colind, s.K);                            /* :169 */
    U = (double *)calloc((size_t)N, sizeof(double));                     /* Vec_U: zero at creation, then carried over */
    sn = (double *)malloc(sizeof(double) * (size_t)p->nnds * cp);
    memset(aggregate_u, 0, sizeof(double) * (size_t)N);
    if (aggregate_stress) memset(aggregate_stress, 0, sizeof(double) * (size_t)p->nnds * cp);
    for (;;) {
        const double t_init = p->dt * (double)step;
        int it = 0, reason, lin = 0;
        double fn = 0.0;
        if (t_init >= p->t || step >= maxsteps) break;
        s.step = step;
        if (c->variant & 4) memset(U, 0, sizeof(double) * (size_t)N);
        dm_form_rhs(p, t_init, p->dt, s.F);                              /* FormRHS :236 */
        bc_force(&s, s.F);                                               /* (the legacy scaling, when asked for) */
        if (!(c->variant & 16) || step == 0) calc_jacobian(&s, U);       /* :241 */
        reason = snes_solve(&s, U, inner, ksp_rtol, ksp_abstol, &it, &fn, &lin);     /* SNESSolve :242 */
        if (its) its[step] = it;
        if (reasons) reasons[step] = reason;
        if (fnorms) fnorms[step] = fn;
        if (U_hist) memcpy(U_hist + (size_t)step * N, U, sizeof(double) * (size_t)N);
        for (i = 0; i < N; i++) aggregate_u[i] += U[i];                  /* :250 */
        if (aggregate_stress) {
            dm_recover_nodal_stress(p, U, sn, NULL);
            for (i = 0; i < p->nnds * cp; i++) aggregate_stress[i] += sn[i];
        }
        step++;
        if (reason < 0) break;
    }
    free(U); free(sn); free(s.rowptr); free(s.colind); free(s.K); free(s.J); free(s.F);
    return step;
}

/* One evaluation of the two callbacks on given state (tests of the HIP callbacks): J (CSR values on the pattern of
 * dm_csr_pattern) and R for the displacement increment du, with aggregate_u and Vec_F supplied. */
void dm_snes_callbacks(const dm_problem *p, const dm_cohesive *c, const int *rowptr, const int *colind, const double *Kbc,
                       const double *F, const double *aggregate_u, const double *du, double *J, double *R)
{
    snes_ctx s;
    memset(&s, 0, sizeof(s));
    s.p = p; s.c = c; s.N = p->nnds * p->pdim; s.dt = p->dt;
    s.nnz = rowptr[s.N];
    s.rowptr = (int *)rowptr; s.colind = (int *)colind;
    s.K = (double *)Kbc; s.J = J; s.F = (double *)F; s.agg = (double *)aggregate_u;
    calc_jacobian(&s, du);
    if (R) calc_residual(&s, du, R);
}
