/* oracle/cpu_baseline.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The timed CPU leg of bench.py (`cpu_baseline`, kind "port"): the same path as the restatement in
 * defmod_oracle.c (sparsity -> element loop + MatSetValues(ADD) -> ApplyKBC -> FormRHS -> KSPCG/PCJACOBI),
 * organised for a many-core two-socket host instead of for line-by-line readability:
 *
 *   - every large array is malloc'ed here and FIRST TOUCHED by the thread that will stream it (static
 *     schedule over rows, threads bound by OMP_PROC_BIND/OMP_PLACES, Linux first-touch policy), so that each
 *     socket's share of the matrix and of the vectors lives in its own DRAM;
 *   - the sparsity is built in parallel over nodes, the element loop runs in parallel with atomic adds
 *     (what dm_assemble does with nthreads > 1), and every CG vector pass is parallel and fused the way
 *     PETSc's VecAXPY/VecAYPX/VecDot chain cannot be -- i.e. the baseline is FASTER than the reference's own
 *     code would be on the same cores (no per-element allocate, no LAPACK call per Gauss point, no
 *     MatSetValues search/stash; SURVEY.md 8d "CPU baseline timing");
 *   - the algorithm and its numbers are those of the oracle: tests/test_oracle_golden.py checks this file's
 *     matrix and CG iterates against defmod_oracle.c on small meshes.
 *
 * Also timed: the matrix-free (element-by-element) product on the CPU, because the GPU headline applies
 * Mat_K that way.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "defmod_oracle.h"

#define MAXN 8
#define MAXD 24

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

static inline int find_col(const int *colind, long long lo, long long hi, int c)
{
    while (lo < hi) {
        long long mid = lo + (hi - lo) / 2;
        if (colind[mid] < c) lo = mid + 1; else hi = mid;
    }
    return (int)lo;
}

typedef struct {
    long long nnz;
    int *rowptr, *colind;
    double *val;
} csr_t;

/* Sparsity of MPIAIJ after the MatSetValues calls of main.F90:157-166, built in parallel over nodes; the
 * thread that fills (= first touches) a block of rows is the one that later multiplies with them. */
static int build_pattern(const dm_problem *p, int nthreads, csr_t *A)
{
    const int nn = p->nnds, pd = p->pdim, N = nn * pd;
    int *cnt = (int *)calloc((size_t)nn + 2, sizeof(int)), *n2e, *rowlen;
    int e, a, v;
    if (!cnt) return -1;
    for (e = 0; e < p->nels; e++) {
        if (p->eltype[e] < DM_TRI || p->eltype[e] > DM_HEX) continue;
        for (a = p->eptr[e]; a < p->eptr[e + 1]; a++) cnt[p->conn[a] + 1]++;
    }
    for (v = 0; v < nn; v++) cnt[v + 1] += cnt[v];
    n2e = (int *)malloc(((size_t)cnt[nn] + 1) * sizeof(int));
    rowlen = (int *)malloc(((size_t)nn + 1) * sizeof(int));
    {
        int *fill = (int *)malloc(((size_t)nn + 1) * sizeof(int));
        memcpy(fill, cnt, ((size_t)nn + 1) * sizeof(int));
        for (e = 0; e < p->nels; e++) {
            if (p->eltype[e] < DM_TRI || p->eltype[e] > DM_HEX) continue;
            for (a = p->eptr[e]; a < p->eptr[e + 1]; a++) n2e[fill[p->conn[a]]++] = e;
        }
        free(fill);
    }
    A->rowptr = (int *)malloc(((size_t)N + 1) * sizeof(int));
    /* pass 1: neighbour count per node */
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (v = 0; v < nn; v++) {
        int list[512], m = 0, q, b, ee, dup;
        for (q = cnt[v]; q < cnt[v + 1]; q++) {
            ee = n2e[q];
            for (b = p->eptr[ee]; b < p->eptr[ee + 1]; b++) {
                const int u = p->conn[b];
                int z;
                dup = 0;
                for (z = 0; z < m; z++) if (list[z] == u) { dup = 1; break; }
                if (!dup && m < 512) list[m++] = u;
            }
        }
        if (m == 0) { list[0] = v; m = 1; }   /* isolated node keeps its diagonal (as the HIP pattern does) */
        rowlen[v] = m;
    }
    {
        long long k = 0;
        A->rowptr[0] = 0;
        for (v = 0; v < nn; v++) {
            int d;
            for (d = 0; d < pd; d++) {
                k += (long long)rowlen[v] * pd;
                if (k > 2147483647LL) { free(cnt); free(n2e); free(rowlen); return -2; }
                A->rowptr[pd * v + d + 1] = (int)k;
            }
        }
        A->nnz = k;
    }
    A->colind = (int *)malloc((size_t)A->nnz * sizeof(int));
    A->val = (double *)malloc((size_t)A->nnz * sizeof(double));
    if (!A->colind || !A->val) return -1;
    /* pass 2: fill + first touch (colind and val) by the owner of the rows */
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (v = 0; v < nn; v++) {
        int list[512], m = 0, q, b, ee, d, c, z;
        for (q = cnt[v]; q < cnt[v + 1]; q++) {
            ee = n2e[q];
            for (b = p->eptr[ee]; b < p->eptr[ee + 1]; b++) {
                const int u = p->conn[b];
                int dup = 0;
                for (z = 0; z < m; z++) if (list[z] == u) { dup = 1; break; }
                if (!dup && m < 512) list[m++] = u;
            }
        }
        if (m == 0) { list[0] = v; m = 1; }
        qsort(list, (size_t)m, sizeof(int), cmp_int);
        for (d = 0; d < pd; d++) {
            const long long o = A->rowptr[pd * v + d];
            for (z = 0; z < m; z++)
                for (c = 0; c < pd; c++) { A->colind[o + (long long)z * pd + c] = pd * list[z] + c; A->val[o + (long long)z * pd + c] = 0.0; }
        }
    }
    free(cnt); free(n2e); free(rowlen);
    return 0;
}

static void gather_ecoords(const dm_problem *p, int e, double *ec)
{
    int n = p->eptr[e + 1] - p->eptr[e], a, d;
    const int *nd = p->conn + p->eptr[e];
    for (a = 0; a < n; a++)
        for (d = 0; d < p->pdim; d++) ec[a + d * n] = p->coords[(size_t)nd[a] * p->pdim + d];
}

/* main.F90:157-168 (same arithmetic as dm_assemble: dm_form_el_ke + transposed insert + atomics) */
static void assemble(const dm_problem *p, int nthreads, csr_t *A)
{
    int e;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (e = 0; e < p->nels; e++) {
        int t = p->eltype[e], n = p->eptr[e + 1] - p->eptr[e], nd = n * p->pdim, i, j;
        double ec[MAXN * 3], k[MAXD * MAXD], E, nu;
        int idx[MAXD];
        if (t < DM_TRI || t > DM_HEX) continue;
        E = p->mats[(size_t)p->mat_id[e] * 5 + 0]; nu = p->mats[(size_t)p->mat_id[e] * 5 + 1];
        gather_ecoords(p, e, ec);
        dm_form_el_ke(t, ec, E, nu, k);
        dm_form_el_indx(p->pdim, n, p->conn + p->eptr[e], idx);
        for (i = 0; i < nd; i++)
            for (j = 0; j < nd; j++) {
                const int pos = find_col(A->colind, A->rowptr[idx[i]], A->rowptr[idx[i] + 1], idx[j]);
#pragma omp atomic
                A->val[pos] += k[j + i * nd];
            }
    }
}

/* y = K x by element-by-element recomputation (what `-mat_mult_type ebe` does on the GPU): f_e = K_e x_e
 * from dm_form_el_ke-free kinematics -- per Gauss point B, strain, stress, B^T sigma w detJ -- scattered
 * with atomics.  Penalty rows: y_i += (PENALTY - assembled diagonal) x_i is left out (timing only). */
static void ebe_mult(const dm_problem *p, int nthreads, const double *x, double *y)
{
    const int pd = p->pdim, N = p->nnds * pd;
    int i, e;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (i = 0; i < N; i++) y[i] = 0.0;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (e = 0; e < p->nels; e++) {
        const int t = p->eltype[e];
        int n, nip, cp, nd, g, a, d, r;
        double ec[MAXN * 3], ue[MAXD], fe[MAXD], ip[MAXN * 3], w[MAXN], D[36], dN[3 * MAXN], eps[6], sig[6], xi[3], detj;
        if (t < DM_TRI || t > DM_HEX) continue;
        n = p->eptr[e + 1] - p->eptr[e]; nip = dm_nip(t); cp = (pd == 2) ? 3 : 6; nd = n * pd;
        gather_ecoords(p, e, ec);
        for (a = 0; a < n; a++)
            for (d = 0; d < pd; d++) ue[a * pd + d] = x[(size_t)p->conn[p->eptr[e] + a] * pd + d];
        dm_sam_pts(t, ip, w);
        dm_dmat(pd, p->mats[(size_t)p->mat_id[e] * 5 + 0], p->mats[(size_t)p->mat_id[e] * 5 + 1], D);
        for (r = 0; r < nd; r++) fe[r] = 0.0;
        for (g = 0; g < nip; g++) {
            double wd;
            for (d = 0; d < pd; d++) xi[d] = ip[g + d * nip];
            dm_form_dn_detj(t, xi, ec, dN, &detj);
            wd = w[g] * detj;
            for (r = 0; r < cp; r++) eps[r] = 0.0;
            for (a = 0; a < n; a++) {
                const double *dn = dN + a * pd, *u = ue + a * pd;
                if (pd == 2) { eps[0] += dn[0] * u[0]; eps[1] += dn[1] * u[1]; eps[2] += dn[1] * u[0] + dn[0] * u[1]; }
                else {
                    eps[0] += dn[0] * u[0]; eps[1] += dn[1] * u[1]; eps[2] += dn[2] * u[2];
                    eps[3] += dn[1] * u[0] + dn[0] * u[1]; eps[4] += dn[2] * u[1] + dn[1] * u[2];
                    eps[5] += dn[2] * u[0] + dn[0] * u[2];
                }
            }
            for (r = 0; r < cp; r++) {
                double s = 0.0;
                int c;
                for (c = 0; c < cp; c++) s += D[r + c * cp] * eps[c];
                sig[r] = s * wd;
            }
            for (a = 0; a < n; a++) {
                const double *dn = dN + a * pd;
                double *f = fe + a * pd;
                if (pd == 2) { f[0] += dn[0] * sig[0] + dn[1] * sig[2]; f[1] += dn[1] * sig[1] + dn[0] * sig[2]; }
                else {
                    f[0] += dn[0] * sig[0] + dn[1] * sig[3] + dn[2] * sig[5];
                    f[1] += dn[1] * sig[1] + dn[0] * sig[3] + dn[2] * sig[4];
                    f[2] += dn[2] * sig[2] + dn[1] * sig[4] + dn[0] * sig[5];
                }
            }
        }
        for (a = 0; a < n; a++)
            for (d = 0; d < pd; d++) {
                double *dst = y + (size_t)p->conn[p->eptr[e] + a] * pd + d;
#pragma omp atomic
                *dst += fe[a * pd + d];
            }
    }
}

static void spmv(const csr_t *A, int N, const double *x, double *y, int nthreads)
{
    int i;
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (i = 0; i < N; i++) {
        double s = 0.0;
        long long k;
        for (k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) s += A->val[k] * x[A->colind[k]];
        y[i] = s;
    }
}

/* KSPSolve_CG with PCJACOBI (same order of operations as dm_ksp_cg_jacobi), every vector pass parallel.
 * use_ebe: apply Mat_K element by element (timing of the matrix-free CPU product). */
static int cg_jacobi(const dm_problem *p, const csr_t *A, int N, const double *dinv, const double *b, double *x,
                     double *R, double *Z, double *P, double *W, double rtol, int maxit, int nthreads, int use_ebe,
                     int *its_out, double *rnorm_out)
{
    double dp = 0.0, beta = 0.0, betaold = 0.0, dpi = 0.0, dpiold, ttol = 0.0, zz = 0.0, zr = 0.0;
    int i, q, reason = 0;
#pragma omp parallel for schedule(static) reduction(+ : zz, zr) num_threads(nthreads)
    for (q = 0; q < N; q++) { x[q] = 0.0; R[q] = b[q]; Z[q] = dinv[q] * R[q]; P[q] = 0.0; zz += Z[q] * Z[q]; zr += Z[q] * R[q]; }
    dp = sqrt(zz); ttol = fmax(rtol * dp, 1e-50);
    *its_out = 0;
    if (dp <= ttol) { *rnorm_out = dp; return DM_KSP_CONVERGED_RTOL; }
    beta = zr;
    for (i = 0; i < maxit; i++) {
        double a, bb = (i == 0) ? 0.0 : beta / betaold, pw = 0.0;
        *its_out = i + 1;
        if (beta == 0.0) { reason = DM_KSP_CONVERGED_ATOL; break; }
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (q = 0; q < N; q++) P[q] = Z[q] + bb * P[q];
        dpiold = dpi;
        if (use_ebe) ebe_mult(p, nthreads, P, W); else spmv(A, N, P, W, nthreads);
#pragma omp parallel for schedule(static) reduction(+ : pw) num_threads(nthreads)
        for (q = 0; q < N; q++) pw += P[q] * W[q];
        dpi = pw; betaold = beta;
        if (dpi == 0.0 || (i > 0 && dpi * dpiold <= 0.0)) { reason = DM_KSP_DIVERGED_INDEFINITE_MAT; break; }
        a = beta / dpi; zz = 0.0; zr = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : zz, zr) num_threads(nthreads)
        for (q = 0; q < N; q++) {
            double r;
            x[q] += a * P[q];
            r = R[q] - a * W[q]; R[q] = r; Z[q] = dinv[q] * r;
            zz += Z[q] * Z[q]; zr += Z[q] * r;
        }
        dp = sqrt(zz);
        if (isnan(dp) || isinf(dp)) { reason = DM_KSP_DIVERGED_NANORINF; break; }
        if (dp <= ttol) { reason = DM_KSP_CONVERGED_RTOL; break; }
        beta = zr;
    }
    if (!reason) reason = DM_KSP_DIVERGED_ITS;
    *rnorm_out = dp;
    return reason;
}

/* The whole timed leg.  t[0] sparsity, t[1] element loop + ApplyKBC, t[2] Jacobi set-up + FormRHS,
 * t[3] seconds per CG iteration with the assembled SpMV (ncg iterations, or to convergence when it comes
 * first), t[4] seconds per element-by-element product, t[5] seconds per SpMV alone; info[0] nnz,
 * info[1] CG iterations run, info[2] KSP reason.  x_out (may be NULL) receives the CG iterate, val_out/
 * rowptr_out/colind_out (may be NULL, caller-sized) the matrix -- used by the CPU tests only. */
int dm_cpu_baseline(const dm_problem *p, int nthreads, int ncg, double rtol, int nprod, double *t, long long *info,
                    double *x_out, int *rowptr_out, int *colind_out, double *val_out)
{
    const int N = p->nnds * p->pdim;
    csr_t A = {0, NULL, NULL, NULL};
    double *F, *x, *R, *Z, *P, *W, *dinv, t0, t1, rn = 0.0;
    int i, rc, its = 0, reason, k;
    if (nthreads < 1) nthreads = 1;
    t0 = now_s();
    rc = build_pattern(p, nthreads, &A);
    if (rc) return rc;
    t1 = now_s(); t[0] = t1 - t0;
    assemble(p, nthreads, &A);
    dm_apply_kbc(p, A.rowptr, A.colind, A.val);
    t0 = now_s(); t[1] = t0 - t1;
    F = (double *)malloc(sizeof(double) * (size_t)N); x = (double *)malloc(sizeof(double) * (size_t)N);
    R = (double *)malloc(sizeof(double) * (size_t)N); Z = (double *)malloc(sizeof(double) * (size_t)N);
    P = (double *)malloc(sizeof(double) * (size_t)N); W = (double *)malloc(sizeof(double) * (size_t)N);
    dinv = (double *)malloc(sizeof(double) * (size_t)N);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (i = 0; i < N; i++) {   /* first touch + PCSetUp_Jacobi */
        long long q;
        double d = 0.0;
        for (q = A.rowptr[i]; q < A.rowptr[i + 1]; q++) if (A.colind[q] == i) { d = A.val[q]; break; }
        dinv[i] = (d != 0.0) ? 1.0 / d : 1.0;
        F[i] = 0.0; x[i] = 0.0; R[i] = 0.0; Z[i] = 0.0; P[i] = 0.0; W[i] = 0.0;
    }
    dm_form_rhs(p, 0.0, p->dt, F);
    t1 = now_s(); t[2] = t1 - t0;
    reason = cg_jacobi(p, &A, N, dinv, F, x, R, Z, P, W, rtol, ncg, nthreads, 0, &its, &rn);
    t0 = now_s(); t[3] = (t0 - t1) / (double)(its > 0 ? its : 1);
    info[0] = A.nnz; info[1] = its; info[2] = reason;
    if (x_out) memcpy(x_out, x, sizeof(double) * (size_t)N);
    if (rowptr_out) memcpy(rowptr_out, A.rowptr, sizeof(int) * ((size_t)N + 1));
    if (colind_out) memcpy(colind_out, A.colind, sizeof(int) * (size_t)A.nnz);
    if (val_out) memcpy(val_out, A.val, sizeof(double) * (size_t)A.nnz);
    /* products alone, on the last search direction */
    t[4] = t[5] = 0.0;
    if (nprod > 0) {
        ebe_mult(p, nthreads, P, W);
        t1 = now_s();
        for (k = 0; k < nprod; k++) ebe_mult(p, nthreads, P, W);
        t0 = now_s(); t[4] = (t0 - t1) / nprod;
        for (k = 0; k < nprod; k++) spmv(&A, N, P, W, nthreads);
        t1 = now_s(); t[5] = (t1 - t0) / nprod;
        if (val_out == NULL && x_out == NULL) { /* keep W alive for the optimiser */ rn += W[0] * 0.0; }
    }
    free(F); free(x); free(R); free(Z); free(P); free(W); free(dinv);
    free(A.rowptr); free(A.colind); free(A.val);
    return 0;
}

/* y = K x element by element for the tests (checks ebe_mult against the assembled product). */
void dm_cpu_ebe_mult(const dm_problem *p, int nthreads, const double *x, double *y) { ebe_mult(p, nthreads, x, y); }
