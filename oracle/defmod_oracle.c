/* oracle/defmod_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11, fp64) of the reference's hot path; see
 * defmod_oracle.h for scope, conventions and how it is pinned.  Each function
 * names the reference file:line (relative to /root/reference/src) it follows.
 * Written from the reference's behaviour, not copied from it: the reference is
 * Fortran 90 + PETSc, this is C with explicit loops.
 */
#include "defmod_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 8   /* max nodes per element (hex8) */
#define MAXD 24  /* max element dofs */

/* ------------------------------------------------------------------ */
/* element type tables                                                  */
/* ------------------------------------------------------------------ */

int dm_node_count(int t)
{ /* m_elems.F90:27, with the hex count corrected 6 -> 8 */
    static const int n[6] = {0, 3, 4, 4, 8, 4};
    return (t >= 1 && t <= 5) ? n[t] : 0;
}

int dm_nip(int t)
{ /* getNip, m_elems.F90:92-104 */
    static const int n[6] = {0, 1, 4, 1, 8, 2};
    return (t >= 1 && t <= 5) ? n[t] : 0;
}

int dm_dim(int t)
{ /* getDim, m_elems.F90:49-59 */
    static const int n[6] = {0, 2, 2, 3, 3, 2};
    return (t >= 1 && t <= 5) ? n[t] : 0;
}

int dm_nps(int t)
{ /* elNodePerSide, m_elems.F90:28 */
    static const int n[6] = {0, 2, 2, 3, 4, 2};
    return (t >= 1 && t <= 5) ? n[t] : 0;
}

int dm_vtkid(int t)
{ /* getVtkid, m_elems.F90:107-120 */
    static const int n[6] = {0, 5, 9, 10, 12, 9};
    return (t >= 1 && t <= 5) ? n[t] : 0;
}

/* SamPtsTri :212, SamPtsQua :298-316, SamPtsTet :383, SamPtsHex :487-509,
 * SamPtsCoh :602.  ip(nip,pdim) column-major. */
void dm_sam_pts(int t, double *ip, double *w)
{
    const double g = sqrt(1.0 / 3.0);
    int nip = dm_nip(t), i;
    switch (t) {
    case DM_TRI:
        ip[0] = 1.0 / 3.0; ip[1] = 1.0 / 3.0; w[0] = 0.5;
        break;
    case DM_QUA: { /* order (-,-),(-,+),(+,-),(+,+): NOT node order */
        static const int sx[4] = {-1, -1, 1, 1}, sy[4] = {-1, 1, -1, 1};
        for (i = 0; i < 4; i++) { ip[i] = sx[i] * g; ip[i + nip] = sy[i] * g; w[i] = 1.0; }
        break;
    }
    case DM_TET:
        ip[0] = 0.25; ip[1] = 0.25; ip[2] = 0.25; w[0] = 1.0 / 6.0;
        break;
    case DM_HEX: { /* node order: bottom CCW then top CCW */
        static const int sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
        static const int sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
        static const int sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
        for (i = 0; i < 8; i++) {
            ip[i] = sx[i] * g; ip[i + nip] = sy[i] * g; ip[i + 2 * nip] = sz[i] * g; w[i] = 1.0;
        }
        break;
    }
    case DM_COH: { /* CN literal, m_elems.F90:609 */
        const double cn = 0.5773502691896260;
        ip[0] = -cn; ip[1] = cn; ip[2] = 0.0; ip[3] = 0.0; w[0] = 1.0; w[1] = 1.0;
        break;
    }
    default: break;
    }
}

/* Bodies of ShapeFuncPrecompTri :245-268, Qua :332-354, Tet :430-456,
 * Hex :538-566, Coh :628-649 evaluated at one point. */
void dm_shape_func(int t, const double *xi, double *N)
{
    double e = xi[0], n = xi[1], s;
    switch (t) {
    case DM_TRI: N[0] = 1.0 - e - n; N[1] = e; N[2] = n; break;
    case DM_QUA:
        N[0] = 0.25 * (1.0 - e) * (1.0 - n); N[1] = 0.25 * (1.0 + e) * (1.0 - n);
        N[2] = 0.25 * (1.0 + e) * (1.0 + n); N[3] = 0.25 * (1.0 - e) * (1.0 + n);
        break;
    case DM_TET: s = xi[2]; N[0] = 1.0 - e - n - s; N[1] = e; N[2] = n; N[3] = s; break;
    case DM_HEX:
        s = xi[2];
        N[0] = 0.125 * (1.0 - e) * (1.0 - n) * (1.0 - s); N[1] = 0.125 * (1.0 + e) * (1.0 - n) * (1.0 - s);
        N[2] = 0.125 * (1.0 + e) * (1.0 + n) * (1.0 - s); N[3] = 0.125 * (1.0 - e) * (1.0 + n) * (1.0 - s);
        N[4] = 0.125 * (1.0 - e) * (1.0 - n) * (1.0 + s); N[5] = 0.125 * (1.0 + e) * (1.0 - n) * (1.0 + s);
        N[6] = 0.125 * (1.0 + e) * (1.0 + n) * (1.0 + s); N[7] = 0.125 * (1.0 - e) * (1.0 + n) * (1.0 + s);
        break;
    case DM_COH: /* default-real 0.5 literals are exact in binary */
        N[0] = 0.5 * (1 - e); N[1] = 0.5 * (1 + e); N[2] = N[1]; N[3] = N[0];
        break;
    default: break;
    }
}

/* ShapeFuncdTri :273-279, Qua :357-363, Tet :459-466, Hex :569-580.
 * dN(pdim,n) column-major: dN[i + a*pdim]. */
void dm_shape_funcd(int t, const double *xi, double *dN)
{
    double e, n, s;
    int a;
    switch (t) {
    case DM_TRI: {
        static const double d[6] = {-1, -1, 1, 0, 0, 1};
        memcpy(dN, d, sizeof d);
        break;
    }
    case DM_QUA: {
        e = xi[0]; n = xi[1];
        double de[4] = {-(1.0 - n), (1.0 - n), (1.0 + n), -(1.0 + n)};
        double dn[4] = {-(1.0 - e), -(1.0 + e), (1.0 + e), (1.0 - e)};
        for (a = 0; a < 4; a++) { dN[0 + 2 * a] = 0.25 * de[a]; dN[1 + 2 * a] = 0.25 * dn[a]; }
        break;
    }
    case DM_TET: {
        static const double d[12] = {-1, -1, -1, 1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(dN, d, sizeof d);
        break;
    }
    case DM_HEX: {
        e = xi[0]; n = xi[1]; s = xi[2];
        double de[8] = {-(1 - n) * (1 - s), (1 - n) * (1 - s), (1 + n) * (1 - s), -(1 + n) * (1 - s),
                        -(1 - n) * (1 + s), (1 - n) * (1 + s), (1 + n) * (1 + s), -(1 + n) * (1 + s)};
        double dn[8] = {-(1 - e) * (1 - s), -(1 + e) * (1 - s), (1 + e) * (1 - s), (1 - e) * (1 - s),
                        -(1 - e) * (1 + s), -(1 + e) * (1 + s), (1 + e) * (1 + s), (1 - e) * (1 + s)};
        double ds[8] = {-(1 - e) * (1 - n), -(1 + e) * (1 - n), -(1 + e) * (1 + n), -(1 - e) * (1 + n),
                        (1 - e) * (1 - n), (1 + e) * (1 - n), (1 + e) * (1 + n), (1 - e) * (1 + n)};
        for (a = 0; a < 8; a++) {
            dN[0 + 3 * a] = 0.125 * de[a]; dN[1 + 3 * a] = 0.125 * dn[a]; dN[2 + 3 * a] = 0.125 * ds[a];
        }
        break;
    }
    default: break;
    }
}

/* MatDet, m_utils.F90:60-66 (column-major A(n,n)) */
double dm_mat_det(int n, const double *A)
{
#define A_(i, j) A[((i)-1) + ((j)-1) * n]
    if (n == 2) return A_(1, 1) * A_(2, 2) - A_(1, 2) * A_(2, 1);
    if (n == 3)
        return A_(1, 1) * (A_(2, 2) * A_(3, 3) - A_(2, 3) * A_(3, 2)) -
               A_(1, 2) * (A_(2, 1) * A_(3, 3) - A_(2, 3) * A_(3, 1)) +
               A_(1, 3) * (A_(2, 1) * A_(3, 2) - A_(2, 2) * A_(3, 1));
#undef A_
    return 0.0;
}

/* MatInv, m_utils.F90:45-57: the reference calls LAPACK dgetrf+dgetri (LU with
 * partial pivoting).  Restated as LU with partial (row) pivoting followed by
 * column-by-column solves of A X = I; agrees with LAPACK to rounding. */
int dm_mat_inv(int n, const double *A, double *Ainv)
{
    double lu[MAXN * MAXN], col[MAXN];
    int piv[MAXN], i, j, k;
    if (n > MAXN) return -1;
    memcpy(lu, A, sizeof(double) * n * n);
#define LU(i, j) lu[(i) + (j) * n]
    for (k = 0; k < n; k++) {
        int p = k;
        double mx = fabs(LU(k, k));
        for (i = k + 1; i < n; i++)
            if (fabs(LU(i, k)) > mx) { mx = fabs(LU(i, k)); p = i; }
        piv[k] = p;
        if (mx == 0.0) return k + 1;
        if (p != k)
            for (j = 0; j < n; j++) { double tmp = LU(k, j); LU(k, j) = LU(p, j); LU(p, j) = tmp; }
        for (i = k + 1; i < n; i++) {
            LU(i, k) /= LU(k, k);
            for (j = k + 1; j < n; j++) LU(i, j) -= LU(i, k) * LU(k, j);
        }
    }
    for (j = 0; j < n; j++) {
        for (i = 0; i < n; i++) col[i] = (i == j) ? 1.0 : 0.0;
        for (k = 0; k < n; k++)
            if (piv[k] != k) { double tmp = col[k]; col[k] = col[piv[k]]; col[piv[k]] = tmp; }
        for (i = 0; i < n; i++)
            for (k = 0; k < i; k++) col[i] -= LU(i, k) * col[k];
        for (i = n - 1; i >= 0; i--) {
            for (k = i + 1; k < n; k++) col[i] -= LU(i, k) * col[k];
            col[i] /= LU(i, i);
        }
        for (i = 0; i < n; i++) Ainv[i + j * n] = col[i];
    }
#undef LU
    return 0;
}

/* NodalStressInv, m_elems.F90:725-745.  Only for types with nip==nodecount and
 * getDim==pdim.  N2(gp,node) = N_node(xi_gp); returns inv(N2) as (nip,n)
 * column-major.  Returns 1 if filled, 0 if the reference skips the type. */
int dm_nodal_stress_inv(int t, int pdim, double *N2inv)
{
    int n = dm_node_count(t), nip = dm_nip(t), g, a, d;
    double ip[MAXN * 3], w[MAXN], N2[MAXN * MAXN], N[MAXN], xi[3];
    if (dm_dim(t) != pdim || nip != n) return 0;
    dm_sam_pts(t, ip, w);
    for (g = 0; g < nip; g++) {
        for (d = 0; d < pdim; d++) xi[d] = ip[g + d * nip];
        dm_shape_func(t, xi, N);
        for (a = 0; a < n; a++) N2[g + a * n] = N[a];
    }
    return dm_mat_inv(n, N2, N2inv) == 0 ? 1 : 0;
}

/* TriArea, m_utils.F90:25-33 */
static double tri_area(double x1, double y1, double z1, double x2, double y2, double z2,
                       double x3, double y3, double z3)
{
    double m1[9] = {x1, x2, x3, y1, y2, y3, 1, 1, 1};
    double m2[9] = {y1, y2, y3, z1, z2, z3, 1, 1, 1};
    double m3[9] = {z1, z2, z3, x1, x2, x3, 1, 1, 1};
    double d1 = dm_mat_det(3, m1), d2 = dm_mat_det(3, m2), d3 = dm_mat_det(3, m3);
    return 0.5 * sqrt(fabs(d1) * fabs(d1) + fabs(d2) * fabs(d2) + fabs(d3) * fabs(d3));
}

/* EdgeAreaNodesTri :282-293, Qua :366-378, Tet :469-482, Hex :583-599.
 * ecoords(n,pdim) column-major; side 1-based; snodes gets enodes(j). */
void dm_edge_area_nodes(int t, const int *enodes, const double *ecoords, int side, double *area,
                        int *snodes)
{
    int n = dm_node_count(t), nps = dm_nps(t), j[4] = {0, 0, 0, 0}, i;
#define EC(a, d) ecoords[(a) + (d) * n]
    switch (t) {
    case DM_TRI:
        if (side == 1) { j[0] = 0; j[1] = 1; } else if (side == 2) { j[0] = 1; j[1] = 2; }
        else if (side == 3) { j[0] = 2; j[1] = 0; }
        break;
    case DM_QUA:
        if (side == 1) { j[0] = 0; j[1] = 1; } else if (side == 2) { j[0] = 1; j[1] = 2; }
        else if (side == 3) { j[0] = 2; j[1] = 3; } else if (side == 4) { j[0] = 3; j[1] = 0; }
        break;
    case DM_TET:
        if (side == 1) { j[0] = 0; j[1] = 1; j[2] = 3; } else if (side == 2) { j[0] = 1; j[1] = 2; j[2] = 3; }
        else if (side == 3) { j[0] = 0; j[1] = 2; j[2] = 3; } else if (side == 4) { j[0] = 0; j[1] = 1; j[2] = 2; }
        break;
    case DM_HEX: {
        static const int f[6][4] = {{0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}, {0, 1, 2, 3}, {4, 5, 6, 7}};
        if (side >= 1 && side <= 6) for (i = 0; i < 4; i++) j[i] = f[side - 1][i];
        break;
    }
    default: break;
    }
    for (i = 0; i < nps; i++) snodes[i] = enodes[j[i]];
    if (t == DM_TRI || t == DM_QUA) {
        double dx = EC(j[0], 0) - EC(j[1], 0), dy = EC(j[0], 1) - EC(j[1], 1);
        *area = sqrt(dx * dx + dy * dy);
    } else if (t == DM_TET) {
        *area = tri_area(EC(j[0], 0), EC(j[0], 1), EC(j[0], 2), EC(j[1], 0), EC(j[1], 1), EC(j[1], 2),
                         EC(j[2], 0), EC(j[2], 1), EC(j[2], 2));
    } else if (t == DM_HEX) { /* QuadArea, m_utils.F90:36-42 */
        double ta1 = tri_area(EC(j[0], 0), EC(j[0], 1), EC(j[0], 2), EC(j[1], 0), EC(j[1], 1), EC(j[1], 2),
                              EC(j[2], 0), EC(j[2], 1), EC(j[2], 2));
        double ta2 = tri_area(EC(j[0], 0), EC(j[0], 1), EC(j[0], 2), EC(j[2], 0), EC(j[2], 1), EC(j[2], 2),
                              EC(j[3], 0), EC(j[3], 1), EC(j[3], 2));
        *area = fabs(ta1) + fabs(ta2);
    }
#undef EC
}

/* ------------------------------------------------------------------ */
/* element integration (m_local.F90)                                    */
/* ------------------------------------------------------------------ */

/* FormdNdetJ, m_local.F90:175-189: dN/dxi -> J = dN*X -> det (signed) ->
 * dN <- inv(J)*dN.  ecoords(n,pdim), dN(pdim,n) column-major. */
void dm_form_dn_detj(int t, const double *xi, const double *ecoords, double *dN, double *detj)
{
    int n = dm_node_count(t), pd = dm_dim(t), i, j, a;
    double jac[9], inv[9], tmp[3 * MAXN];
    dm_shape_funcd(t, xi, dN);
    for (i = 0; i < pd; i++)
        for (j = 0; j < pd; j++) {
            double s = 0.0;
            for (a = 0; a < n; a++) s += dN[i + a * pd] * ecoords[a + j * n];
            jac[i + j * pd] = s;
        }
    *detj = dm_mat_det(pd, jac);
    dm_mat_inv(pd, jac, inv);
    for (a = 0; a < n; a++)
        for (i = 0; i < pd; i++) {
            double s = 0.0;
            for (j = 0; j < pd; j++) s += inv[i + j * pd] * dN[j + a * pd];
            tmp[i + a * pd] = s;
        }
    memcpy(dN, tmp, sizeof(double) * pd * n);
}

/* BMat, m_local.F90:148-171.  B(cpdim, n*pdim) column-major. */
void dm_bmat(int pdim, int n, const double *dN, double *B)
{
    int cp = (pdim == 2) ? 3 : 6, j;
    memset(B, 0, sizeof(double) * cp * n * pdim);
#define B_(r, c) B[((r)-1) + ((c)-1) * cp]
#define dN_(i, a) dN[((i)-1) + ((a)-1) * pdim]
    if (pdim == 2) {
        for (j = 1; j <= n; j++) {
            B_(1, 2 * j - 1) = dN_(1, j);
            B_(2, 2 * j) = dN_(2, j);
            B_(3, 2 * j - 1) = dN_(2, j); B_(3, 2 * j) = dN_(1, j);
        }
    } else {
        for (j = 1; j <= n; j++) {
            B_(1, 3 * j - 2) = dN_(1, j);
            B_(2, 3 * j - 1) = dN_(2, j);
            B_(3, 3 * j) = dN_(3, j);
            B_(4, 3 * j - 2) = dN_(2, j); B_(4, 3 * j - 1) = dN_(1, j);
            B_(5, 3 * j - 1) = dN_(3, j); B_(5, 3 * j) = dN_(2, j);
            B_(6, 3 * j - 2) = dN_(3, j); B_(6, 3 * j) = dN_(1, j);
        }
    }
#undef B_
#undef dN_
}

/* DMat/DMat2d/DMat3d, m_local.F90:204-228: plane strain / 3-D isotropic. */
void dm_dmat(int pdim, double E, double nu, double *D)
{
    int cp = (pdim == 2) ? 3 : 6, i;
    double c = E / ((1.0 + nu) * (1.0 - 2.0 * nu));
    memset(D, 0, sizeof(double) * cp * cp);
    if (pdim == 2) {
        D[0] = c * (1.0 - nu); D[1] = c * nu; D[3] = c * nu; D[4] = c * (1.0 - nu);
        D[8] = c * ((1.0 - 2.0 * nu) / 2.0);
    } else {
        int j;
        for (i = 0; i < 3; i++)
            for (j = 0; j < 3; j++) D[i + j * 6] = c * ((i == j) ? (1.0 - nu) : nu);
        for (i = 3; i < 6; i++) D[i + i * 6] = c * ((1.0 - 2.0 * nu) / 2.0);
    }
}

/* FormElKE, m_local.F90:21-67: k = sum_gp B^T (D B) w detj.  k(nd,nd) col-major. */
void dm_form_el_ke(int t, const double *ecoords, double E, double nu, double *k)
{
    int n = dm_node_count(t), pd = dm_dim(t), nip = dm_nip(t), cp = (pd == 2) ? 3 : 6;
    int nd = n * pd, g, i, j, r, d;
    double ip[MAXN * 3], w[MAXN], D[36], dN[3 * MAXN], B[6 * MAXD], DB[6 * MAXD], xi[3], detj;
    dm_sam_pts(t, ip, w);
    memset(k, 0, sizeof(double) * nd * nd);
    dm_dmat(pd, E, nu, D);
    for (g = 0; g < nip; g++) {
        for (d = 0; d < pd; d++) xi[d] = ip[g + d * nip];
        dm_form_dn_detj(t, xi, ecoords, dN, &detj);
        dm_bmat(pd, n, dN, B);
        for (j = 0; j < nd; j++)
            for (r = 0; r < cp; r++) {
                double s = 0.0;
                for (i = 0; i < cp; i++) s += D[r + i * cp] * B[i + j * cp];
                DB[r + j * cp] = s;
            }
        for (j = 0; j < nd; j++)
            for (i = 0; i < nd; i++) {
                double s = 0.0;
                for (r = 0; r < cp; r++) s += B[r + i * cp] * DB[r + j * cp];
                k[i + j * nd] += s * w[g] * detj;
            }
    }
}

/* FormElIndx, m_local.F90:70-78, shifted to 0-based node and dof ids. */
void dm_form_el_indx(int pdim, int n, const int *enodes, int *indx)
{
    int j, d;
    for (j = 0; j < n; j++)
        for (d = 0; d < pdim; d++) indx[pdim * j + d] = pdim * enodes[j] + d;
}

/* CalcElStress, m_local.F90:93-123: estress(nip,cpdim) col-major. */
void dm_calc_el_stress(int t, const double *ecoords, const double *edisp, double E, double nu,
                       double *estress)
{
    int n = dm_node_count(t), pd = dm_dim(t), nip = dm_nip(t), cp = (pd == 2) ? 3 : 6;
    int nd = n * pd, g, i, j, d;
    double ip[MAXN * 3], w[MAXN], D[36], dN[3 * MAXN], B[6 * MAXD], eps[6], xi[3], detj;
    dm_sam_pts(t, ip, w);
    dm_dmat(pd, E, nu, D);
    for (g = 0; g < nip; g++) {
        for (d = 0; d < pd; d++) xi[d] = ip[g + d * nip];
        dm_form_dn_detj(t, xi, ecoords, dN, &detj);
        dm_bmat(pd, n, dN, B);
        for (i = 0; i < cp; i++) {
            double s = 0.0;
            for (j = 0; j < nd; j++) s += B[i + j * cp] * edisp[j];
            eps[i] = s;
        }
        for (i = 0; i < cp; i++) {
            double s = 0.0;
            for (j = 0; j < cp; j++) s += D[i + j * cp] * eps[j];
            estress[g + i * nip] = s;
        }
    }
}

/* ------------------------------------------------------------------ */
/* global level                                                         */
/* ------------------------------------------------------------------ */

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

static void gather_ecoords(const dm_problem *p, int e, double *ec)
{ /* ecoords = coords(nodes,:), m_io.F90:102 */
    int n = p->eptr[e + 1] - p->eptr[e], a, d;
    const int *nd = p->conn + p->eptr[e];
    for (a = 0; a < n; a++)
        for (d = 0; d < p->pdim; d++) ec[a + d * n] = p->coords[(size_t)nd[a] * p->pdim + d];
}

/* The sparsity PETSc's MPIAIJ ends up with after all MatSetValues calls of
 * main.F90:157-166: entry (i,j) exists iff dofs i and j share an element. */
long long dm_csr_pattern(const dm_problem *p, int *rowptr, int *colind)
{
    int nn = p->nnds, pd = p->pdim, e, a, b, v;
    int *cnt = (int *)calloc((size_t)nn + 1, sizeof(int));
    int *n2e, *mark, *list;
    long long nnz = 0;
    size_t tot;
    for (e = 0; e < p->nels; e++)
        for (a = p->eptr[e]; a < p->eptr[e + 1]; a++) cnt[p->conn[a] + 1]++;
    for (v = 0; v < nn; v++) cnt[v + 1] += cnt[v];
    tot = (size_t)cnt[nn];
    n2e = (int *)malloc((tot ? tot : 1) * sizeof(int));
    {
        int *fill = (int *)malloc(((size_t)nn + 1) * sizeof(int));
        memcpy(fill, cnt, ((size_t)nn + 1) * sizeof(int));
        for (e = 0; e < p->nels; e++)
            for (a = p->eptr[e]; a < p->eptr[e + 1]; a++) n2e[fill[p->conn[a]]++] = e;
        free(fill);
    }
    mark = (int *)malloc((size_t)(nn ? nn : 1) * sizeof(int));
    for (v = 0; v < nn; v++) mark[v] = -1;
    list = (int *)malloc(4096 * sizeof(int));
    int cap = 4096;
    if (rowptr) rowptr[0] = 0;
    for (v = 0; v < nn; v++) {
        int m = 0, d, c;
        for (a = cnt[v]; a < cnt[v + 1]; a++) {
            e = n2e[a];
            for (b = p->eptr[e]; b < p->eptr[e + 1]; b++) {
                int u = p->conn[b];
                if (mark[u] != v) {
                    mark[u] = v;
                    if (m == cap) { cap *= 2; list = (int *)realloc(list, (size_t)cap * sizeof(int)); }
                    list[m++] = u;
                }
            }
        }
        qsort(list, (size_t)m, sizeof(int), cmp_int);
        for (d = 0; d < pd; d++) {
            if (colind)
                for (a = 0; a < m; a++)
                    for (c = 0; c < pd; c++) colind[nnz + (long long)a * pd + c] = pd * list[a] + c;
            nnz += (long long)m * pd;
            if (rowptr) rowptr[pd * v + d + 1] = (int)nnz;
        }
    }
    free(cnt); free(n2e); free(mark); free(list);
    return nnz;
}

static inline int find_col(const int *colind, int lo, int hi, int c)
{
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (colind[mid] < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* main.F90:157-166 + FormLocalK m_global.F90:239-261.  MatSetValues is handed
 * the column-major k as if it were row-major (main.F90:162), i.e. it inserts
 * the transpose; restated literally (K[idx_i][idx_j] += k(j,i)). */
void dm_assemble(const dm_problem *p, const int *rowptr, const int *colind, double *val, int nthreads)
{
    long long nnz = rowptr[(size_t)p->nnds * p->pdim];
    int e;
    memset(val, 0, sizeof(double) * (size_t)nnz);
    if (nthreads <= 1) {
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
                    int pos = find_col(colind, rowptr[idx[i]], rowptr[idx[i] + 1], idx[j]);
                    val[pos] += k[j + i * nd];
                }
        }
        return;
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
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
                int pos = find_col(colind, rowptr[idx[i]], rowptr[idx[i] + 1], idx[j]);
#ifdef _OPENMP
#pragma omp atomic
#endif
                val[pos] += k[j + i * nd];
            }
    }
}

/* ApplyKBC -> UpdateKMatBcs, m_global.F90:264-299: for every BC record and
 * component with flag == BC_PRESENT (0): MatSetValue(j,j,PENALTY,INSERT). */
void dm_apply_kbc(const dm_problem *p, const int *rowptr, const int *colind, double *val)
{
    int i, d, pd = p->pdim;
    for (i = 0; i < p->nbcs; i++) {
        int node = p->bcnode[(size_t)i * (1 + pd)];
        for (d = 0; d < pd; d++)
            if (p->bcnode[(size_t)i * (1 + pd) + 1 + d] == 0) {
                int j = pd * node + d;
                int pos = find_col(colind, rowptr[j], rowptr[j + 1], j);
                val[pos] = DM_PENALTY_PARAM;
            }
    }
}

/* FormRHS, m_global.F90:373-436 (nceqs = 0) + EnforceBCForce :439-455 +
 * ApplyNodalForce_All :302-322 / _BC :324-339 + ApplyTraction :343-370. */
void dm_form_rhs(const dm_problem *p, double t_init, double dt, double *F)
{
    int pd = p->pdim, i, d, k;
    double t_end = t_init + dt;
    memset(F, 0, sizeof(double) * (size_t)p->nnds * pd);
    for (i = 0; i < p->nfrcs; i++) {
        const double *fv = p->fval + (size_t)i * (pd + 2);
        double t1 = fv[pd], t2 = fv[pd + 1], applied;
        if (t_end < t1 || t_init > t2) continue;
        applied = fmin(t2, t_end) - fmax(t1, t_init);
        for (d = 0; d < pd; d++) F[(size_t)pd * p->fnode[i] + d] += fv[d] * applied / (t2 - t1);
    }
    for (i = 0; i < p->ntrcs; i++) {
        const double *tv = p->tval + (size_t)i * (pd + 2);
        /* the traction window is divided by dt, the force window is not
         * (m_global.F90:414-415 vs :402-403); restated as is */
        double t1 = tv[pd] / dt, t2 = tv[pd + 1] / dt, applied, area, vv[3], ec[MAXN * 3];
        int el = p->telsd[2 * i], side = p->telsd[2 * i + 1], sn[4], nps, t;
        if (t_end < t1 || t_init > t2) continue;
        if (el < 0) continue;
        applied = fmin(t2, t_end) - fmax(t1, t_init);
        for (d = 0; d < pd; d++) vv[d] = tv[d] * (applied / (t2 - t1));
        t = p->eltype[el];
        nps = dm_nps(t);
        gather_ecoords(p, el, ec);
        dm_edge_area_nodes(t, p->conn + p->eptr[el], ec, side, &area, sn);
        for (d = 0; d < pd; d++) vv[d] = vv[d] * area / (double)nps;
        for (k = 0; k < nps; k++)
            for (d = 0; d < pd; d++) F[(size_t)pd * sn[k] + d] += vv[d];
    }
    for (i = 0; i < p->nbcs; i++) { /* INSERT, after the ADD phase was assembled */
        int node = p->bcnode[(size_t)i * (1 + pd)];
        for (d = 0; d < pd; d++)
            if (p->bcnode[(size_t)i * (1 + pd) + 1 + d] == 0)
                F[(size_t)pd * node + d] = DM_PENALTY_PARAM * p->bcval[(size_t)i * pd + d] * dt / p->t;
    }
}

void dm_spmv(int n, const int *rowptr, const int *colind, const double *val, const double *x,
             double *y, int nthreads)
{
    int i;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (i = 0; i < n; i++) {
        double s = 0.0;
        int k;
        for (k = rowptr[i]; k < rowptr[i + 1]; k++) s += val[k] * x[colind[k]];
        y[i] = s;
    }
}

static double vdot(int n, const double *a, const double *b, int nthreads)
{
    double s = 0.0;
    int i;
    if (nthreads <= 1) {
        for (i = 0; i < n; i++) s += a[i] * b[i];
        return s;
    }
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : s) schedule(static) num_threads(nthreads)
#endif
    for (i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* KSPConvergedDefault (PETSc 3.5 iterativ.c) for the zero-initial-guess case:
 * it 0 fixes ttol = max(rtol*rnorm0, abstol). */
static int converged_default(int it, double rnorm, double rtol, double abstol, double dtol,
                             double *rnorm0, double *ttol)
{
    if (it == 0) { *rnorm0 = rnorm; *ttol = fmax(rtol * rnorm, abstol); }
    if (isnan(rnorm) || isinf(rnorm)) return DM_KSP_DIVERGED_NANORINF;
    if (rnorm <= *ttol) return (rnorm < abstol) ? DM_KSP_CONVERGED_ATOL : DM_KSP_CONVERGED_RTOL;
    if (rnorm >= dtol * (*rnorm0)) return DM_KSP_DIVERGED_DTOL;
    return 0;
}

/* KSPSolve_CG (PETSc 3.5 src/ksp/ksp/impls/cg/cg.c), KSP_NORM_PRECONDITIONED,
 * real symmetric, with PCJACOBI (diag reciprocal, zero diagonal -> 1.0).
 * This is what `-ksp_type cg -pc_type jacobi` selects on the reference when
 * MUMPS is absent (main.F90:354,377).  PETSc is not vendored in the reference;
 * the algorithm is restated from its published source. */
int dm_ksp_cg_jacobi(int n, const int *rowptr, const int *colind, const double *val,
                     const double *b, double *x, double rtol, double abstol, double dtol,
                     int maxit, int *its, double *rnorm, double *hist, int hist_len, int nthreads)
{
    double *R = (double *)malloc(sizeof(double) * n), *Z = (double *)malloc(sizeof(double) * n);
    double *P = (double *)malloc(sizeof(double) * n), *W = (double *)malloc(sizeof(double) * n);
    double *dinv = (double *)malloc(sizeof(double) * n);
    double dp, beta = 0.0, betaold = 0.0, dpi = 0.0, dpiold, a, bb, rnorm0 = 0.0, ttol = 0.0;
    int i, k, reason = 0, nh = 0;
    for (i = 0; i < n; i++) {
        double d = 0.0;
        for (k = rowptr[i]; k < rowptr[i + 1]; k++)
            if (colind[k] == i) { d = val[k]; break; }
        dinv[i] = (d != 0.0) ? 1.0 / d : 0.0;
        if (dinv[i] == 0.0) dinv[i] = 1.0;
    }
    for (i = 0; i < n; i++) { x[i] = 0.0; R[i] = b[i]; Z[i] = dinv[i] * R[i]; }
    dp = sqrt(vdot(n, Z, Z, nthreads));
    if (hist && nh < hist_len) hist[nh++] = dp;
    *its = 0;
    reason = converged_default(0, dp, rtol, abstol, dtol, &rnorm0, &ttol);
    if (!reason) {
        beta = vdot(n, Z, R, nthreads);
        i = 0;
        do {
            int q;
            *its = i + 1;
            if (beta == 0.0) { reason = DM_KSP_CONVERGED_ATOL; break; }
            if (i > 0 && beta * betaold < 0.0) { reason = DM_KSP_DIVERGED_INDEFINITE_PC; break; }
            if (i == 0) { memcpy(P, Z, sizeof(double) * n); }
            else { bb = beta / betaold; for (q = 0; q < n; q++) P[q] = Z[q] + bb * P[q]; }
            dpiold = dpi;
            dm_spmv(n, rowptr, colind, val, P, W, nthreads);
            dpi = vdot(n, P, W, nthreads);
            betaold = beta;
            if (dpi == 0.0 || (i > 0 && dpi * dpiold <= 0.0)) { reason = DM_KSP_DIVERGED_INDEFINITE_MAT; break; }
            a = beta / dpi;
            for (q = 0; q < n; q++) { x[q] += a * P[q]; R[q] -= a * W[q]; Z[q] = dinv[q] * R[q]; }
            dp = sqrt(vdot(n, Z, Z, nthreads));
            if (hist && nh < hist_len) hist[nh++] = dp;
            reason = converged_default(i + 1, dp, rtol, abstol, dtol, &rnorm0, &ttol);
            if (reason) break;
            beta = vdot(n, Z, R, nthreads);
            i++;
        } while (i < maxit);
        if (!reason && i >= maxit) reason = DM_KSP_DIVERGED_ITS;
    }
    *rnorm = dp;
    free(R); free(Z); free(P); free(W); free(dinv);
    return reason;
}

/* Stand-in for KSPPREONLY + PCLU/MUMPS (main.F90:355-358): dense LU with
 * partial pivoting of the same assembled matrix.  O(n^3): small decks only. */
int dm_dense_lu_solve(int n, const int *rowptr, const int *colind, const double *val,
                      const double *b, double *x)
{
    double *A = (double *)calloc((size_t)n * n, sizeof(double));
    int i, j, k;
    if (!A) return -1;
#define A_(i, j) A[(size_t)(i) * n + (j)]
    for (i = 0; i < n; i++)
        for (k = rowptr[i]; k < rowptr[i + 1]; k++) A_(i, colind[k]) += val[k];
    for (i = 0; i < n; i++) x[i] = b[i];
    for (k = 0; k < n; k++) {
        int pv = k;
        double mx = fabs(A_(k, k));
        for (i = k + 1; i < n; i++)
            if (fabs(A_(i, k)) > mx) { mx = fabs(A_(i, k)); pv = i; }
        if (mx == 0.0) { free(A); return k + 1; }
        if (pv != k) {
            for (j = 0; j < n; j++) { double tmp = A_(k, j); A_(k, j) = A_(pv, j); A_(pv, j) = tmp; }
            double tb = x[k]; x[k] = x[pv]; x[pv] = tb;
        }
        for (i = k + 1; i < n; i++) {
            double f = A_(i, k) / A_(k, k);
            if (f != 0.0) {
                for (j = k + 1; j < n; j++) A_(i, j) -= f * A_(k, j);
                x[i] -= f * x[k];
            }
        }
    }
    for (i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (j = i + 1; j < n; j++) s -= A_(i, j) * x[j];
        x[i] = s / A_(i, i);
    }
#undef A_
    free(A);
    return 0;
}

/* main.F90:253-279: RecoverStress (m_global.F90:466-486) for every element,
 * RecoverNodalStress (:488-515), count + sum per node, divide by valence.
 * Single rank: global valence == local valence. */
void dm_recover_nodal_stress(const dm_problem *p, const double *uu, double *stress_node,
                             int *count_node)
{
    int pd = p->pdim, cp = (pd == 2) ? 3 : 6, nn = p->nnds, e, a, c, g;
    double n2inv[6][MAXN * MAXN];
    int has[6] = {0, 0, 0, 0, 0, 0}, t;
    int *cnt = count_node ? count_node : (int *)malloc(sizeof(int) * (size_t)(nn ? nn : 1));
    for (t = DM_TRI; t <= DM_HEX; t++) has[t] = dm_nodal_stress_inv(t, pd, n2inv[t]);
    memset(stress_node, 0, sizeof(double) * (size_t)nn * cp);
    memset(cnt, 0, sizeof(int) * (size_t)nn);
    for (e = 0; e < p->nels; e++) {
        int n = p->eptr[e + 1] - p->eptr[e], nip;
        const int *nd = p->conn + p->eptr[e];
        double ec[MAXN * 3], ed[MAXD], es[MAXN * 6], sn[MAXN * 6], E, nu;
        int idx[MAXD];
        t = p->eltype[e];
        if (t < DM_TRI || t > DM_HEX) continue;
        nip = dm_nip(t);
        E = p->mats[(size_t)p->mat_id[e] * 5 + 0]; nu = p->mats[(size_t)p->mat_id[e] * 5 + 1];
        gather_ecoords(p, e, ec);
        dm_form_el_indx(pd, n, nd, idx);
        for (a = 0; a < n * pd; a++) ed[a] = uu[idx[a]];
        dm_calc_el_stress(t, ec, ed, E, nu, es);
        if (nip == n && has[t]) {
            for (c = 0; c < cp; c++)
                for (a = 0; a < n; a++) {
                    double s = 0.0;
                    for (g = 0; g < nip; g++) s += n2inv[t][a + g * n] * es[g + c * nip];
                    sn[a + c * n] = s;
                }
        } else { /* nip == 1: broadcast the single Gauss-point value */
            for (c = 0; c < cp; c++)
                for (a = 0; a < n; a++) sn[a + c * n] = es[0 + c * nip];
        }
        for (a = 0; a < n; a++) {
            cnt[nd[a]] += 1;
            for (c = 0; c < cp; c++) stress_node[(size_t)nd[a] + (size_t)c * nn] += sn[a + c * n];
        }
    }
    for (a = 0; a < nn; a++)   /* a node that only cohesive elements touch has no solid contribution: its stress stays 0 */
        for (c = 0; c < cp; c++)
            if (cnt[a]) stress_node[(size_t)a + (size_t)c * nn] /= (double)cnt[a];
    if (!count_node) free(cnt);
}

/* Increment loop, main.F90:216-296, linear (ncohmats == 0) branch. */
int dm_run(const dm_problem *p, int ksp_type, double rtol, double abstol, double dtol, int maxit,
           double *aggregate_u, double *aggregate_stress, int *its_out, int its_len, int nthreads)
{
    int pd = p->pdim, cp = (pd == 2) ? 3 : 6, N = p->nnds * pd, i, step = 0;
    long long nnz;
    int *rowptr = (int *)malloc(sizeof(int) * ((size_t)N + 1)), *colind;
    double *val, *F, *U, *sn;
    nnz = dm_csr_pattern(p, rowptr, NULL);
    colind = (int *)malloc(sizeof(int) * (size_t)(nnz ? nnz : 1));
    val = (double *)malloc(sizeof(double) * (size_t)(nnz ? nnz : 1));
    dm_csr_pattern(p, rowptr, colind);
    dm_assemble(p, rowptr, colind, val, nthreads);
    dm_apply_kbc(p, rowptr, colind, val);
    F = (double *)malloc(sizeof(double) * N); U = (double *)malloc(sizeof(double) * N);
    sn = (double *)malloc(sizeof(double) * (size_t)p->nnds * cp);
    memset(aggregate_u, 0, sizeof(double) * N);
    memset(aggregate_stress, 0, sizeof(double) * (size_t)p->nnds * cp);
    for (;;) {
        double t_init = p->dt * (double)step;
        int its = 0, reason;
        double rn = 0.0;
        if (t_init >= p->t) break;
        dm_form_rhs(p, t_init, p->dt, F);
        if (ksp_type == 0) {
            reason = dm_ksp_cg_jacobi(N, rowptr, colind, val, F, U, rtol, abstol, dtol, maxit, &its, &rn,
                                      NULL, 0, nthreads);
            if (reason < 0) { step = reason; break; }
        } else {
            if (dm_dense_lu_solve(N, rowptr, colind, val, F, U) != 0) { step = -1; break; }
            its = 1;
        }
        if (its_out && step < its_len) its_out[step] = its;
        for (i = 0; i < N; i++) aggregate_u[i] += U[i];
        dm_recover_nodal_stress(p, U, sn, NULL);
        for (i = 0; i < p->nnds * cp; i++) aggregate_stress[i] += sn[i];
        step++;
    }
    free(rowptr); free(colind); free(val); free(F); free(U); free(sn);
    return step;
}

/* ---- cohesive element, element level ------------------------------------------------- */
/* SamPtsCoh src/m_elems.F90:602-624 (CN = 0.5773502691896260, weights 1) and ShapeFuncPrecompCoh :626-647 */
static const double COH_CN = 0.5773502691896260;
static void coh_shape(int gp, double *N)
{
    const double eta = (gp == 0) ? -COH_CN : COH_CN;
    N[0] = 0.5 * (1 - eta);
    N[1] = 0.5 * (1 + eta);
    N[2] = N[1];
    N[3] = N[0];
}

void dm_coh_values(const double *ec, double *tangent, double *normal, double *det)
{
    /* ecoords(node, dim): ec[node + 4*dim]; src/m_elems.F90:664-672 */
    tangent[0] = 0.5 * (ec[1] - ec[0] + ec[2] - ec[3]);
    tangent[1] = 0.5 * (ec[4 + 1] - ec[4 + 0] + ec[4 + 2] - ec[4 + 3]);
    const double mag = sqrt(tangent[0] * tangent[0] + tangent[1] * tangent[1]);   /* magnitude, src/m_utils.F90 */
    tangent[0] /= mag;
    tangent[1] /= mag;
    *det = mag * 0.5;
    normal[0] = -tangent[1];
    normal[1] = tangent[0];
}

void dm_coh_rels(const double *u8, int gp, double dt, double *urel, double *vrel)
{
    double N[4];
    coh_shape(gp, N);
    urel[0] = urel[1] = 0.0;
    for (int n1 = 0; n1 < 2; n1++) {          /* node1 = 1..nodecount/2, node2 = node1 + nodecount/2 */
        const int n2 = n1 + 2;
        for (int i = 0; i < 2; i++) urel[i] = urel[i] + N[n2] * u8[2 * n2 + i] - N[n1] * u8[2 * n1 + i];
    }
    vrel[0] = urel[0] / dt;
    vrel[1] = urel[1] / dt;
}

void dm_coh_gaps(const double *t, const double *n, const double *urel, const double *vrel, double *gap, double *vgap)
{
    gap[0] = n[0] * urel[0] + n[1] * urel[1];
    gap[1] = t[0] * urel[0] + t[1] * urel[1];
    vgap[0] = n[0] * vrel[0] + n[1] * vrel[1];
    vgap[1] = t[0] * vrel[0] + t[1] * vrel[1];
}

void dm_seplaw1_tract(const double *props, const double *gap, const double *vgap, double dtime, double *tract)
{
    (void)dtime;
    const double sepwrk = exp(1.0) * props[0] * props[1];
    const double dn = props[1], dt = props[2], q = props[3], r = props[4], zeta = props[5];
    double c1, c2;
    c1 = (1.0 - exp(-gap[1] * gap[1] / (dt * dt)));
    c1 = c1 * (1.0 - q) / (r - 1.0) * (r - gap[0] / dn);
    c2 = (gap[0] / dn) * exp(-gap[1] * gap[1] / (dt * dt));
    tract[0] = (sepwrk / dn) * exp(-gap[0] / dn) * (c2 + c1);
    c1 = q + (r - q) / (r - 1.0) * (gap[0] / dn);
    c1 = c1 * exp(-gap[0] / dn) * exp(-gap[1] * gap[1] / (dt * dt));
    c1 = c1 * 2.0 * (dn / dt) * (sepwrk / dn);
    tract[1] = c1 * gap[1] / dt;
    tract[0] = tract[0] + zeta * props[0] * vgap[0] / dn;
}

void dm_seplaw1_stiff(const double *props, const double *gap, const double *vgap, double dtime, double *stiff)
{
    (void)vgap;
    const double sep_work = exp(1.0) * props[0] * props[1];
    const double dn = props[1], dt = props[2], q = props[3], r = props[4], zeta = props[5];
    double c1, s11, s22, s12;
    c1 = (1.0 - q) / (r - 1) * (1.0 - exp(-gap[1] * gap[1] / (dt * dt)));
    c1 = c1 * (r + 1.0 - gap[0] / dn);
    c1 = (1.0 - gap[0] / dn) * exp(-gap[1] * gap[1] / (dt * dt)) - c1;
    s11 = (sep_work / (dn * dn)) * exp(-gap[0] / dn) * c1;
    c1 = q + (gap[0] / dn) * (r - q) / (r - 1.0);
    c1 = c1 * exp(-gap[0] / dn) * exp(-gap[1] * gap[1] / (dt * dt));
    c1 = 2.0 * (sep_work / (dt * dt)) * c1;
    s22 = c1 * (1.0 - 2.0 * (gap[1] * gap[1] / (dt * dt)));
    c1 = -gap[0] / dn + (1.0 - q) / (r - 1.0) * (r - gap[0] / dn);
    c1 = c1 * exp(-gap[0] / dn) * exp(-gap[1] * gap[1] / (dt * dt));
    c1 = 2.0 * (sep_work / (dt * dn)) * c1;
    s12 = (gap[1] / dt) * c1;
    s11 = s11 + zeta * props[0] / dn / dtime;
    stiff[0] = s11; stiff[1] = s12; stiff[2] = s12; stiff[3] = s22;   /* (1,1) (2,1) (1,2) (2,2) */
}

void dm_coh_apply_tract_1(const double *ec, const double *u8, const double *props, double dt, double *result)
{
    double t[2], n[2], det, urel[2], vrel[2], gap[2], vgap[2], f[2], N[4];
    dm_coh_values(ec, t, n, &det);
    for (int k = 0; k < 8; k++) result[k] = 0.0;
    for (int gp = 0; gp < 2; gp++) {
        dm_coh_rels(u8, gp, dt, urel, vrel);
        dm_coh_gaps(t, n, urel, vrel, gap, vgap);
        dm_seplaw1_tract(props, gap, vgap, dt, f);
        coh_shape(gp, N);
        double sig = 1.0;
        for (int node = 0; node < 4; node++) {
            if (node + 1 > 2) sig = -1.0;
            for (int dof = 0; dof < 2; dof++)   /* "=": every integration point overwrites, src/m_global.F90:668 */
                result[node * 2 + dof] = sig * N[node] * (f[0] * n[dof] + f[1] * t[dof]) * 1.0 * det;
        }
    }
}

void dm_coh_apply_stiff_1(const double *ec, const double *u8, const double *props, double dt, double *result)
{
    double t[2], n[2], det, urel[2], vrel[2], gap[2], vgap[2], st[4], N[4];
    dm_coh_values(ec, t, n, &det);
    for (int k = 0; k < 64; k++) result[k] = 0.0;
#define ST(i, j) st[(i) + 2 * (j)]
    for (int gp = 0; gp < 2; gp++) {
        dm_coh_rels(u8, gp, dt, urel, vrel);
        dm_coh_gaps(t, n, urel, vrel, gap, vgap);
        dm_seplaw1_stiff(props, gap, vgap, dt, st);
        coh_shape(gp, N);
        double sig1 = 1.0;
        for (int n1 = 0; n1 < 4; n1++) {
            if (n1 + 1 > 2) sig1 = -1.0;
            double sig2 = 1.0;
            for (int n2 = 0; n2 < 4; n2++) {
                if (n2 + 1 > 2) sig2 = -1.0;
                for (int d1 = 0; d1 < 2; d1++)
                    for (int d2 = 0; d2 < 2; d2++) {
                        /* src/m_global.F90:829-831: only the second term carries sig1*sig2*N*N*w*det */
                        double *r = &result[(n1 * 2 + d1) + 8 * (n2 * 2 + d2)];
                        *r = *r + ((ST(0, 0) * n[d1] + ST(1, 0) * t[d1]) * n[d2]) +
                             ((ST(0, 1) * n[d1] + ST(1, 1) * t[d1]) * t[d2]) * sig1 * sig2 * N[n1] * N[n2] * 1.0 * det;
                    }
            }
        }
    }
#undef ST
}
