// fem_amd/csrc/elements.h -- element library tables (module `elems` of the
// reference, src/m_elems.F90) as plain data for __constant__ memory, and the
// per-Gauss-point kinematics shared by every element kernel.
#pragma once
#include "common.h"

namespace fem {

// One record per element type (index = type number 1..5, src/m_elems.F90:26).
struct ElemTable {
    int type, npe, nip, dim, nps, vtk, has_inv, pad_;
    double xi[8][3];      // Gauss points      SamPts*            (src/m_elems.F90:212,298,383,487,602)
    double w[8];          // weights
    double N[8][8];       // N[gp][node]       ShapeFuncPrecomp*  (:245,332,430,538,628)
    double dNxi[8][3][8]; // dN/dxi [gp][d][node]  ShapeFuncd*    (:273,357,459,569)
    double n2inv[8][8];   // [node][gp] = inverse of N[gp][node]  NodalStressInv (:725-745)
};

// Host: fill the table for one type (0 on success).
int build_elem_table(int type, ElemTable* t);
// Host: all five tables, built once.
const ElemTable* host_elem_tables();

// elNodeCount (with hex = 8), getNip, getDim
inline int type_npe(int t) { static const int v[6] = {0, 3, 4, 4, 8, 4}; return (t >= 1 && t <= 5) ? v[t] : 0; }
inline int type_nip(int t) { static const int v[6] = {0, 1, 4, 1, 8, 2}; return (t >= 1 && t <= 5) ? v[t] : 0; }
inline int type_dim(int t) { static const int v[6] = {0, 2, 2, 3, 3, 2}; return (t >= 1 && t <= 5) ? v[t] : 0; }

#ifdef __HIPCC__
#define FEM_HD __host__ __device__
#else
#define FEM_HD
#endif
// ShapeFuncd* (src/m_elems.F90:273,357,459,569; coh :651-656) at an arbitrary point: d[i][a] = dN_a/dxi_i
FEM_HD inline void elem_shape_d(int t, const double* x, double (&d)[3][8])
{
    const double e = x[0], n = x[1], s = x[2];
    for (int i = 0; i < 3; i++)
        for (int a = 0; a < 8; a++) d[i][a] = 0.0;
    switch (t) {
    case FEM_TRI:
        d[0][0] = -1; d[0][1] = 1; d[0][2] = 0;
        d[1][0] = -1; d[1][1] = 0; d[1][2] = 1;
        break;
    case FEM_QUA: {
        const double de[4] = {-(1.0 - n), (1.0 - n), (1.0 + n), -(1.0 + n)};
        const double dn[4] = {-(1.0 - e), -(1.0 + e), (1.0 + e), (1.0 - e)};
        for (int a = 0; a < 4; a++) { d[0][a] = 0.25 * de[a]; d[1][a] = 0.25 * dn[a]; }
        break;
    }
    case FEM_TET:
        d[0][0] = -1; d[0][1] = 1;
        d[1][0] = -1; d[1][2] = 1;
        d[2][0] = -1; d[2][3] = 1;
        break;
    case FEM_HEX: {
        const double sg[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1},
                                 {-1, -1, 1},  {1, -1, 1},  {1, 1, 1},  {-1, 1, 1}};
        for (int a = 0; a < 8; a++) {
            const double fe = 1.0 + sg[a][0] * e, fn = 1.0 + sg[a][1] * n, fs = 1.0 + sg[a][2] * s;
            d[0][a] = 0.125 * (sg[a][0] * fn * fs);
            d[1][a] = 0.125 * (sg[a][1] * fe * fs);
            d[2][a] = 0.125 * (sg[a][2] * fe * fn);
        }
        break;
    }
    case FEM_COH: /* ShapeFuncdCoh, src/m_elems.F90:651-656 */
        d[0][0] = -0.5; d[0][1] = 0.5; d[0][2] = 0.5; d[0][3] = -0.5;
        for (int a = 0; a < 4; a++) d[1][a] = d[0][a];
        break;
    default: break;
    }
}

#ifdef __HIPCC__
// FormdNdetJ (src/m_local.F90:175-189) at one Gauss point: J = dNxi * X,
// detj = det J (signed, MatDet src/m_utils.F90:60-66), dN = J^-1 * dNxi.
// The reference inverts J with LAPACK LU; here the closed-form adjugate is
// used (differs at rounding level only).  X[a][d], dN[a][d].
template <int PD, int NPE>
__device__ __forceinline__ void dndx_at_gp(const double (&dNxi)[3][8], const double (&X)[NPE][PD],
                                           double (&dN)[NPE][PD], double& detj)
{
    double J[PD][PD];
#pragma unroll
    for (int i = 0; i < PD; i++)
#pragma unroll
        for (int j = 0; j < PD; j++) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < NPE; a++) s += dNxi[i][a] * X[a][j];
            J[i][j] = s;
        }
    double inv[PD][PD];
    if constexpr (PD == 2) {
        detj = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        const double r = 1.0 / detj;
        inv[0][0] = J[1][1] * r; inv[0][1] = -J[0][1] * r;
        inv[1][0] = -J[1][0] * r; inv[1][1] = J[0][0] * r;
    } else {
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
        const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        detj = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
        const double r = 1.0 / detj;
        inv[0][0] = c00 * r;
        inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
        inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
        inv[1][0] = -c01 * r;
        inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
        inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
        inv[2][0] = c02 * r;
        inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
        inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
    }
#pragma unroll
    for (int a = 0; a < NPE; a++)
#pragma unroll
        for (int i = 0; i < PD; i++) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < PD; j++) s += inv[i][j] * dNxi[j][a];
            dN[a][i] = s;
        }
}

// DMat (src/m_local.F90:204-228): the three distinct entries of the isotropic
// plane-strain / 3-D elasticity matrix.
struct DCoef {
    double d11, d12, d44;
};
__device__ __forceinline__ DCoef dmat_coef(double E, double nu)
{
    const double c = E / ((1.0 + nu) * (1.0 - 2.0 * nu));
    return DCoef{c * (1.0 - nu), c * nu, c * ((1.0 - 2.0 * nu) / 2.0)};
}
#endif

}  // namespace fem
