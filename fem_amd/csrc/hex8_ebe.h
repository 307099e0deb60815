// fem_amd/csrc/hex8_ebe.h -- K_e x_e of one hex8 element without K_e: the device functions shared by the
// element-by-element MatMult kernels (elem_kernels.hip: element buffer + node gather; ebe_tile.hip: tiles with the nodal
// sums in LDS).  Reference arithmetic: FormElKE (src/m_local.F90:21-67) applied to a vector, 2x2x2 Gauss rule.
#pragma once
#include <hip/hip_runtime.h>

#include "elements.h"

namespace fem {

// hex8: the same product in the monomial basis of the trilinear element (sum factorisation written out).
//   v(xi) = c0 + cx xi + cy eta + cz zeta + cxy xi eta + cxz xi zeta + cyz eta zeta + cxyz xi eta zeta
// Eight nodal values become seven coefficients by a three-stage butterfly (c0 never enters a gradient); at the Gauss
// point (s1 g, s2 g, s3 g) of the 2x2x2 rule the reference gradient of a field is three signed sums of coefficients
// that carry their powers of g, shared between points with equal signs, instead of 24 multiply-adds against dN/dxi;
// the element forces are accumulated as the seven moments of T against the gradients of the monomials and go back to
// nodal values by the transposed butterfly.  ~1,550 fp64 instructions per element instead of ~2,750.
//   H' = adj(J) G and T = (w/det) D eps(H') adj(J): the inverse Jacobian itself is never formed.
struct Mono { double x, y, z, xy, xz, yz, xyz; };

__device__ __forceinline__ Mono hex_mono(const double (&v)[8], double g8, double gg8, double& c0)
{
    // stage xi
    const double s00 = v[1] + v[0], d00 = v[1] - v[0];   // (eta-, zeta-)
    const double s10 = v[2] + v[3], d10 = v[2] - v[3];   // (eta+, zeta-)
    const double s01 = v[5] + v[4], d01 = v[5] - v[4];   // (eta-, zeta+)
    const double s11 = v[6] + v[7], d11 = v[6] - v[7];   // (eta+, zeta+)
    // stage eta
    const double ss0 = s10 + s00, sd0 = s10 - s00, ds0 = d10 + d00, dd0 = d10 - d00;
    const double ss1 = s11 + s01, sd1 = s11 - s01, ds1 = d11 + d01, dd1 = d11 - d01;
    // stage zeta, with 1/8 and the powers of g the gradients at the Gauss points need
    Mono c;
    c.x = 0.125 * (ds1 + ds0);
    c.y = 0.125 * (sd1 + sd0);
    c.z = 0.125 * (ss1 - ss0);
    c.xy = g8 * (dd1 + dd0);
    c.xz = g8 * (ds1 - ds0);
    c.yz = g8 * (sd1 - sd0);
    c.xyz = gg8 * (dd1 - dd0);
    c0 = 0.125 * (ss1 + ss0);
    return c;
}

// T[l][j] = (w / det J) * sum_i (D eps(adj(J) G))[l][i] adj(J)[i][j] at one Gauss point: the nodal forces are
// f_a[l] += sum_j T[l][j] dN_a/dxi_j
__device__ __forceinline__ void hex_point_T(const double (&J)[3][3], const double (&G)[3][3], double w, double lam, double mu,
                                            double (&T)[3][3])
{
    double adj[3][3];   // adj[i][j] = det * d xi_j / d x_i
    adj[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    adj[1][0] = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    adj[2][0] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    adj[0][1] = J[0][2] * J[2][1] - J[0][1] * J[2][2];
    adj[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
    adj[2][1] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
    adj[0][2] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
    adj[1][2] = J[0][2] * J[1][0] - J[0][0] * J[1][2];
    adj[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const double det = J[0][0] * adj[0][0] + J[0][1] * adj[1][0] + J[0][2] * adj[2][0];
    double rd = __builtin_amdgcn_rcp(det);      // two Newton steps: full double precision
    rd = fma(fma(-det, rd, 1.0), rd, rd);
    rd = fma(fma(-det, rd, 1.0), rd, rd);
    double H[3][3];                             // H[l][i] = det * d u_l / d x_i
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int i = 0; i < 3; i++) H[l][i] = adj[i][0] * G[0][l] + adj[i][1] * G[1][l] + adj[i][2] * G[2][l];
    // DMat (src/m_local.F90:204) as lambda, mu; one factor 1/det from H, one from adj, w det from the quadrature
    const double wr = w * rd;
    const double lw = lam * wr, mw = mu * wr, m2 = mw + mw;
    const double lt = lw * ((H[0][0] + H[1][1]) + H[2][2]);
    double S[3][3];
    S[0][0] = fma(m2, H[0][0], lt);
    S[1][1] = fma(m2, H[1][1], lt);
    S[2][2] = fma(m2, H[2][2], lt);
    S[0][1] = S[1][0] = mw * (H[0][1] + H[1][0]);
    S[1][2] = S[2][1] = mw * (H[1][2] + H[2][1]);
    S[0][2] = S[2][0] = mw * (H[2][0] + H[0][2]);
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[l][j] = S[l][0] * adj[0][j] + S[l][1] * adj[1][j] + S[l][2] * adj[2][j];
}

// the two Gauss points (s1 g, s2 g, -+g): signs s1, s2 are run-time wave-uniform values of a rolled loop (the live set
// stays at coefficients + moments + one pair's temporaries), the zeta sign is written out so that the pair shares the
// partial sums of its gradients and of its moments
__device__ __forceinline__ void hex_pair(double s1, double s2, const Mono (&cX)[3], const Mono (&cU)[3], double w, double lam,
                                         double mu, Mono (&F)[3])
{
    double Jm[3][3], Jp[3][3], Gm[3][3], Gp[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        {
            const Mono& c = cX[j];
            const double A1 = fma(s1, c.xy, c.y), A2 = fma(s1, c.xyz, c.yz), A3 = fma(s1, c.xz, c.z);
            const double B1 = fma(s2, c.xy, c.x), B2 = fma(s2, c.xyz, c.xz);
            Jm[0][j] = B1 - B2; Jp[0][j] = B1 + B2;
            Jm[1][j] = A1 - A2; Jp[1][j] = A1 + A2;
            Jm[2][j] = Jp[2][j] = fma(s2, A2, A3);
        }
        {
            const Mono& c = cU[j];
            const double A1 = fma(s1, c.xy, c.y), A2 = fma(s1, c.xyz, c.yz), A3 = fma(s1, c.xz, c.z);
            const double B1 = fma(s2, c.xy, c.x), B2 = fma(s2, c.xyz, c.xz);
            Gm[0][j] = B1 - B2; Gp[0][j] = B1 + B2;
            Gm[1][j] = A1 - A2; Gp[1][j] = A1 + A2;
            Gm[2][j] = Gp[2][j] = fma(s2, A2, A3);
        }
    }
    double Tm[3][3], Tp[3][3];
    hex_point_T(Jm, Gm, w, lam, mu, Tm);
    hex_point_T(Jp, Gp, w, lam, mu, Tp);
    const double s12 = s1 * s2;
#pragma unroll
    for (int l = 0; l < 3; l++) {
        const double U0 = Tp[l][0] + Tm[l][0], U1 = Tp[l][1] + Tm[l][1], U2 = Tp[l][2] + Tm[l][2];
        const double V0 = Tp[l][0] - Tm[l][0], V1 = Tp[l][1] - Tm[l][1];
        F[l].x += U0;
        F[l].y += U1;
        F[l].z += U2;
        F[l].xy = fma(s1, U1, fma(s2, U0, F[l].xy));
        F[l].xz = fma(s1, U2, F[l].xz + V0);
        F[l].yz = fma(s2, U2, F[l].yz + V1);
        F[l].xyz = fma(s12, U2, fma(s1, V1, fma(s2, V0, F[l].xyz)));
    }
}

// Parallelepipeds (constant Jacobian -- the "affine cell" class of matrix-free finite-element codes): adj(J), det J and so the
// map L: G -> T are the same at the eight Gauss points, T(q) = sum_p p(q) L(G_p) over the seven sign patterns p of the
// gradient of a trilinear field, and the sum over the points keeps, for every moment, the pattern(s) of that monomial's own
// gradient (the sign patterns are orthogonal over the 2x2x2 rule: sum_q p(q) p'(q) = 8 [p = p']).  Seven sparse
// applications of L instead of eight full ones.
//   RM: rows of G that are not zero, CM: columns of T that are wanted (bit i = row / column i)
template <int RM, int CM>
__device__ __forceinline__ void affine_T(const double (&A)[3][3], const double (&G)[3][3], double lw, double mw, double m2,
                                         double (&T)[3][3])
{
    double H[3][3];
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            double h = 0.0;
            bool first = true;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if ((RM >> j) & 1) { h = first ? A[i][j] * G[j][l] : fma(A[i][j], G[j][l], h); first = false; }
            H[l][i] = h;
        }
    const double lt = lw * ((H[0][0] + H[1][1]) + H[2][2]);
    double S[3][3];
    S[0][0] = fma(m2, H[0][0], lt);
    S[1][1] = fma(m2, H[1][1], lt);
    S[2][2] = fma(m2, H[2][2], lt);
    S[0][1] = S[1][0] = mw * (H[0][1] + H[1][0]);
    S[1][2] = S[2][1] = mw * (H[1][2] + H[2][1]);
    S[0][2] = S[2][0] = mw * (H[2][0] + H[0][2]);
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            if ((CM >> j) & 1) T[l][j] = S[l][0] * A[0][j] + S[l][1] * A[1][j] + S[l][2] * A[2][j];
}

// geometry of a parallelepiped: A = adj(J) (A[i][j] = det * d xi_j / d x_i) and the two Lame factors with the quadrature
// weight of all eight points and 1/det folded in
__device__ __forceinline__ void hex_affine_geom(const double (&J)[3][3], double w, double lam, double mu, double (&A)[3][3],
                                                double& lw, double& mw)
{
    A[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    A[1][0] = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    A[2][0] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    A[0][1] = J[0][2] * J[2][1] - J[0][1] * J[2][2];
    A[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
    A[2][1] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
    A[0][2] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
    A[1][2] = J[0][2] * J[1][0] - J[0][0] * J[1][2];
    A[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const double det = J[0][0] * A[0][0] + J[0][1] * A[1][0] + J[0][2] * A[2][0];
    double rd = __builtin_amdgcn_rcp(det);
    rd = fma(fma(-det, rd, 1.0), rd, rd);
    rd = fma(fma(-det, rd, 1.0), rd, rd);
    const double wr = 8.0 * w * rd;   // eight points of equal weight
    lw = lam * wr; mw = mu * wr;
}
// Jacobian of a parallelepiped from one corner and its three neighbours (nodes 0, 1, 3, 4)
__device__ __forceinline__ void hex_affine_jac(const double (&X0)[3], const double (&X1)[3], const double (&X3)[3],
                                               const double (&X4)[3], double (&J)[3][3])
{
#pragma unroll
    for (int j = 0; j < 3; j++) {
        J[0][j] = 0.5 * (X1[j] - X0[j]);
        J[1][j] = 0.5 * (X3[j] - X0[j]);
        J[2][j] = 0.5 * (X4[j] - X0[j]);
    }
}

__device__ __forceinline__ void hex_affine_apply(const double (&A)[3][3], double lw, double mw, const Mono (&cU)[3],
                                                 Mono (&F)[3])
{
    const double m2 = mw + mw;
    double G[3][3], T[3][3];
    // pattern 1: G = (c.x, c.y, c.z) -> moments x, y, z
#pragma unroll
    for (int l = 0; l < 3; l++) { G[0][l] = cU[l].x; G[1][l] = cU[l].y; G[2][l] = cU[l].z; }
    affine_T<7, 7>(A, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) { F[l].x = T[l][0]; F[l].y = T[l][1]; F[l].z = T[l][2]; }
    // pattern s1 (sign of xi): G = (0, c.xy, c.xz) -> second terms of xy and xz
#pragma unroll
    for (int l = 0; l < 3; l++) { G[1][l] = cU[l].xy; G[2][l] = cU[l].xz; }
    affine_T<6, 6>(A, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) { F[l].xy = T[l][1]; F[l].xz = T[l][2]; }
    // pattern s2: G = (c.xy, 0, c.yz) -> first term of xy, second of yz
#pragma unroll
    for (int l = 0; l < 3; l++) { G[0][l] = cU[l].xy; G[2][l] = cU[l].yz; }
    affine_T<5, 5>(A, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) { F[l].xy += T[l][0]; F[l].yz = T[l][2]; }
    // pattern s3: G = (c.xz, c.yz, 0) -> first terms of xz and yz
#pragma unroll
    for (int l = 0; l < 3; l++) { G[0][l] = cU[l].xz; G[1][l] = cU[l].yz; }
    affine_T<3, 3>(A, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) { F[l].xz += T[l][0]; F[l].yz += T[l][1]; }
    // patterns s2 s3, s1 s3, s1 s2: the xyz coefficient in one row -> the three terms of xyz
#pragma unroll
    for (int l = 0; l < 3; l++) G[0][l] = G[1][l] = G[2][l] = cU[l].xyz;
    affine_T<1, 1>(A, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) F[l].xyz = T[l][0];
    affine_T<2, 2>(A, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) F[l].xyz += T[l][1];
    affine_T<4, 4>(A, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) F[l].xyz += T[l][2];
}

__device__ __forceinline__ void hex_affine(const double (&J)[3][3], const Mono (&cU)[3], double w, double lam, double mu,
                                           Mono (&F)[3])
{
    double A[3][3], lw, mw;
    hex_affine_geom(J, w, lam, mu, A, lw, mw);
    hex_affine_apply(A, lw, mw, cU, F);
}


// moments -> nodal forces (the transposed butterfly) and x_e . f_e
__device__ __forceinline__ void hex_finish(const Mono (&cU)[3], const Mono (&F)[3], double g8, double gg8, double (&fe)[8][3],
                                           double& dot)
{
    double d = 0.0;
#pragma unroll
    for (int l = 0; l < 3; l++) {
        // x_e . f_e in the monomial basis: the butterflies are orthogonal up to the factor 8 the scalings carry
        d += cU[l].x * F[l].x + cU[l].y * F[l].y + cU[l].z * F[l].z + cU[l].xy * F[l].xy + cU[l].xz * F[l].xz +
             cU[l].yz * F[l].yz + cU[l].xyz * F[l].xyz;
        // moments -> nodal forces: f_a = 1/8 sum_m (sign pattern of monomial m at node a) * moment_m
        const double P = 0.125 * F[l].x, Q = 0.125 * F[l].y, R = 0.125 * F[l].z;
        const double PQ = g8 * F[l].xy, PR = g8 * F[l].xz, QR = g8 * F[l].yz, PQR = gg8 * F[l].xyz;
        // zeta, then eta, then xi
        const double a10m = P - PR, a10p = P + PR, a01m = Q - QR, a01p = Q + QR, a11m = PQ - PQR, a11p = PQ + PQR;
        const double b0mm = -R - a01m, b0pm = -R + a01m, b0mp = R - a01p, b0pp = R + a01p;      // [eta][zeta]
        const double b1mm = a10m - a11m, b1pm = a10m + a11m, b1mp = a10p - a11p, b1pp = a10p + a11p;
        fe[0][l] = b0mm - b1mm; fe[1][l] = b0mm + b1mm;   // (xi-+, eta-, zeta-)
        fe[3][l] = b0pm - b1pm; fe[2][l] = b0pm + b1pm;   // eta+, zeta-
        fe[4][l] = b0mp - b1mp; fe[5][l] = b0mp + b1mp;   // eta-, zeta+
        fe[7][l] = b0pp - b1pp; fe[6][l] = b0pp + b1pp;   // eta+, zeta+
    }
    dot = d;
}

// a parallelepiped known as such from the set-up (same test as in hex_forces, once per mesh): the Jacobian is read off
// three edges of one corner
__device__ __forceinline__ void hex_forces_affine(const double (&X)[8][3], const double (&u)[8][3], double g, double w,
                                                  double lam, double mu, double (&fe)[8][3], double& dot)
{
    const double g8 = 0.125 * g, gg8 = g8 * g;
    Mono cU[3], F[3];
    double J[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double v[8], c0;
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = u[a][j];
        cU[j] = hex_mono(v, g8, gg8, c0);
    }
    hex_affine_jac(X[0], X[1], X[3], X[4], J);
    hex_affine(J, cU, w, lam, mu, F);
    hex_finish(cU, F, g8, gg8, fe, dot);
}

// Axis-aligned bricks (the "Cartesian cell" class): adj(J) is diagonal, a[i] = det * d xi_i / d x_i, and the seven sparse
// applications of L lose their zero terms by hand -- 177 instead of 321 fp64 instructions, 40 instead of 88 B of stored
// geometry per element.  RM: rows of G that are not zero, CM: columns of T that are wanted.
template <int RM, int CM>
__device__ __forceinline__ void cart_T(const double (&a)[3], const double (&G)[3][3], double lw, double mw, double m2,
                                       double (&T)[3][3])
{
    constexpr bool r0 = RM & 1, r1 = (RM >> 1) & 1, r2 = (RM >> 2) & 1;
    double H[3][3];   // H[l][i] = a[i] G[i][l] for the rows i of RM (the other entries are zero and never touched)
#pragma unroll
    for (int l = 0; l < 3; l++) {
        if (r0) H[l][0] = a[0] * G[0][l];
        if (r1) H[l][1] = a[1] * G[1][l];
        if (r2) H[l][2] = a[2] * G[2][l];
    }
    double tr = 0.0;
    if (r0 && r1 && r2) tr = (H[0][0] + H[1][1]) + H[2][2];
    else if (r0 && r1) tr = H[0][0] + H[1][1];
    else if (r0 && r2) tr = H[0][0] + H[2][2];
    else if (r1 && r2) tr = H[1][1] + H[2][2];
    else if (r0) tr = H[0][0];
    else if (r1) tr = H[1][1];
    else tr = H[2][2];
    const double lt = lw * tr;
    const bool has[3] = {r0, r1, r2};
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (!((CM >> j) & 1)) continue;
#pragma unroll
        for (int l = 0; l < 3; l++) {
            double s;
            if (l == j) s = has[j] ? fma(m2, H[j][j], lt) : lt;
            else if (has[j] && has[l]) s = mw * (H[l][j] + H[j][l]);
            else if (has[j]) s = mw * H[l][j];
            else if (has[l]) s = mw * H[j][l];
            else s = 0.0;
            T[l][j] = s * a[j];
        }
    }
}
__device__ __forceinline__ void hex_cart_apply(const double (&a)[3], double lw, double mw, const Mono (&cU)[3], Mono (&F)[3])
{
    const double m2 = mw + mw;
    double G[3][3], T[3][3];
#pragma unroll
    for (int l = 0; l < 3; l++) { G[0][l] = cU[l].x; G[1][l] = cU[l].y; G[2][l] = cU[l].z; }
    cart_T<7, 7>(a, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) { F[l].x = T[l][0]; F[l].y = T[l][1]; F[l].z = T[l][2]; }
#pragma unroll
    for (int l = 0; l < 3; l++) { G[1][l] = cU[l].xy; G[2][l] = cU[l].xz; }
    cart_T<6, 6>(a, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) { F[l].xy = T[l][1]; F[l].xz = T[l][2]; }
#pragma unroll
    for (int l = 0; l < 3; l++) { G[0][l] = cU[l].xy; G[2][l] = cU[l].yz; }
    cart_T<5, 5>(a, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) { F[l].xy += T[l][0]; F[l].yz = T[l][2]; }
#pragma unroll
    for (int l = 0; l < 3; l++) { G[0][l] = cU[l].xz; G[1][l] = cU[l].yz; }
    cart_T<3, 3>(a, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) { F[l].xz += T[l][0]; F[l].yz += T[l][1]; }
#pragma unroll
    for (int l = 0; l < 3; l++) G[0][l] = G[1][l] = G[2][l] = cU[l].xyz;
    cart_T<1, 1>(a, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) F[l].xyz = T[l][0];
    cart_T<2, 2>(a, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) F[l].xyz += T[l][1];
    cart_T<4, 4>(a, G, lw, mw, m2, T);
#pragma unroll
    for (int l = 0; l < 3; l++) F[l].xyz += T[l][2];
}
// The same map with its coefficients formed first.  In the basis of the sign patterns (the monomials' coefficients without their
// scalings: c^ = B u, B the three-stage butterfly, B B^T = 8 I) the element matrix of a brick is  f = B^T K^ c^  with
// K^ = D K D, D = diag(1/8 | g/8 | g^2/8) by the order of the monomial, and K has 45 entries that are not zero:
//   first order   F^j.j = P_j c^j.j + Q_ji c^i.i + Q_jk c^k.k          F^l.j = R_j c^l.j + S_jl c^j.l   (l != j)
//   second order  F^l.ij = (diagonal: sums of two of P, R) c^l.ij + one Q or two S terms (below)
//   third order   F^l.xyz = (P_l + R_i + R_k) c^l.xyz
// with  P_j = (2 mw + lw) a_j^2,  R_j = mw a_j^2,  Q_ij = lw a_i a_j,  S_ij = mw a_i a_j  (what cart_T<7,7> ... cart_T<4,4> add up
// to).  30 coefficients: 12 + 15 + 3, formed once per tile by k_tile_first for the kernels whose tiles carry one geometry record
// (k_ebe_tile_x<2, .., UNI>, k_ebe_tile_x4<2>: wave-uniform, in scalar registers).  45 multiply-adds and two butterflies without
// scalings instead of 177 + 42 instructions; the results differ from hex_forces_cart's in the last bits.
constexpr int CART_NK = 30;
__device__ __host__ __forceinline__ void hex_cart_consts(const double (&a)[3], double lw, double mw, double g, double (&K)[CART_NK])
{
#pragma clang fp contract(off)
    const double lw1 = lw * 0.015625, mw1 = mw * 0.015625;   // D^2 of the first order: 1/64
    const double pl = (mw1 + mw1) + lw1;
    const double aa[3] = {a[0] * a[0], a[1] * a[1], a[2] * a[2]};
    const double ab[3] = {a[0] * a[1], a[0] * a[2], a[1] * a[2]};   // pairs 01, 02, 12
    double P[3], R[3], Q[3], S[3];
    for (int j = 0; j < 3; j++) { P[j] = pl * aa[j]; R[j] = mw1 * aa[j]; Q[j] = lw1 * ab[j]; S[j] = mw1 * ab[j]; }
    const double g2 = g * g, g4 = g2 * g2;
    for (int j = 0; j < 3; j++) { K[j] = P[j]; K[3 + j] = R[j]; K[6 + j] = Q[j]; K[9 + j] = S[j]; }
    // second order, diagonal: rows (component, monomial) 0.xy 1.xy 2.xy 0.xz 1.xz 2.xz 0.yz 1.yz 2.yz
    K[12] = (R[1] + P[0]) * g2; K[13] = (P[1] + R[0]) * g2; K[14] = (R[1] + R[0]) * g2;
    K[15] = (R[2] + P[0]) * g2; K[16] = (R[2] + R[0]) * g2; K[17] = (P[2] + R[0]) * g2;
    K[18] = (R[2] + R[1]) * g2; K[19] = (R[2] + P[1]) * g2; K[20] = (P[2] + R[1]) * g2;
    for (int j = 0; j < 3; j++) { K[21 + j] = Q[j] * g2; K[24 + j] = S[j] * g2; }
    K[27] = (P[0] + (R[1] + R[2])) * g4; K[28] = (P[1] + (R[0] + R[2])) * g4; K[29] = (P[2] + (R[0] + R[1])) * g4;
}
// c^ = B v (no scalings, no constant term)
__device__ __forceinline__ Mono hex_signs(const double (&v)[8])
{
    const double s00 = v[1] + v[0], d00 = v[1] - v[0];
    const double s10 = v[2] + v[3], d10 = v[2] - v[3];
    const double s01 = v[5] + v[4], d01 = v[5] - v[4];
    const double s11 = v[6] + v[7], d11 = v[6] - v[7];
    const double ss0 = s10 + s00, sd0 = s10 - s00, ds0 = d10 + d00, dd0 = d10 - d00;
    const double ss1 = s11 + s01, sd1 = s11 - s01, ds1 = d11 + d01, dd1 = d11 - d01;
    Mono c;
    c.x = ds1 + ds0; c.y = sd1 + sd0; c.z = ss1 - ss0;
    c.xy = dd1 + dd0; c.xz = ds1 - ds0; c.yz = sd1 - sd0;
    c.xyz = dd1 - dd0;
    return c;
}
// nodal forces of an axis-aligned brick from the coefficients K^ (hex_cart_consts) and the nodal values u
__device__ __forceinline__ void hex_forces_cart_k(const double (&K)[CART_NK], const double (&u)[8][3], double (&fe)[8][3], double& dot)
{
    Mono c[3], F[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double v[8];
#pragma unroll
        for (int a_ = 0; a_ < 8; a_++) v[a_] = u[a_][j];
        c[j] = hex_signs(v);
    }
    const double *P = K, *R = K + 3, *Q = K + 6, *S = K + 9, *D = K + 12, *Q2 = K + 21, *S2 = K + 24, *X = K + 27;   // Q, S: pairs 01, 02, 12
    F[0].x = fma(Q[1], c[2].z, fma(Q[0], c[1].y, P[0] * c[0].x));
    F[1].y = fma(Q[2], c[2].z, fma(Q[0], c[0].x, P[1] * c[1].y));
    F[2].z = fma(Q[2], c[1].y, fma(Q[1], c[0].x, P[2] * c[2].z));
    F[1].x = fma(S[0], c[0].y, R[0] * c[1].x);
    F[2].x = fma(S[1], c[0].z, R[0] * c[2].x);
    F[0].y = fma(S[0], c[1].x, R[1] * c[0].y);
    F[2].y = fma(S[2], c[1].z, R[1] * c[2].y);
    F[0].z = fma(S[1], c[2].x, R[2] * c[0].z);
    F[1].z = fma(S[2], c[2].y, R[2] * c[1].z);
    F[0].xy = fma(Q2[1], c[2].yz, D[0] * c[0].xy);
    F[1].xy = fma(Q2[2], c[2].xz, D[1] * c[1].xy);
    F[2].xy = fma(S2[1], c[0].yz, fma(S2[2], c[1].xz, D[2] * c[2].xy));
    F[0].xz = fma(Q2[0], c[1].yz, D[3] * c[0].xz);
    F[1].xz = fma(S2[0], c[0].yz, fma(S2[2], c[2].xy, D[4] * c[1].xz));
    F[2].xz = fma(Q2[2], c[1].xy, D[5] * c[2].xz);
    F[0].yz = fma(S2[0], c[1].xz, fma(S2[1], c[2].xy, D[6] * c[0].yz));
    F[1].yz = fma(Q2[0], c[0].xz, D[7] * c[1].yz);
    F[2].yz = fma(Q2[1], c[0].xy, D[8] * c[2].yz);
#pragma unroll
    for (int l = 0; l < 3; l++) F[l].xyz = X[l] * c[l].xyz;
    double d = 0.0;
#pragma unroll
    for (int l = 0; l < 3; l++) {
        // x_e . f_e = c^ . K^ c^
        d += c[l].x * F[l].x + c[l].y * F[l].y + c[l].z * F[l].z + c[l].xy * F[l].xy + c[l].xz * F[l].xz + c[l].yz * F[l].yz +
             c[l].xyz * F[l].xyz;
        // f = B^T F: zeta, then eta, then xi (hex_finish without its scalings)
        const double a10m = F[l].x - F[l].xz, a10p = F[l].x + F[l].xz, a01m = F[l].y - F[l].yz, a01p = F[l].y + F[l].yz;
        const double a11m = F[l].xy - F[l].xyz, a11p = F[l].xy + F[l].xyz;
        const double b0mm = -F[l].z - a01m, b0pm = -F[l].z + a01m, b0mp = F[l].z - a01p, b0pp = F[l].z + a01p;      // [eta][zeta]
        const double b1mm = a10m - a11m, b1pm = a10m + a11m, b1mp = a10p - a11p, b1pp = a10p + a11p;
        fe[0][l] = b0mm - b1mm; fe[1][l] = b0mm + b1mm;
        fe[3][l] = b0pm - b1pm; fe[2][l] = b0pm + b1pm;
        fe[4][l] = b0mp - b1mp; fe[5][l] = b0mp + b1mp;
        fe[7][l] = b0pp - b1pp; fe[6][l] = b0pp + b1pp;
    }
    dot = d;
}
// nodal forces of an axis-aligned brick from its stored geometry (a[3], lw, mw) and the nodal values u: the seven sparse applications
// (kernels that keep a record per lane slot: the 30 coefficients of the other form would not fit their registers)
__device__ __forceinline__ void hex_forces_cart(const double (&a)[3], double lw, double mw, const double (&u)[8][3], double g,
                                                double (&fe)[8][3], double& dot)
{
    const double g8 = 0.125 * g, gg8 = g8 * g;
    Mono cU[3], F[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double v[8], c0;
#pragma unroll
        for (int a_ = 0; a_ < 8; a_++) v[a_] = u[a_][j];
        cU[j] = hex_mono(v, g8, gg8, c0);
    }
    hex_cart_apply(a, lw, mw, cU, F);
    hex_finish(cU, F, g8, gg8, fe, dot);
}

// nodal forces of a parallelepiped from its stored geometry (hex_affine_geom) and the nodal values u
__device__ __forceinline__ void hex_forces_geom(const double (&A)[3][3], double lw, double mw, const double (&u)[8][3], double g,
                                                double (&fe)[8][3], double& dot)
{
    const double g8 = 0.125 * g, gg8 = g8 * g;
    Mono cU[3], F[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double v[8], c0;
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = u[a][j];
        cU[j] = hex_mono(v, g8, gg8, c0);
    }
    hex_affine_apply(A, lw, mw, cU, F);
    hex_finish(cU, F, g8, gg8, fe, dot);
}

// nodal forces fe and x_e . f_e of one hex8 from nodal coordinates X and values u (2x2x2 Gauss rule: points +-g, weight w)
template <bool AFFINE_PATH>
__device__ __forceinline__ void hex_forces(const double (&X)[8][3], const double (&u)[8][3], double g, double w, double lam,
                                           double mu, double (&fe)[8][3], double& dot)
{
    const double g8 = 0.125 * g, gg8 = g8 * g;
    Mono cX[3], cU[3], F[3];
    bool affine = true;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double v[8], c0;
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = X[a][j];
        cX[j] = hex_mono(v, g8, gg8, c0);
        // mixed terms below the rounding noise of the coordinates themselves (2^-50 of their magnitude): a parallelepiped
        const double tol = 0x1p-50 * (fabs(c0) + fabs(cX[j].x) + fabs(cX[j].y) + fabs(cX[j].z));
        affine = affine && fabs(cX[j].xy) <= tol && fabs(cX[j].xz) <= tol && fabs(cX[j].yz) <= tol && fabs(cX[j].xyz) <= tol;
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = u[a][j];
        cU[j] = hex_mono(v, g8, gg8, c0);
    }
    if (AFFINE_PATH && __all(affine)) {
        double J[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) { J[0][j] = cX[j].x; J[1][j] = cX[j].y; J[2][j] = cX[j].z; }
        hex_affine(J, cU, w, lam, mu, F);
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++) F[j] = Mono{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma clang loop unroll(disable)
        for (int q = 0; q < 4; q++) hex_pair((q & 1) ? 1.0 : -1.0, (q & 2) ? 1.0 : -1.0, cX, cU, w, lam, mu, F);
    }
    hex_finish(cU, F, g8, gg8, fe, dot);
}

// ---- nodal stresses of one hex8 element (RecoverNodalStress, src/m_global.F90:489-516) ----------------------------
// Strain and stress at the 2x2x2 Gauss points (CalcElStress, src/m_local.F90:114-120), extrapolated to the nodes with N2^-1: the
// nodal values of the trilinear function that takes the Gauss-point values at the Gauss points.  In the monomial basis:
// the eight moments of the point values against the sign patterns, scaled by the powers of 1/g, and the butterfly back to
// the nodes -- no 8 x 8 table.  HALF 0: sigma_xx, sigma_yy, sigma_zz; HALF 1: sigma_xy, sigma_yz, sigma_zx (the Voigt
// order of BMat, src/m_local.F90:160-168); the tile kernel keeps three sums per node in LDS, so the six components take
// two passes.
template <int HALF>
__device__ __forceinline__ void hex_point_strain(const double (&J)[3][3], const double (&G)[3][3], double (&e)[3])
{
    double adj[3][3];   // adj[i][j] = det * d xi_j / d x_i
    adj[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    adj[1][0] = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    adj[2][0] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    adj[0][1] = J[0][2] * J[2][1] - J[0][1] * J[2][2];
    adj[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
    adj[2][1] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
    adj[0][2] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
    adj[1][2] = J[0][2] * J[1][0] - J[0][0] * J[1][2];
    adj[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const double det = J[0][0] * adj[0][0] + J[0][1] * adj[1][0] + J[0][2] * adj[2][0];
    double rd = __builtin_amdgcn_rcp(det);
    rd = fma(fma(-det, rd, 1.0), rd, rd);
    rd = fma(fma(-det, rd, 1.0), rd, rd);
    auto H = [&](int l, int i) { return adj[i][0] * G[0][l] + adj[i][1] * G[1][l] + adj[i][2] * G[2][l]; };   // det * du_l/dx_i
    if constexpr (HALF == 0) { e[0] = H(0, 0) * rd; e[1] = H(1, 1) * rd; e[2] = H(2, 2) * rd; }
    else { e[0] = (H(0, 1) + H(1, 0)) * rd; e[1] = (H(1, 2) + H(2, 1)) * rd; e[2] = (H(2, 0) + H(0, 2)) * rd; }
}
template <int HALF>
__device__ __forceinline__ void hex_nodal_stress(const double (&X)[8][3], const double (&u)[8][3], double g, double E,
                                                 double nu, double (&sn)[8][3])
{
    const double g8 = 0.125 * g, gg8 = g8 * g;
    Mono cX[3], cU[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double v[8], c0;
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = X[a][j];
        cX[j] = hex_mono(v, g8, gg8, c0);
#pragma unroll
        for (int a = 0; a < 8; a++) v[a] = u[a][j];
        cU[j] = hex_mono(v, g8, gg8, c0);
    }
    Mono M[3];        // moments of the strains at the points against the sign patterns of the monomials (sigma = D eps,
    double M0[3];     // DMat src/m_local.F90:204, is applied to the moments: one matrix for all eight points)
#pragma unroll
    for (int c = 0; c < 3; c++) { M[c] = Mono{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}; M0[c] = 0.0; }
#pragma clang loop unroll(disable)
    for (int q = 0; q < 4; q++) {   // the Gauss points (s1 g, s2 g, -+g), gradients as in hex_pair
        const double s1 = (q & 1) ? 1.0 : -1.0, s2 = (q & 2) ? 1.0 : -1.0, s12 = s1 * s2;
        double Jm[3][3], Jp[3][3], Gm[3][3], Gp[3][3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            {
                const Mono& c = cX[j];
                const double A1 = fma(s1, c.xy, c.y), A2 = fma(s1, c.xyz, c.yz), A3 = fma(s1, c.xz, c.z);
                const double B1 = fma(s2, c.xy, c.x), B2 = fma(s2, c.xyz, c.xz);
                Jm[0][j] = B1 - B2; Jp[0][j] = B1 + B2;
                Jm[1][j] = A1 - A2; Jp[1][j] = A1 + A2;
                Jm[2][j] = Jp[2][j] = fma(s2, A2, A3);
            }
            {
                const Mono& c = cU[j];
                const double A1 = fma(s1, c.xy, c.y), A2 = fma(s1, c.xyz, c.yz), A3 = fma(s1, c.xz, c.z);
                const double B1 = fma(s2, c.xy, c.x), B2 = fma(s2, c.xyz, c.xz);
                Gm[0][j] = B1 - B2; Gp[0][j] = B1 + B2;
                Gm[1][j] = A1 - A2; Gp[1][j] = A1 + A2;
                Gm[2][j] = Gp[2][j] = fma(s2, A2, A3);
            }
        }
        double sm[3], sp[3];
        hex_point_strain<HALF>(Jm, Gm, sm);
        hex_point_strain<HALF>(Jp, Gp, sp);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double U = sp[c] + sm[c], V = sp[c] - sm[c];
            M0[c] += U;
            M[c].x = fma(s1, U, M[c].x);
            M[c].y = fma(s2, U, M[c].y);
            M[c].z += V;
            M[c].xy = fma(s12, U, M[c].xy);
            M[c].xz = fma(s1, V, M[c].xz);
            M[c].yz = fma(s2, V, M[c].yz);
            M[c].xyz = fma(s12, V, M[c].xyz);
        }
    }
    // coefficients of the trilinear fit (the points sit at +-g: one factor 1/g per coordinate), then its nodal values
    const double r8 = 0.125, r1 = 0.125 / g, r2 = r1 / g, r3 = r2 / g;
    const DCoef D = dmat_coef(E, nu);
    auto dmat = [&](double& a, double& b, double& c) {
        if constexpr (HALF == 0) {
            const double exx = a, eyy = b, ezz = c;
            a = D.d11 * exx + D.d12 * eyy + D.d12 * ezz;
            b = D.d12 * exx + D.d11 * eyy + D.d12 * ezz;
            c = D.d12 * exx + D.d12 * eyy + D.d11 * ezz;
        } else { a *= D.d44; b *= D.d44; c *= D.d44; }
    };
    dmat(M0[0], M0[1], M0[2]);
    dmat(M[0].x, M[1].x, M[2].x); dmat(M[0].y, M[1].y, M[2].y); dmat(M[0].z, M[1].z, M[2].z);
    dmat(M[0].xy, M[1].xy, M[2].xy); dmat(M[0].xz, M[1].xz, M[2].xz); dmat(M[0].yz, M[1].yz, M[2].yz);
    dmat(M[0].xyz, M[1].xyz, M[2].xyz);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double c0 = r8 * M0[c], P = r1 * M[c].x, Q = r1 * M[c].y, R = r1 * M[c].z;
        const double PQ = r2 * M[c].xy, PR = r2 * M[c].xz, QR = r2 * M[c].yz, PQR = r3 * M[c].xyz;
        const double a0m = c0 - R, a0p = c0 + R, axm = P - PR, axp = P + PR, aym = Q - QR, ayp = Q + QR;
        const double bm = PQ - PQR, bp = PQ + PQR;                       // zeta -, +
        const double p0m = a0m - aym, q0m = a0m + aym, pxm = axm - bm, qxm = axm + bm;   // eta -, + at zeta -
        const double p0p = a0p - ayp, q0p = a0p + ayp, pxp = axp - bp, qxp = axp + bp;   // eta -, + at zeta +
        sn[0][c] = p0m - pxm; sn[1][c] = p0m + pxm;
        sn[3][c] = q0m - qxm; sn[2][c] = q0m + qxm;
        sn[4][c] = p0p - pxp; sn[5][c] = p0p + pxp;
        sn[7][c] = q0p - qxp; sn[6][c] = q0p + qxp;
    }
}

}  // namespace fem
