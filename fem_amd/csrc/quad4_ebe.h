// fem_amd/csrc/quad4_ebe.h -- K_e x_e of one quad4 element (plane strain) without K_e, in the monomial basis of the bilinear
// element: v = c0 + cx xi + cy eta + cxy xi eta.  At the Gauss point (s1 g, s2 g) of the 2x2 rule the reference gradient of
// a field is (cx + s2 cxy', cy + s1 cxy') with cxy' = g cxy; the forces are accumulated as the three moments of
// T = (w/det J) D eps(adj(J) G) adj(J) against the gradients of the monomials and go back to nodal values at the end
// (hex8_ebe.h is the same construction in three dimensions).  Reference arithmetic: FormElKE (src/m_local.F90:21-67)
// applied to a vector, DMat2d (:204-216), node order of ShapeFuncQua (src/m_elems.F90:332-343).
#pragma once
#include <hip/hip_runtime.h>

namespace fem {

struct Mono2 { double x, y, xy; };

__device__ __forceinline__ Mono2 quad_mono(double v0, double v1, double v2, double v3, double g4)
{
    const double a = v1 - v0, b = v2 - v3;   // xi differences on the eta- and the eta+ edge
    Mono2 c;
    c.x = 0.25 * (a + b);
    c.y = 0.25 * ((v3 - v0) + (v2 - v1));
    c.xy = g4 * (b - a);
    return c;
}

// nodal forces fe and x_e . f_e from nodal coordinates X and values u (Gauss points +-g, weight w; lam, mu of the plane-strain D)
__device__ __forceinline__ void quad4_forces(const double (&X)[4][2], const double (&u)[4][2], double g, double w, double lam,
                                             double mu, double (&fe)[4][2], double& dot)
{
    const double g4 = 0.25 * g;
    Mono2 cX[2], cU[2], F[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        cX[j] = quad_mono(X[0][j], X[1][j], X[2][j], X[3][j], g4);
        cU[j] = quad_mono(u[0][j], u[1][j], u[2][j], u[3][j], g4);
        F[j] = Mono2{0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const double s1 = (q & 1) ? 1.0 : -1.0, s2 = (q & 2) ? 1.0 : -1.0;
        double J[2][2], G[2][2];   // J[i][j] = d x_j / d xi_i, G[i][l] = d u_l / d xi_i
#pragma unroll
        for (int j = 0; j < 2; j++) {
            J[0][j] = fma(s2, cX[j].xy, cX[j].x); J[1][j] = fma(s1, cX[j].xy, cX[j].y);
            G[0][j] = fma(s2, cU[j].xy, cU[j].x); G[1][j] = fma(s1, cU[j].xy, cU[j].y);
        }
        // adj[i][j] = det * d xi_j / d x_i
        const double A00 = J[1][1], A01 = -J[0][1], A10 = -J[1][0], A11 = J[0][0];
        const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        double rd = __builtin_amdgcn_rcp(det);
        rd = fma(fma(-det, rd, 1.0), rd, rd);
        rd = fma(fma(-det, rd, 1.0), rd, rd);
        double H[2][2];            // H[l][i] = det * d u_l / d x_i
#pragma unroll
        for (int l = 0; l < 2; l++) {
            H[l][0] = A00 * G[0][l] + A01 * G[1][l];
            H[l][1] = A10 * G[0][l] + A11 * G[1][l];
        }
        const double wr = w * rd, lw = lam * wr, mw = mu * wr, m2 = mw + mw;
        const double lt = lw * (H[0][0] + H[1][1]);
        const double S00 = fma(m2, H[0][0], lt), S11 = fma(m2, H[1][1], lt), S01 = mw * (H[0][1] + H[1][0]);
        const double T[2][2] = {{S00 * A00 + S01 * A10, S00 * A01 + S01 * A11}, {S01 * A00 + S11 * A10, S01 * A01 + S11 * A11}};
#pragma unroll
        for (int l = 0; l < 2; l++) {
            F[l].x += T[l][0];
            F[l].y += T[l][1];
            F[l].xy = fma(s1, T[l][1], fma(s2, T[l][0], F[l].xy));
        }
    }
    double d = 0.0;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        d += cU[l].x * F[l].x + cU[l].y * F[l].y + cU[l].xy * F[l].xy;   // orthogonal butterflies: x_e . f_e from the coefficients
        const double P = 0.25 * F[l].x, Q = 0.25 * F[l].y, PQ = g4 * F[l].xy;
        fe[0][l] = (PQ - P) - Q;    // (xi-, eta-)
        fe[1][l] = (P - PQ) - Q;    // (xi+, eta-)
        fe[2][l] = (P + PQ) + Q;    // (xi+, eta+)
        fe[3][l] = Q - (P + PQ);    // (xi-, eta+)
    }
    dot = d;
}

}  // namespace fem
