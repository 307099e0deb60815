// fem_amd/csrc/elem_kernels.hip -- the element hot loops as HIP kernels:
//   * FormElKE + MatSetValues(ADD) into Mat_K  (src/m_local.F90:21-67, src/main.F90:157-168): row-owner assembly
//   * FormElKE for a caller-supplied batch     (the element-level API)
//   * CalcElStress + RecoverNodalStress          (src/m_local.F90:93-123, src/m_global.F90:466-515)
//   * FormRHS                                    (src/m_global.F90:373-455)
//   * MatMult by element-by-element recomputation
//
// Assembly (k_assemble_rows): one workgroup owns one 64-row slice of the sliced block-ELL matrix and is
// the only writer of its values -- no atomics, every entry (padding included) stored exactly once with
// coalesced stores, two assemblies of the same mesh bitwise equal.  Lane = (block row, incidence): each of
// the <= LPR incident elements of a node per pass gets its own lane, which recomputes that element's
// kinematics at every Gauss point (J, J^-1, dN/dx -- 8x redundant over the element's nodes, ~120 GFLOP
// at config 3, against 13 GB of fp64 atomic traffic in the scatter it replaces) and the NPE blocks
// K_ab = sum_gp w detJ B_a^T D B_b of its row, then adds them to the row's accumulators in LDS through the
// precomputed slot map.  The LDS image of a slice has the layout of the matrix values, so the write-out is
// a straight copy.
// The kernels of the batch API work element-wise: phase 1 thread = (element, Gauss point) parks
// {dN/dx, w*detJ} in LDS, phase 2 thread = (element, node a) forms block row a of the element matrix.
#include <cstring>
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "cg_device.h"
#include "comm.h"
#include "scan.h"
#include "hex8_ebe.h"
#include "objects.h"

using namespace fem;

namespace fem {

__constant__ ElemTable c_tab[6];

void upload_elem_tables(hipStream_t s)
{
    FEM_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_tab), host_elem_tables(), sizeof(ElemTable) * 6, 0,
                                   hipMemcpyHostToDevice, s));
}

// ---- coordinate / displacement sources -----------------------------------------------
template <int PD, int NPE>
struct MeshSrc {  // connectivity + shared coordinate array
    const int* conn;
    const double* coords;
    __device__ __forceinline__ void coordsOf(size_t e, double (&X)[NPE][PD]) const {
#pragma unroll
        for (int a = 0; a < NPE; a++) {
            const size_t v = (size_t)conn[e * NPE + a];
#pragma unroll
            for (int d = 0; d < PD; d++) X[a][d] = coords[v * PD + d];
        }
    }
};
template <int PD, int NPE>
struct DenseSrc {  // caller-supplied ecoords(nodecount,pdim) per element, column-major
    const double* ecoords;
    __device__ __forceinline__ void coordsOf(size_t e, double (&X)[NPE][PD]) const {
#pragma unroll
        for (int a = 0; a < NPE; a++)
#pragma unroll
            for (int d = 0; d < PD; d++) X[a][d] = ecoords[e * NPE * PD + a + d * NPE];
    }
};

// ---- sinks for the element matrix ---------------------------------------------------------
template <int PD, int NPE>
struct DenseSink {  // k(nd,nd) column-major per element (FormElKE's out-argument)
    double* k;
    __device__ __forceinline__ void add(size_t e, int a, int b, const double (&blk)[PD][PD]) const {
        constexpr int ND = NPE * PD;
        double* p = k + e * ND * ND;
#pragma unroll
        for (int i = 0; i < PD; i++)
#pragma unroll
            for (int q = 0; q < PD; q++) p[(a * PD + i) + (size_t)(b * PD + q) * ND] = blk[i][q];
    }
};

// local nodes on the +x face (quad4: 1,2; hex8: 1,2,5,6): the hand-over of the element-by-element product
template <int NPE>
__host__ __device__ constexpr bool plus_face_x(int a)
{
    return NPE == 8 ? (a == 1 || a == 2 || a == 5 || a == 6) : (a == 1 || a == 2);
}

// ---- element stiffness ----------------------------------------------------------------------
template <int PD, int NPE, int NIP, int ELB, class Src, class Sink>
__global__ void __launch_bounds__(ELB*(NIP > NPE ? NIP : NPE))
    k_form_ke(int type, int nel, Src src, const double* __restrict__ Ev, const double* __restrict__ nuv, Sink sink)
{
    constexpr int K = NPE * PD + 1;
    __shared__ double G[NIP * K * ELB];
    const int t = threadIdx.x, el = t % ELB, q = t / ELB;
    const size_t e = (size_t)blockIdx.x * ELB + el;
    const ElemTable& tab = c_tab[type];
    if (q < NIP && e < (size_t)nel) {
        double X[NPE][PD], dN[NPE][PD], detj;
        src.coordsOf(e, X);
        dndx_at_gp<PD, NPE>(tab.dNxi[q], X, dN, detj);
#pragma unroll
        for (int a = 0; a < NPE; a++)
#pragma unroll
            for (int d = 0; d < PD; d++) G[(q * K + a * PD + d) * ELB + el] = dN[a][d];
        G[(q * K + K - 1) * ELB + el] = tab.w[q] * detj;  // weights(i)*detj, src/m_local.F90:64
    }
    __syncthreads();
    if (!(q < NPE && e < (size_t)nel)) return;
    const DCoef D = dmat_coef(Ev[e], nuv[e]);
    double da[NIP][PD];  // w*detJ * dN_a/dx at every Gauss point
#pragma unroll
    for (int g = 0; g < NIP; g++) {
        const double w = G[(g * K + K - 1) * ELB + el];
#pragma unroll
        for (int d = 0; d < PD; d++) da[g][d] = w * G[(g * K + q * PD + d) * ELB + el];
    }
#pragma unroll 1
    for (int b = 0; b < NPE; b++) {
        double blk[PD][PD];
#pragma unroll
        for (int i = 0; i < PD; i++)
#pragma unroll
            for (int j = 0; j < PD; j++) blk[i][j] = 0.0;
#pragma unroll
        for (int g = 0; g < NIP; g++) {
            double db[PD], prod[PD], dot = 0.0;
#pragma unroll
            for (int d = 0; d < PD; d++) {
                db[d] = G[(g * K + b * PD + d) * ELB + el];
                prod[d] = da[g][d] * db[d];
                dot += prod[d];
            }
            // B_a^T D B_b with D = {d11 on the normal diagonal, d12 off-diagonal, d44 shear}
#pragma unroll
            for (int i = 0; i < PD; i++)
#pragma unroll
                for (int j = 0; j < PD; j++)
                    blk[i][j] += (i == j) ? D.d11 * prod[i] + D.d44 * (dot - prod[i])
                                          : D.d12 * da[g][i] * db[j] + D.d44 * da[g][j] * db[i];
        }
        sink.add(e, q, b, blk);
    }
}

// ---- stresses --------------------------------------------------------------------------------
template <int PD, int NPE>
struct MeshU {  // increment displacement gathered through the connectivity (uu(indx), src/m_global.F90:483)
    const int* conn;
    const double* U;
    __device__ __forceinline__ void dispOf(size_t e, double (&u)[NPE][PD]) const {
#pragma unroll
        for (int a = 0; a < NPE; a++) {
            const size_t v = (size_t)conn[e * NPE + a];
#pragma unroll
            for (int d = 0; d < PD; d++) u[a][d] = U[v * PD + d];
        }
    }
};
template <int PD, int NPE>
struct DenseU {
    const double* edisp;
    __device__ __forceinline__ void dispOf(size_t e, double (&u)[NPE][PD]) const {
#pragma unroll
        for (int a = 0; a < NPE; a++)
#pragma unroll
            for (int d = 0; d < PD; d++) u[a][d] = edisp[e * NPE * PD + a * PD + d];
    }
};

// sigma = D * (B u) at one Gauss point (CalcElStress body, src/m_local.F90:114-120)
template <int PD, int NPE>
__device__ __forceinline__ void gp_stress(const double (&dN)[NPE][PD], const double (&u)[NPE][PD], const DCoef& D,
                                          double (&sig)[PD == 2 ? 3 : 6])
{
    if constexpr (PD == 2) {
        double exx = 0, eyy = 0, gxy = 0;
#pragma unroll
        for (int a = 0; a < NPE; a++) {
            exx += dN[a][0] * u[a][0];
            eyy += dN[a][1] * u[a][1];
            gxy += dN[a][1] * u[a][0] + dN[a][0] * u[a][1];
        }
        sig[0] = D.d11 * exx + D.d12 * eyy;
        sig[1] = D.d12 * exx + D.d11 * eyy;
        sig[2] = D.d44 * gxy;
    } else {  // Voigt order xx, yy, zz, xy, yz, zx (BMat, src/m_local.F90:160-168)
        double exx = 0, eyy = 0, ezz = 0, gxy = 0, gyz = 0, gzx = 0;
#pragma unroll
        for (int a = 0; a < NPE; a++) {
            exx += dN[a][0] * u[a][0];
            eyy += dN[a][1] * u[a][1];
            ezz += dN[a][2] * u[a][2];
            gxy += dN[a][1] * u[a][0] + dN[a][0] * u[a][1];
            gyz += dN[a][2] * u[a][1] + dN[a][1] * u[a][2];
            gzx += dN[a][2] * u[a][0] + dN[a][0] * u[a][2];
        }
        sig[0] = D.d11 * exx + D.d12 * eyy + D.d12 * ezz;
        sig[1] = D.d12 * exx + D.d11 * eyy + D.d12 * ezz;
        sig[2] = D.d12 * exx + D.d12 * eyy + D.d11 * ezz;
        sig[3] = D.d44 * gxy;
        sig[4] = D.d44 * gyz;
        sig[5] = D.d44 * gzx;
    }
}

// NODAL = false: write Gauss-point stresses estress(nip,cpdim) per element (CalcElStress).
// NODAL = true : extrapolate to the element's nodes with N2^-1 (RecoverNodalStress) and park them in an
//                element-major buffer out[(a*CP + c)*nel + e] (consecutive elements write consecutive words);
//                k_stress_gather then sums every node's entries in a fixed order -- no atomics.
template <int PD, int NPE, int NIP, int ELB, bool NODAL, class Src, class USrc>
__global__ void __launch_bounds__(ELB*(NIP > NPE ? NIP : NPE))
    k_stress(int type, int nel, Src src, USrc usrc, const double* __restrict__ Ev, const double* __restrict__ nuv,
             double* __restrict__ out)
{
    constexpr int CP = (PD == 2) ? 3 : 6;
    __shared__ double S[NODAL ? NIP * CP * ELB : 1];
    const int t = threadIdx.x, el = t % ELB, q = t / ELB;
    const size_t e = (size_t)blockIdx.x * ELB + el;
    const ElemTable& tab = c_tab[type];
    if (q < NIP && e < (size_t)nel) {
        double X[NPE][PD], u[NPE][PD], dN[NPE][PD], detj, sig[CP];
        src.coordsOf(e, X);
        usrc.dispOf(e, u);
        dndx_at_gp<PD, NPE>(tab.dNxi[q], X, dN, detj);
        gp_stress<PD, NPE>(dN, u, dmat_coef(Ev[e], nuv[e]), sig);
        if constexpr (NODAL) {
#pragma unroll
            for (int c = 0; c < CP; c++) S[(q * CP + c) * ELB + el] = sig[c];
        } else {
#pragma unroll
            for (int c = 0; c < CP; c++) out[e * NIP * CP + q + c * NIP] = sig[c];
        }
    }
    if constexpr (NODAL) {
        __syncthreads();
        if (q < NPE && e < (size_t)nel) {
#pragma unroll
            for (int c = 0; c < CP; c++) {
                double sn = 0.0;
                if constexpr (NIP == NPE) {  // sigma_node = N2^-1 sigma_gp, src/m_global.F90:506-509
#pragma unroll
                    for (int g = 0; g < NIP; g++) sn += tab.n2inv[q][g] * S[(g * CP + c) * ELB + el];
                } else {  // nip == 1: copy, src/m_global.F90:510-513
                    sn = S[c * ELB + el];
                }
                out[(size_t)(q * CP + c) * nel + e] = sn;
            }
        }
    }
}

struct StressViews {
    const double* f[4];
    int nel[4];
};
// thread = node: stress_node(n,:) = sum over the incident elements (ascending (group, element) order) of their
// extrapolated values at n (src/main.F90:258-268), optionally divided by the valence (:277-279) in the same pass
template <int CP>
__global__ void __launch_bounds__(256) k_stress_gather(int nn, const int* __restrict__ ptr, const unsigned* __restrict__ lst,
                                                       StressViews sv, const double* __restrict__ valence,
                                                       double* __restrict__ S)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    double s[CP];
#pragma unroll
    for (int c = 0; c < CP; c++) s[c] = 0.0;
    for (int k = ptr[v]; k < ptr[v + 1]; k++) {
        const unsigned code = lst[k];
        const int g = code >> 30;
        const size_t e = (code & 0x3fffffffu) >> 3, a = code & 7u, ne = (size_t)sv.nel[g];
#pragma unroll
        for (int c = 0; c < CP; c++) s[c] += sv.f[g][(a * CP + c) * ne + e];
    }
    const double rv = valence ? valence[v] : 1.0;
#pragma unroll
    for (int c = 0; c < CP; c++) S[(size_t)c * nn + v] = (valence && rv != 0.0) ? s[c] / rv : s[c];   // (no solid element: 0)
}

__global__ void __launch_bounds__(256) k_divide_by_valence(size_t nn, int cp, double* __restrict__ s,
                                                           const double* __restrict__ valence)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t tot = nn * cp, st = (size_t)gridDim.x * blockDim.x;
    for (; i < tot; i += st) {   // stress_node(i,:)/global_count_node, src/main.F90:277-279
        const double rv = valence[i % nn];
        if (rv != 0.0) s[i] /= rv;
    }
}

// ---- FormRHS -------------------------------------------------------------------------------------
// nodal forces ramped over [t1,t2] (src/m_global.F90:400-411, ApplyNodalForce_All :302-322)
__global__ void __launch_bounds__(256) k_rhs_forces(int nfrcs, int pd, const int* __restrict__ fnode,
                                                    const double* __restrict__ fval, double t_init, double t_end,
                                                    double* __restrict__ F)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfrcs) return;
    const double* fv = fval + (size_t)i * (pd + 2);
    const double t1 = fv[pd], t2 = fv[pd + 1];
    if (t_end < t1 || t_init > t2) return;
    const double applied = fmin(t2, t_end) - fmax(t1, t_init);
    for (int d = 0; d < pd; d++) unsafeAtomicAdd(F + (size_t)pd * fnode[i] + d, fv[d] * applied / (t2 - t1));
}

__device__ inline double det3(double a11, double a21, double a31, double a12, double a22, double a32, double a13,
                              double a23, double a33)
{  // MatDet, src/m_utils.F90:60-66
    return a11 * (a22 * a33 - a23 * a32) - a12 * (a21 * a33 - a23 * a31) + a13 * (a21 * a32 - a22 * a31);
}
__device__ inline double tri_area(const double* p1, const double* p2, const double* p3)
{  // TriArea, src/m_utils.F90:25-33
    const double d1 = det3(p1[0], p2[0], p3[0], p1[1], p2[1], p3[1], 1, 1, 1);
    const double d2 = det3(p1[1], p2[1], p3[1], p1[2], p2[2], p3[2], 1, 1, 1);
    const double d3 = det3(p1[2], p2[2], p3[2], p1[0], p2[0], p3[0], 1, 1, 1);
    return 0.5 * sqrt(fabs(d1) * fabs(d1) + fabs(d2) * fabs(d2) + fabs(d3) * fabs(d3));
}

// tractions: lumped area/nps on the side's nodes (ApplyTraction src/m_global.F90:343-370,
// EdgeAreaNodes* src/m_elems.F90:282,366,469,583).  tnodes[i][8] = the element's nodes.
__global__ void __launch_bounds__(256) k_rhs_tractions(int ntrcs, int pd, const int* __restrict__ ttype,
                                                       const int* __restrict__ tnodes, const int* __restrict__ tside,
                                                       const double* __restrict__ tval,
                                                       const double* __restrict__ coords, double t_init, double t_end,
                                                       double dt, double* __restrict__ F)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntrcs) return;
    const double* tv = tval + (size_t)i * (pd + 2);
    // the traction window is divided by dt (src/m_global.F90:414-415), the force window is not
    const double t1 = tv[pd] / dt, t2 = tv[pd + 1] / dt;
    if (t_end < t1 || t_init > t2) return;
    const double applied = fmin(t2, t_end) - fmax(t1, t_init);
    const int type = ttype[i], side = tside[i];
    const int* en = tnodes + (size_t)i * 8;
    int j[4] = {0, 0, 0, 0}, nps = 2;
    if (type == FEM_TRI) { j[0] = side - 1; j[1] = side % 3; }
    else if (type == FEM_QUA) { j[0] = side - 1; j[1] = side % 4; }
    else if (type == FEM_TET) {
        const int f[4][3] = {{0, 1, 3}, {1, 2, 3}, {0, 2, 3}, {0, 1, 2}};
        nps = 3;
        for (int k = 0; k < 3; k++) j[k] = f[side - 1][k];
    } else {
        const int f[6][4] = {{0, 1, 5, 4}, {1, 2, 6, 5}, {2, 3, 7, 6}, {3, 0, 4, 7}, {0, 1, 2, 3}, {4, 5, 6, 7}};
        nps = 4;
        for (int k = 0; k < 4; k++) j[k] = f[side - 1][k];
    }
    double P[4][3];
    for (int k = 0; k < nps; k++)
        for (int d = 0; d < 3; d++) P[k][d] = (d < pd) ? coords[(size_t)en[j[k]] * pd + d] : 0.0;
    double area;
    if (nps == 2) {
        const double dx = P[0][0] - P[1][0], dy = P[0][1] - P[1][1];
        area = sqrt(dx * dx + dy * dy);
    } else if (nps == 3) {
        area = tri_area(P[0], P[1], P[2]);
    } else {  // QuadArea, src/m_utils.F90:36-42
        area = fabs(tri_area(P[0], P[1], P[2])) + fabs(tri_area(P[0], P[2], P[3]));
    }
    for (int d = 0; d < pd; d++) {
        const double vv = tv[d] * (applied / (t2 - t1)) * area / (double)nps;
        for (int k = 0; k < nps; k++) unsafeAtomicAdd(F + (size_t)pd * en[j[k]] + d, vv);
    }
}

// EnforceBCForce (src/m_global.F90:439-455): INSERT PENALTY*val*dt/t on constrained dofs
__global__ void __launch_bounds__(256) k_rhs_bc(int nbcs, int pd, const int* __restrict__ bcnode,
                                                const double* __restrict__ bcval, double dt, double t_total,
                                                double* __restrict__ F)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbcs) return;
    const int node = bcnode[(size_t)i * (1 + pd)];
    for (int d = 0; d < pd; d++)
        if (bcnode[(size_t)i * (1 + pd) + 1 + d] == 0)
            F[(size_t)pd * node + d] = PENALTY_PARAM * bcval[(size_t)i * pd + d] * dt / t_total;  // left to right, as :451
}

}  // namespace fem

// ---- launch helpers -------------------------------------------------------------------------------

template <int PD, int NPE, int NIP, int ELB, class Src, class Sink>
static void launch_ke(int type, int nel, Src src, const double* E, const double* nu, Sink sink, hipStream_t s)
{
    if (nel <= 0) return;
    constexpr int TPB = ELB * (NIP > NPE ? NIP : NPE);
    hipLaunchKernelGGL((k_form_ke<PD, NPE, NIP, ELB, Src, Sink>), dim3(cdiv(nel, ELB)), dim3(TPB), 0, s, type, nel, src, E,
                       nu, sink);
}

template <class F>
static void dispatch_type(int type, F&& f)
{
    switch (type) {
    case FEM_TRI: f(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{},
                    std::integral_constant<int, 64>{}); break;
    case FEM_QUA: f(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{},
                    std::integral_constant<int, 32>{}); break;
    case FEM_TET: f(std::integral_constant<int, 3>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{},
                    std::integral_constant<int, 64>{}); break;
    case FEM_HEX: f(std::integral_constant<int, 3>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{},
                    std::integral_constant<int, 32>{}); break;
    default: throw Error(3, "unsupported element type " + std::to_string(type));
    }
}

// ---- row-owner assembly (main.F90:157-168 for the whole mesh) ----------------------------------------
namespace fem {

struct AsmGroup {
    const int* conn;
    const double *E, *nu;
    const uint16_t* slot;
    int type, npe;
};
struct AsmGroups {
    AsmGroup g[4];
    int ng;
};

// The NPE blocks K_ab (b = 0..NPE-1) of element e for its local node a:
//   M_ab = sum_gp (w detJ dN_a/dx) (dN_b/dx)^T, then K_ab = d12 M + d44 M^T + (d11 - d12 - d44) diag(M) + d44 tr(M) I
// (the closed form of B_a^T D B_b for the isotropic D of DMat, src/m_local.F90:204-228: the material enters after
// the quadrature sum).  J^-1 is recomputed here for every (element, node) pair; dN/dx is never stored.
template <int PD, int NPE, int NIP, int TYPE>
__device__ __forceinline__ void row_blocks(const AsmGroup& gv, const double* __restrict__ coords, size_t e, int a,
                                           double (&K)[NPE][PD][PD], uint2& s01, uint2& s23)
{
    const ElemTable& tab = c_tab[TYPE];
    // hex8: the shape functions sum to one, so sum_b dN_b/dx = 0 and the last block of the row is minus the sum of the others,
    // M_a7 = -sum_{b<7} M_ab: 63 running sums per lane instead of 72 (the 72 did not fit beside the element's coordinates in the
    // 256 registers two waves per SIMD leave a lane: 17 of them lived in scratch) and an eighth of the inner products less
    constexpr int NB = (NPE == 8) ? NPE - 1 : NPE;
    double X[NPE][PD];
#pragma unroll
    for (int b = 0; b < NPE; b++) {
        const size_t v = (size_t)gv.conn[e * NPE + b];
        if constexpr (PD == 3) {   // one 16 B and one 8 B load per node (8 B aligned)
            struct alignas(8) P3 { double a, b, c; };
            const P3 t = *reinterpret_cast<const P3*>(coords + v * 3);
            X[b][0] = t.a; X[b][1] = t.b; X[b][2] = t.c;
        } else {
            const double2 t = reinterpret_cast<const double2*>(coords)[v];
            X[b][0] = t.x; X[b][1] = t.y;
        }
    }
#pragma unroll
    for (int b = 0; b < NPE; b++)
#pragma unroll
        for (int i = 0; i < PD; i++)
#pragma unroll
            for (int j = 0; j < PD; j++) K[b][i][j] = 0.0;
    // hex8: the coordinates as the seven gradient coefficients of the trilinear map (hex8_ebe.h: the three-stage butterfly, once per
    // lane); the Jacobian at a Gauss point (s1 g, s2 g, s3 g) is then eight multiply-adds per component instead of the 24 against
    // dN/dxi -- 24 instead of 72 per point
    [[maybe_unused]] Mono cX[3];
    if constexpr (NPE == 8 && PD == 3) {
        const double gq = fabs(tab.xi[0][0]), g8 = 0.125 * gq, gg8 = g8 * gq;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double v[8], c0;
#pragma unroll
            for (int b = 0; b < 8; b++) v[b] = X[b][j];
            cX[j] = hex_mono(v, g8, gg8, c0);
        }
    }
    double Em = 0.0, num = 0.0;
#pragma unroll 1
    for (int g = 0; g < NIP; g++) {
        if (g == (NIP > 2 ? NIP - 2 : 0)) {   // what the epilogue needs, a little ahead of it: E, nu and the row's slots
            const uint16_t* sp = gv.slot + (e * NPE + a) * NPE;
            if constexpr (NPE == 8) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sp));   // (read once: streaming load)
                s01 = make_uint2(q.x, q.y); s23 = make_uint2(q.z, q.w);
            } else if constexpr (NPE == 4) {
                s01 = *reinterpret_cast<const uint2*>(sp);
            } else {
                s01 = make_uint2((unsigned)sp[0] | ((unsigned)sp[1] << 16), (unsigned)sp[2]);
            }
            Em = gv.E[e]; num = gv.nu[e];
        }
        double J[PD][PD];
        if constexpr (NPE == 8 && PD == 3) {
            const double s1 = tab.xi[g][0] > 0.0 ? 1.0 : -1.0, s2 = tab.xi[g][1] > 0.0 ? 1.0 : -1.0, s3 = tab.xi[g][2] > 0.0 ? 1.0 : -1.0;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const Mono& c = cX[j];
                const double A1 = fma(s1, c.xy, c.y), A2 = fma(s1, c.xyz, c.yz), A3 = fma(s1, c.xz, c.z);
                const double B1 = fma(s2, c.xy, c.x), B2 = fma(s2, c.xyz, c.xz);
                J[0][j] = fma(s3, B2, B1);
                J[1][j] = fma(s3, A2, A1);
                J[2][j] = fma(s2, A2, A3);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PD; i++)
#pragma unroll
                for (int j = 0; j < PD; j++) {
                    double sj = 0.0;
#pragma unroll
                    for (int b = 0; b < NPE; b++) sj += tab.dNxi[g][i][b] * X[b][j];
                    J[i][j] = sj;
                }
        }
        double inv[PD][PD], detj;        // inv[i][j] = d xi_j / d x_i, as in dndx_at_gp (FormdNdetJ)
        if constexpr (PD == 2) {
            detj = J[0][0] * J[1][1] - J[0][1] * J[1][0];
            const double rd = 1.0 / detj;
            inv[0][0] = J[1][1] * rd; inv[0][1] = -J[0][1] * rd;
            inv[1][0] = -J[1][0] * rd; inv[1][1] = J[0][0] * rd;
        } else {
            const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
            const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
            const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
            detj = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
            double rd = __builtin_amdgcn_rcp(detj);      // two Newton steps: full double precision (as hex_point_T)
            rd = fma(fma(-detj, rd, 1.0), rd, rd);
            rd = fma(fma(-detj, rd, 1.0), rd, rd);
            inv[0][0] = c00 * rd;
            inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * rd;
            inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * rd;
            inv[1][0] = -c01 * rd;
            inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * rd;
            inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * rd;
            inv[2][0] = c02 * rd;
            inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * rd;
            inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rd;
        }
        const double wd = tab.w[g] * detj;   // weights(i)*detj, src/m_local.F90:64
        double da[PD];                       // the row's own node: a differs from lane to lane
#pragma unroll
        for (int i = 0; i < PD; i++) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < PD; j++) t += inv[i][j] * tab.dNxi[g][j][a];
            da[i] = wd * t;
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
            double db[PD];
#pragma unroll
            for (int i = 0; i < PD; i++) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < PD; j++) t += inv[i][j] * tab.dNxi[g][j][b];
                db[i] = t;
            }
#pragma unroll
            for (int i = 0; i < PD; i++)
#pragma unroll
                for (int j = 0; j < PD; j++) K[b][i][j] += da[i] * db[j];
        }
    }
    if constexpr (NB < NPE) {
#pragma unroll
        for (int i = 0; i < PD; i++)
#pragma unroll
            for (int j = 0; j < PD; j++) {
                double t = 0.0;
#pragma unroll
                for (int b = 0; b < NB; b++) t += K[b][i][j];
                K[NPE - 1][i][j] = -t;
            }
    }
    const DCoef D = dmat_coef(Em, num);
#pragma unroll
    for (int b = 0; b < NPE; b++) {
        double M[PD][PD], tr = 0.0;
#pragma unroll
        for (int i = 0; i < PD; i++) {
#pragma unroll
            for (int j = 0; j < PD; j++) M[i][j] = K[b][i][j];
            tr += M[i][i];
        }
#pragma unroll
        for (int i = 0; i < PD; i++)
#pragma unroll
            for (int j = 0; j < PD; j++)
                K[b][i][j] = (i == j) ? D.d11 * M[i][i] + D.d44 * (tr - M[i][i]) : D.d12 * M[i][j] + D.d44 * M[j][i];
    }
}

// One incidence per lane: its element's blocks into the accumulators of its row.  The LDS image holds one row
// after the other, acc[row*RS + j*BB + c] with an odd row stride RS = 1 (mod 32) where the budget allows: the lanes of
// a row (different slots j) and the rows of a half-wave then spread over the banks (two-way conflicts at worst on
// hex8 grids, against eight-way with the rows innermost), and the write-out, which walks the rows, is conflict free.
// Lanes of one wave-instruction hold different rows, or the same row and different local columns b -- which are
// different slots unless the slice is flagged `serial`, where the lanes of a row take turns.
// MODE (meshes whose elements are all the same element -- config 3's bricks -- need the blocks of ONE element, whichever row asks):
//   0  every lane computes its element's blocks (row_blocks)
//   1  the same, and compares them bit for bit with the table of the group's first element (AsmUniform::ke); a difference raises
//      *diff.  This is the first assembly on a mesh: it decides whether the later ones may take mode 2
//   2  the blocks come from the table: the kernel is slot look-ups, LDS adds and the write-out -- and writes the same bits
struct AsmUniform {
    const double* ke;   // [NPE][NPE][PD][PD] blocks K_ab of the group's first element
    int* diff;
};
template <int NPE>
__device__ __forceinline__ void row_slots(const AsmGroup& gv, size_t e, int a, uint2& s01, uint2& s23)
{
    const uint16_t* sp = gv.slot + (e * NPE + a) * NPE;
    if constexpr (NPE == 8) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sp));   // (read once: streaming load)
        s01 = make_uint2(q.x, q.y); s23 = make_uint2(q.z, q.w);
    } else if constexpr (NPE == 4) {
        s01 = *reinterpret_cast<const uint2*>(sp);
    } else {
        s01 = make_uint2((unsigned)sp[0] | ((unsigned)sp[1] << 16), (unsigned)sp[2]);
    }
}
template <int PD, int NPE, int NIP, int TYPE, int LPR, int MODE>
__device__ __forceinline__ void rows_add(bool mine, unsigned code, const AsmGroup& gv, const double* __restrict__ coords,
                                         double* __restrict__ accrow, int k, int j0, int ww, bool serial, const AsmUniform& uni)
{
    constexpr int BB = PD * PD;
    double K[NPE][PD][PD];
    uint2 s01 = make_uint2(0, 0), s23 = make_uint2(0, 0);   // the row's NPE slots, 16 bits each
    if (mine) {
        const size_t e = (size_t)((code & 0x3fffffffu) >> 3);
        const int a = (int)(code & 7u);
        if constexpr (MODE == 2) {
            row_slots<NPE>(gv, e, a, s01, s23);
            const double* t = uni.ke + (size_t)a * NPE * BB;
#pragma unroll
            for (int b = 0; b < NPE; b++)
#pragma unroll
                for (int i = 0; i < PD; i++)
#pragma unroll
                    for (int j = 0; j < PD; j++) K[b][i][j] = t[(b * PD + i) * PD + j];
        } else {
            row_blocks<PD, NPE, NIP, TYPE>(gv, coords, e, a, K, s01, s23);
            if constexpr (MODE == 1) {
                const double* t = uni.ke + (size_t)a * NPE * BB;
                bool same = true;
#pragma unroll
                for (int b = 0; b < NPE; b++)
#pragma unroll
                    for (int i = 0; i < PD; i++)
#pragma unroll
                        for (int j = 0; j < PD; j++)
                            same &= __double_as_longlong(K[b][i][j]) == __double_as_longlong(t[(b * PD + i) * PD + j]);
                if (!same) *uni.diff = 1;
            }
        }
    }
    const unsigned sw[4] = {s01.x, s01.y, s23.x, s23.y};
    auto add_all = [&] {
#pragma unroll
        for (int b = 0; b < NPE; b++) {
            const int slb = (int)((sw[b >> 1] >> ((b & 1) * 16)) & 0xffffu) - j0;
            if ((unsigned)slb < (unsigned)ww) {
                double* q = accrow + slb * BB;
#pragma unroll
                for (int i = 0; i < PD; i++)
#pragma unroll
                    for (int j = 0; j < PD; j++) q[i * PD + j] += K[b][i][j];
            }
        }
    };
    if (!serial) {
        if (mine) add_all();
    } else {
        for (int kk = 0; kk < LPR; kk++)
            if (mine && k == kk) add_all();   // one wave: its LDS operations complete in order
    }
}

// ROWS block rows per workgroup (a 64-row slice is shared by 64/ROWS workgroups), LPR lanes per row.
template <int PD, int LPR, int ROWS, int MODE = 0>
__global__ void __launch_bounds__(ROWS* LPR, 2)   // two waves per SIMD: at most 256 registers per lane
    k_assemble_rows(int nn, int nparts_real, int npass, const int* __restrict__ boff, const int* __restrict__ n2e_ptr,
                    const unsigned* __restrict__ n2e, AsmGroups gs, const double* __restrict__ coords,
                    const unsigned char* __restrict__ slice_serial, int wcap, int rs_pad, int add, double* __restrict__ val,
                    AsmUniform uni)
{
    constexpr int BB = PD * PD, TPB = ROWS * LPR, PARTS = 64 / ROWS;
    extern __shared__ double acc[];   // [ROWS][RS]: the image of this workgroup's rows of the slice
    const int t = threadIdx.x, rl = t / LPR, k = t % LPR;
    // Workgroups b and b + 8 share an XCD (and its L2): hand every XCD a contiguous eighth of the parts, so that
    // neighbouring parts -- which read the same elements' nodes and coordinates -- meet in one L2.  Speed only.
    const int nparts = gridDim.x;
    const int chunk = (nparts + 7) / 8;
    const int part_id = ((int)blockIdx.x % 8) * chunk + (int)blockIdx.x / 8;
    if (part_id >= nparts_real) return;
    const int sl = part_id / PARTS, part = part_id % PARTS;
    const int row = sl * 64 + part * ROWS + rl;
    const int o0 = boff[sl], W = boff[sl + 1] - o0;
    const bool valid = row < nn;
    const int p0 = valid ? n2e_ptr[row] : 0;
    const int cnt = valid ? n2e_ptr[row + 1] - p0 : 0;
    const bool serial = slice_serial[sl] != 0;
    for (int j0 = 0; j0 < W; j0 += wcap) {   // one window unless a slice is wider than the LDS image allows
        const int ww = min(wcap, W - j0), len = ww * BB;
        // row stride: odd, and = 1 (mod 32) when the padding fits (rs_pad)
        const int RS = rs_pad ? ((len + 30) / 32) * 32 + 1 : (len | 1);
        for (int i = t; i < ROWS * RS; i += TPB) acc[i] = 0.0;
        __syncthreads();
        for (int pass = 0; pass < npass; pass++) {
            const int idx = pass * LPR + k;
            const bool act = idx < cnt;
            const unsigned code = act ? __builtin_nontemporal_load(n2e + p0 + idx) : 0u;   // (read once)
            const int gi = (int)(code >> 30);
            for (int g = 0; g < gs.ng; g++) {
                const bool mine = act && gi == g;
                if (!__any(mine)) continue;
                const AsmGroup& gv = gs.g[g];
                double* accrow = acc + rl * RS;
                if constexpr (PD == 3) {
                    if (gv.type == FEM_HEX) rows_add<3, 8, 8, FEM_HEX, LPR, MODE>(mine, code, gv, coords, accrow, k, j0, ww, serial, uni);
                    else rows_add<3, 4, 1, FEM_TET, LPR, MODE>(mine, code, gv, coords, accrow, k, j0, ww, serial, uni);
                } else {
                    if (gv.type == FEM_QUA) rows_add<2, 4, 4, FEM_QUA, LPR, MODE>(mine, code, gv, coords, accrow, k, j0, ww, serial, uni);
                    else rows_add<2, 3, 1, FEM_TRI, LPR, MODE>(mine, code, gv, coords, accrow, k, j0, ww, serial, uni);
                }
            }
        }
        __syncthreads();
        // write-out: entry (j, c) of the ROWS rows is ROWS consecutive words of val; two rows (16 bytes) per lane
        double* dst = val + (size_t)(o0 + j0) * BB * 64 + part * ROWS;
        for (int i = t; i < len * (ROWS / 2); i += TPB) {
            const int jc = i / (ROWS / 2), r = (i % (ROWS / 2)) * 2;
            double2* q = reinterpret_cast<double2*>(dst + (size_t)jc * 64 + r);
            double2 v = make_double2(acc[r * RS + jc], acc[(r + 1) * RS + jc]);
            if (add) { const double2 o = *q; v.x += o.x; v.y += o.y; }   // ADD_VALUES on a matrix that already holds values
            // 6.5 GB at config 3 that nothing reads before the caches have turned over: streaming stores (4.18 -> 3.78 ms)
            typedef double f64x2 __attribute__((ext_vector_type(2)));
            f64x2 t; t.x = v.x; t.y = v.y;
            __builtin_nontemporal_store(t, reinterpret_cast<f64x2*>(q));
        }
        __syncthreads();
    }
}

// the blocks of the group's first element, row by row (one lane per local node), by the code the assembly runs
template <int PD, int NPE, int NIP, int TYPE>
__global__ void __launch_bounds__(64) k_ke_ref(AsmGroup gv, const double* __restrict__ coords, double* __restrict__ ke)
{
    const int a = threadIdx.x;
    if (a >= NPE) return;
    double K[NPE][PD][PD];
    uint2 s01, s23;
    row_blocks<PD, NPE, NIP, TYPE>(gv, coords, 0, a, K, s01, s23);
    for (int b = 0; b < NPE; b++)
        for (int i = 0; i < PD; i++)
            for (int j = 0; j < PD; j++) ke[(((size_t)a * NPE + b) * PD + i) * PD + j] = K[b][i][j];
}

}  // namespace fem

void mat_assemble_elements_impl(fem_mat A)
{
    fem_mesh m = A->mesh;
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const MatPattern& P = *A->pat;
    AsmGroups gs{};
    gs.ng = (int)m->groups.size();
    for (int i = 0; i < gs.ng; i++) {
        auto& g = m->groups[i];
        FEM_CHECK(g.slot.p, 5, "fem_mat_assemble_elements: matrix pattern was not built for this mesh");
        gs.g[i] = AsmGroup{g.conn.p, g.E.p, g.nu.p, g.slot.p, g.type, g.npe};
    }
    const int add = A->lazy_zero ? 0 : 1;
    AsmUniform uni{nullptr, nullptr};
    constexpr int LDS_BUDGET = 76 * 1024;   // two workgroups per CU (160 KiB)
    // FEM_AMD_ASM_WCAP (tests): a smaller accumulator window, so that small decks exercise the multi-window path
    // a slice wider than the LDS image takes
    static const int wcap_env = std::getenv("FEM_AMD_ASM_WCAP") ? std::atoi(std::getenv("FEM_AMD_ASM_WCAP")) : 0;
    auto launch = [&](auto kernel, int bb, int rows, int tpb) {
        const int per_row = LDS_BUDGET / (rows * 8);                 // doubles per row of the image
        int wcap = std::max(1, std::min(P.maxw, (per_row - 1) / bb));
        if (wcap_env > 0) wcap = std::min(wcap, wcap_env);
        const int len = wcap * bb;
        const int rs_pad = (((len + 30) / 32) * 32 + 1 <= per_row) ? 1 : 0;
        const int RS = rs_pad ? ((len + 30) / 32) * 32 + 1 : (len | 1);
        const size_t shmem = (size_t)rows * RS * sizeof(double);
        FEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        const int nparts = A->nslices * (64 / rows), lpr = tpb / rows;
        const int npass = std::max(1, (P.max_valence + lpr - 1) / lpr);
        const int grid = ((nparts + 7) / 8) * 8;   // the XCD-wise part map wants a multiple of 8
        hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(tpb), shmem, s, m->nnds, nparts, npass, A->boff.p,
                           A->n2e_ptr.p, A->n2e.p, gs, m->coords.p, P.slice_serial.p, wcap, rs_pad, add, A->val.p, uni);
    };
    // A mesh of one group whose elements all have the blocks of its first one, bit for bit (config 3: bricks of one size and material
    // at coordinates that are exact): found out by the SECOND assembly on the mesh (mode 1: the general kernel, comparing as it goes),
    // used from the third on (mode 2).  Coordinates and materials of a mesh do not change after fem_mesh_create.
    // FEM_AMD_ASM_UNIFORM=0: always mode 0.
    static const bool uni_env = !(std::getenv("FEM_AMD_ASM_UNIFORM") && std::atoi(std::getenv("FEM_AMD_ASM_UNIFORM")) == 0);
    int mode = 0;
    if (uni_env && gs.ng == 1 && m->groups[0].nel > 0) {
        auto& g = m->groups[0];
        if (g.ke_state == 1) {   // the verdict of the first assembly
            int h = 1;
            g.ke_diff.download(&h, 1, s);
            FEM_HIP(hipStreamSynchronize(s));
            g.ke_state = h ? 3 : 2;
        }
        if (g.ke_state == 0) g.ke_state = 4;   // (the first assembly on a mesh is not burdened with the comparison: a host that assembles once pays nothing for it)
        else if (g.ke_state == 4) {
            const int bb = m->pdim * m->pdim;
            g.ke_ref.alloc((size_t)g.npe * g.npe * bb);
            g.ke_diff.alloc(1); g.ke_diff.zero(s);
            switch (g.type) {
            case FEM_HEX: hipLaunchKernelGGL((k_ke_ref<3, 8, 8, FEM_HEX>), dim3(1), dim3(64), 0, s, gs.g[0], m->coords.p, g.ke_ref.p); break;
            case FEM_TET: hipLaunchKernelGGL((k_ke_ref<3, 4, 1, FEM_TET>), dim3(1), dim3(64), 0, s, gs.g[0], m->coords.p, g.ke_ref.p); break;
            case FEM_QUA: hipLaunchKernelGGL((k_ke_ref<2, 4, 4, FEM_QUA>), dim3(1), dim3(64), 0, s, gs.g[0], m->coords.p, g.ke_ref.p); break;
            default: hipLaunchKernelGGL((k_ke_ref<2, 3, 1, FEM_TRI>), dim3(1), dim3(64), 0, s, gs.g[0], m->coords.p, g.ke_ref.p); break;
            }
            g.ke_state = 1;
            mode = 1;
        } else if (g.ke_state == 2) mode = 2;
        uni = AsmUniform{g.ke_ref.p, g.ke_diff.p};
    }
    if (m->pdim == 3) {   // (16 rows per workgroup, four per CU: 3.55 against 3.31 ms, round 6)
        if (mode == 2) launch(k_assemble_rows<3, 8, 32, 2>, 9, 32, 256);   // (16 rows per workgroup: 2.89 against 2.81 ms)
        else if (mode == 1) launch(k_assemble_rows<3, 8, 32, 1>, 9, 32, 256);
        else launch(k_assemble_rows<3, 8, 32, 0>, 9, 32, 256);
    } else {
        if (mode == 2) launch(k_assemble_rows<2, 4, 64, 2>, 4, 64, 256);
        else if (mode == 1) launch(k_assemble_rows<2, 4, 64, 1>, 4, 64, 256);
        else launch(k_assemble_rows<2, 4, 64, 0>, 4, 64, 256);
    }
    A->lazy_zero = false;
    FEM_HIP(hipGetLastError());
}

// ---- MatMult by element-by-element recomputation (`-mat_mult_type ebe`) -----------------------
// y = K x without reading the assembled matrix: every element recomputes f_e = K_e x_e at its Gauss
// points (J, dN/dx, strain, stress, B^T sigma w detJ: ~3.2 kFMA per hex8 against 7 GB of stored
// blocks), writes its npe*pdim nodal forces to an element-major buffer, and a second kernel sums, for
// every node, the entries of its incident elements in ascending (group, element) order: no atomics,
// bitwise reproducible.  Penalty rows: ApplyKBC INSERTs on the diagonal only, so y_i += (inserted -
// assembled diagonal) * x_i.  p.Kp comes for free as sum_e x_e.f_e (+ the penalty term), the same
// un-assembled identity the partitioned product uses.
// Traffic at config 3: 0.11 GB connectivity + 1.27 GB element buffer (write + read) + gathers from
// L2, i.e. 1.5 GB instead of 7.4 GB; the element kernel is bound by the fp64 FMA pipe.
//   VAR 0: Gauss loop against dN/dxi (tri3, quad4, tet4; hex8 only for measurement)   1: hex8 in the monomial basis, with the
//   wave-level short cut for parallelepipeds   2: the same without the short cut (measurement)   3: hex8 workgroups the set-up
//   found to hold parallelepipeds only (128 registers, four waves per SIMD)
template <int PD, int NPE, int NIP, int VAR = 0>
__global__ void __launch_bounds__(256, VAR == 3 ? 4 : 2) k_ebe_apply(int type, int nel, int nblk, const int* __restrict__ blocks,
                                                      const int* __restrict__ conn, const double* __restrict__ coords,
                                                      const double* __restrict__ Ev, const double* __restrict__ nuv,
                                                      const double* __restrict__ x, const unsigned char* __restrict__ donate,
                                                      double* __restrict__ f, double* __restrict__ partial,
                                                      const int* __restrict__ done)
{
    if (done && *done) return;
    constexpr bool PAIR = (NPE == 8 || (NPE == 4 && PD == 2));
    constexpr int NP = NPE / 2;
    constexpr int plus8[4] = {1, 2, 5, 6}, minus8[4] = {0, 3, 4, 7};   // quad4 uses the first two pairs: (1,0) (2,3)
    __shared__ double sh[4];
    const ElemTable& tab = c_tab[type];
    // Workgroups b and b + 8 share an XCD (and its L2): every XCD takes a contiguous eighth of the workgroup list, so that
    // elements that share nodes across rows and layers of the mesh meet in one L2 (the grid is a multiple of 8)
    const unsigned chunk = gridDim.x / 8, lb = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
    if (lb >= (unsigned)nblk) return;
    const unsigned blk = blocks ? (unsigned)blocks[lb] : lb;
    const size_t e = (size_t)blk * blockDim.x + threadIdx.x;
    const bool active = e < (size_t)nel;
    const bool don = PAIR && active && donate[e] != 0;   // the +x face forces go to the next lane (set-up)
    double dot = 0.0;
    double fe[NPE][PD];
#pragma unroll
    for (int a = 0; a < NPE; a++)
#pragma unroll
        for (int d = 0; d < PD; d++) fe[a][d] = 0.0;
    if (active) {
        double X[NPE][PD], u[NPE][PD];
#pragma unroll
        for (int a = 0; a < NPE; a++) {
            const size_t v = (size_t)conn[e * NPE + a];
#pragma unroll
            for (int d = 0; d < PD; d++) {
                // the affine-only kernel reads the coordinates of one corner and its three neighbours
                if (VAR != 3 || a == 0 || a == 1 || a == 3 || a == 4) X[a][d] = coords[v * PD + d];
                u[a][d] = x[v * PD + d];
            }
        }
        const DCoef D = dmat_coef(Ev[e], nuv[e]);
        if constexpr (VAR == 3 && PD == 3 && NPE == 8 && NIP == 8) {
            hex_forces_affine(X, u, fabs(tab.xi[0][0]), tab.w[0], D.d12, D.d44, fe, dot);
        } else if constexpr (VAR != 0 && PD == 3 && NPE == 8 && NIP == 8) {
            hex_forces<VAR == 1>(X, u, fabs(tab.xi[0][0]), tab.w[0], D.d12, D.d44, fe, dot);
        } else {
#pragma unroll
        for (int q = 0; q < NIP; q++) {
            // Everything in reference coordinates, so that dN/dx (npe*pdim values per point) is never formed:
            //   J = dNxi X, G = dNxi x_e (reference gradients), H = grad_x u = J^-1 G, sigma = D eps(H),
            //   T = w detJ sigma J^-1, f_a += T dNxi_a.  The dNxi are wave-uniform (scalar registers).
            double J[PD][PD], G[PD][PD];     // J[i][j] = d x_j / d xi_i,  G[i][l] = d u_l / d xi_i
#pragma unroll
            for (int i = 0; i < PD; i++)
#pragma unroll
                for (int j = 0; j < PD; j++) {
                    double sj = 0.0, sg = 0.0;
#pragma unroll
                    for (int a = 0; a < NPE; a++) { sj += tab.dNxi[q][i][a] * X[a][j]; sg += tab.dNxi[q][i][a] * u[a][j]; }
                    J[i][j] = sj; G[i][j] = sg;
                }
            double inv[PD][PD], detj;        // inv[i][j] = d xi_j / d x_i, as in dndx_at_gp
            if constexpr (PD == 2) {
                detj = J[0][0] * J[1][1] - J[0][1] * J[1][0];
                const double rd = 1.0 / detj;
                inv[0][0] = J[1][1] * rd; inv[0][1] = -J[0][1] * rd;
                inv[1][0] = -J[1][0] * rd; inv[1][1] = J[0][0] * rd;
            } else {
                const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
                const double c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
                const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
                detj = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
                const double rd = 1.0 / detj;
                inv[0][0] = c00 * rd;
                inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * rd;
                inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * rd;
                inv[1][0] = -c01 * rd;
                inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * rd;
                inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * rd;
                inv[2][0] = c02 * rd;
                inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * rd;
                inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rd;
            }
            double H[PD][PD];                // H[l][i] = d u_l / d x_i
#pragma unroll
            for (int l = 0; l < PD; l++)
#pragma unroll
                for (int i = 0; i < PD; i++) {
                    double h = 0.0;
#pragma unroll
                    for (int j = 0; j < PD; j++) h += inv[i][j] * G[j][l];
                    H[l][i] = h;
                }
            const double wd = tab.w[q] * detj;   // weights(i)*detj, src/m_local.F90:64
            double S[PD][PD];                // stress (DMat, src/m_local.F90:204; Voigt order of BMat :160-168) * w detJ
            if constexpr (PD == 2) {
                S[0][0] = wd * (D.d11 * H[0][0] + D.d12 * H[1][1]);
                S[1][1] = wd * (D.d12 * H[0][0] + D.d11 * H[1][1]);
                S[0][1] = S[1][0] = wd * D.d44 * (H[0][1] + H[1][0]);
            } else {
                S[0][0] = wd * (D.d11 * H[0][0] + D.d12 * H[1][1] + D.d12 * H[2][2]);
                S[1][1] = wd * (D.d12 * H[0][0] + D.d11 * H[1][1] + D.d12 * H[2][2]);
                S[2][2] = wd * (D.d12 * H[0][0] + D.d12 * H[1][1] + D.d11 * H[2][2]);
                S[0][1] = S[1][0] = wd * D.d44 * (H[0][1] + H[1][0]);
                S[1][2] = S[2][1] = wd * D.d44 * (H[1][2] + H[2][1]);
                S[0][2] = S[2][0] = wd * D.d44 * (H[2][0] + H[0][2]);
            }
            double T[PD][PD];                // T[l][j] = sum_i S[l][i] inv[i][j]
#pragma unroll
            for (int l = 0; l < PD; l++)
#pragma unroll
                for (int j = 0; j < PD; j++) {
                    double t = 0.0;
#pragma unroll
                    for (int i = 0; i < PD; i++) t += S[l][i] * inv[i][j];
                    T[l][j] = t;
                }
#pragma unroll
            for (int a = 0; a < NPE; a++)
#pragma unroll
                for (int l = 0; l < PD; l++) {
                    double t = fe[a][l];
#pragma unroll
                    for (int j = 0; j < PD; j++) t += T[l][j] * tab.dNxi[q][j][a];
                    fe[a][l] = t;
                }
        }
#pragma unroll
        for (int a = 0; a < NPE; a++)
#pragma unroll
            for (int d = 0; d < PD; d++) dot += u[a][d] * fe[a][d];
        }
    }
    // hand-over along the wavefront (every lane takes part in the cross-lane moves): the +x face forces
    // of lane l-1 are added to the -x face nodes of lane l (same global nodes, checked at set-up), and
    // lane l-1 does not store them
    if constexpr (PAIR) {
        const bool recv = (__shfl_up((int)don, 1) != 0) && ((threadIdx.x & 63) != 0);
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int pa = plus8[k], ma = minus8[k];
#pragma unroll
            for (int d = 0; d < PD; d++) {
                const double v = __shfl_up(fe[pa][d], 1);
                if (recv) fe[ma][d] += v;
            }
        }
    }
    if (active) {
#pragma unroll
        for (int a = 0; a < NPE; a++) {
            if (PAIR && don && plus_face_x<NPE>(a)) continue;
#pragma unroll
            for (int d = 0; d < PD; d++) f[(size_t)(a * PD + d) * nel + e] = fe[a][d];   // component-major: coalesced
        }
    }
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partial[blk] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// donate[e] = 1 when element e+1 sits in the next lane of the same wavefront and its -x face is e's +x face
template <int NPE>
__global__ void __launch_bounds__(256) k_ebe_mark_donors(int nel, const int* __restrict__ conn,
                                                         unsigned char* __restrict__ donate)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    constexpr int NP = NPE / 2;
    constexpr int plus8[4] = {1, 2, 5, 6}, minus8[4] = {0, 3, 4, 7};
    bool ok = ((e & 63) != 63) && (e + 1 < nel);
    if (ok)
        for (int k = 0; k < NP; k++)
            ok = ok && conn[(size_t)e * NPE + plus8[k]] == conn[(size_t)(e + 1) * NPE + minus8[k]];
    donate[e] = ok ? 1 : 0;
}
// hex8 workgroups (256 elements) whose elements are all parallelepipeds: the test of hex_forces, once per mesh
__global__ void __launch_bounds__(256) k_ebe_classify_hex8(int nel, const int* __restrict__ conn, const double* __restrict__ coords,
                                                           double g, unsigned char* __restrict__ blk_affine)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    bool affine = true;
    if (e < nel) {
        const double g8 = 0.125 * g, gg8 = g8 * g;
        for (int j = 0; j < 3; j++) {
            double v[8], c0;
            for (int a = 0; a < 8; a++) v[a] = coords[(size_t)conn[(size_t)e * 8 + a] * 3 + j];
            const Mono c = hex_mono(v, g8, gg8, c0);
            const double tol = 0x1p-50 * (fabs(c0) + fabs(c.x) + fabs(c.y) + fabs(c.z));
            affine = affine && fabs(c.xy) <= tol && fabs(c.xz) <= tol && fabs(c.yz) <= tol && fabs(c.xyz) <= tol;
        }
    }
    const int all = __syncthreads_and(affine ? 1 : 0);
    if (threadIdx.x == 0) blk_affine[blockIdx.x] = all ? 1 : 0;
}
struct DonateViews {
    const unsigned char* d[4];
    int npe[4];
};
__device__ __forceinline__ bool ebe_entry_kept(const DonateViews& dv, unsigned code)
{
    const int g = code >> 30;
    const unsigned char* d = dv.d[g];
    if (!d) return true;
    const size_t e = (code & 0x3fffffffu) >> 3;
    const int a = code & 7u;
    const bool plus = (dv.npe[g] == 8) ? plus_face_x<8>(a) : plus_face_x<4>(a);
    return !(plus && d[e]);
}
// gather lists without the handed-over entries: count, (scan), fill -- order of the sorted lists kept
__global__ void __launch_bounds__(256) k_ebe_count_kept(int nn, const int* __restrict__ ptr, const unsigned* __restrict__ lst,
                                                        DonateViews dv, int* __restrict__ cnt)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    int c = 0;
    for (int k = ptr[v]; k < ptr[v + 1]; k++) c += ebe_entry_kept(dv, lst[k]) ? 1 : 0;
    cnt[v] = c;
}
__global__ void __launch_bounds__(256) k_ebe_fill_kept(int nn, const int* __restrict__ ptr, const unsigned* __restrict__ lst,
                                                       DonateViews dv, const int* __restrict__ ptr2,
                                                       unsigned* __restrict__ lst2)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    int o = ptr2[v];
    for (int k = ptr[v]; k < ptr[v + 1]; k++)
        if (ebe_entry_kept(dv, lst[k])) lst2[o++] = lst[k];
}

struct EbeViews {
    const double* f[4];
    int nel[4];
};
// thread = node: ordered sum of the buffer entries of the incident elements, then the node's penalty rows
// (y_i += shift_i x_i, their share of x.Kx into one partial per workgroup)
template <int PD>
__global__ void __launch_bounds__(256) k_ebe_gather(int nn, const int* __restrict__ ptr, const unsigned* __restrict__ lst,
                                                    EbeViews ev, const int* __restrict__ bc_node,
                                                    const int* __restrict__ bc_dof, const double* __restrict__ bc_shift,
                                                    const double* __restrict__ x, double* __restrict__ y,
                                                    double* __restrict__ partial, const int* __restrict__ done)
{
    if (done && *done) return;
    __shared__ double sh[4];
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    double pen = 0.0;
    if (v < nn) {
        double s[PD];
#pragma unroll
        for (int d = 0; d < PD; d++) s[d] = 0.0;
        for (int k = ptr[v]; k < ptr[v + 1]; k++) {
            const unsigned code = lst[k];
            const int g = code >> 30;
            const size_t e = (code & 0x3fffffffu) >> 3, a = code & 7u, ne = (size_t)ev.nel[g];
#pragma unroll
            for (int d = 0; d < PD; d++) s[d] += ev.f[g][(a * PD + d) * ne + e];
        }
        const int kb = bc_node ? bc_node[v] : -1;
        if (kb >= 0) {
#pragma unroll
            for (int d = 0; d < PD; d++)
                if (bc_dof[kb + d] >= 0) {
                    const double xi = x[(size_t)v * PD + d], t = bc_shift[kb + d] * xi;
                    s[d] += t;
                    pen += t * xi;
                }
        }
#pragma unroll
        for (int d = 0; d < PD; d++) y[(size_t)v * PD + d] = s[d];
    }
    for (int o = 32; o > 0; o >>= 1) pen += __shfl_down(pen, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = pen;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

static int ebe_blocks(const fem_mesh m)
{
    int nb = 0;
    for (auto& g : m->groups) nb += (int)cdiv((size_t)g.nel, 256);
    return nb + (int)cdiv((size_t)m->nnds, 256);   // element workgroups + gather workgroups (penalty rows)
}

std::shared_ptr<EbeTilePlan> ebe_tile_build(fem_mesh m, hipStream_t s);   // ebe_tile.hip
int ebe_tile_partials(const EbeTilePlan& P);
void ebe_tile_mult(fem_mat A, const EbeTilePlan& P, const double* x, double* y, double* part, const int* done, int phase,
                   const CgFuseArgs* cg = nullptr);
bool ebe_tile_can_fuse(fem_mat A, const EbeTilePlan& P);
bool ebe_tile_cg_resident(fem_mat A, const EbeTilePlan& P, const CgResidentArgs* cg);
bool ebe_tile_stress(fem_mesh m, const EbeTilePlan& P, const double* U, double* S, const double* valence);
bool ebe_tile_split_valid(fem_mat A, const EbeTilePlan& P);

void mat_set_mult_type_impl(fem_mat A, const char* type)
{
    const std::string t = type;
    FEM_CHECK(t == "assembled" || t == "ebe" || t == "ebe_gather", 6,
              "fem_mat_set_mult_type: unsupported type " + t + " (assembled|ebe|ebe_gather)");
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const bool ebe = (t == "ebe" || t == "ebe_gather");
    FEM_CHECK(!(ebe && A->unsymmetric), 6, "fem_mat_set_mult_type: this Mat carries cohesive tangents, which the element-by-element "
                                            "product (solid elements only) does not know: -mat_mult_type assembled");
    fem_mesh m = A->mesh;
    MatPattern& P = *A->pat;
    // `ebe` on a mesh of hex8 (or of quad4) elements alone: the one-pass product by tiles (ebe_tile.hip); `ebe_gather` and
    // every other mesh: element buffer + node gather.  FEM_AMD_EBE_TILE=0 keeps `ebe` on the latter as well (measurement).
    static const bool tile_env = !(std::getenv("FEM_AMD_EBE_TILE") && std::atoi(std::getenv("FEM_AMD_EBE_TILE")) == 0);
    if (t == "ebe" && tile_env && !P.ebe_tile_tried) {
        P.ebe_tile = ebe_tile_build(m, s);
        P.ebe_tile_tried = true;
    }
    A->ebe_use_tile = (t == "ebe") && tile_env && P.ebe_tile != nullptr;
    // the two-kernel product (element buffer + node gather), where that is the one in use: once per mesh its hand-over flags,
    // workgroup classes and gather lists
    if (ebe && !A->ebe_use_tile && !P.ebe_ready) {
        FEM_CHECK(A->n2e_ptr.p && A->n2e.p, 5, "fem_mat_set_mult_type: matrix was not created from a mesh");
        P.ebe_plan.clear();
        P.ebe_plan.resize(m->groups.size());
        DonateViews dv{};
        for (size_t g = 0; g < m->groups.size(); g++) {
            auto& grp = m->groups[g];
            auto& pl = P.ebe_plan[g];
            const unsigned nb = cdiv((size_t)grp.nel, 256);
            dv.npe[g] = grp.npe; dv.d[g] = nullptr;
            if (grp.type == FEM_HEX || grp.type == FEM_QUA) {
                pl.donate.alloc((size_t)grp.nel);
                if (grp.type == FEM_HEX)
                    hipLaunchKernelGGL(k_ebe_mark_donors<8>, dim3(nb), dim3(256), 0, s, grp.nel, grp.conn.p, pl.donate.p);
                else
                    hipLaunchKernelGGL(k_ebe_mark_donors<4>, dim3(nb), dim3(256), 0, s, grp.nel, grp.conn.p, pl.donate.p);
                dv.d[g] = pl.donate.p;
            }
            if (grp.type == FEM_HEX) {   // workgroups of parallelepipeds go to the affine kernel
                DBuf<unsigned char> cls((size_t)nb);
                const ElemTable& tab = host_elem_tables()[FEM_HEX];
                hipLaunchKernelGGL(k_ebe_classify_hex8, dim3(nb), dim3(256), 0, s, grp.nel, grp.conn.p, m->coords.p,
                                   std::fabs(tab.xi[0][0]), cls.p);
                std::vector<unsigned char> h_cls((size_t)nb);
                FEM_HIP(hipMemcpyAsync(h_cls.data(), cls.p, (size_t)nb, hipMemcpyDeviceToHost, s));
                FEM_HIP(hipStreamSynchronize(s));
                std::vector<int> ba, bg;
                for (unsigned b = 0; b < nb; b++) (h_cls[b] ? ba : bg).push_back((int)b);
                pl.n_affine = (int)ba.size(); pl.n_general = (int)bg.size();
                pl.blk_affine.alloc(std::max<size_t>(ba.size(), 1)); pl.blk_general.alloc(std::max<size_t>(bg.size(), 1));
                if (!ba.empty()) FEM_HIP(hipMemcpyAsync(pl.blk_affine.p, ba.data(), ba.size() * sizeof(int), hipMemcpyHostToDevice, s));
                if (!bg.empty()) FEM_HIP(hipMemcpyAsync(pl.blk_general.p, bg.data(), bg.size() * sizeof(int), hipMemcpyHostToDevice, s));
                FEM_HIP(hipStreamSynchronize(s));
            }
        }
        {
            const int nn = m->nnds;
            DBuf<int> cnt((size_t)nn), tmp(exclusive_scan_tmp_ints((size_t)nn + 1));
            hipLaunchKernelGGL(k_ebe_count_kept, dim3(cdiv((size_t)nn, 256)), dim3(256), 0, s, nn, A->n2e_ptr.p, A->n2e.p, dv,
                               cnt.p);
            P.ebe_ptr.alloc((size_t)nn + 1);
            exclusive_scan_i32(cnt.p, P.ebe_ptr.p, nn, tmp.p, s);
            int tot = 0;
            FEM_HIP(hipMemcpyAsync(&tot, P.ebe_ptr.p + nn, sizeof(int), hipMemcpyDeviceToHost, s));
            FEM_HIP(hipStreamSynchronize(s));
            P.ebe_lst.alloc((size_t)std::max(tot, 1));
            hipLaunchKernelGGL(k_ebe_fill_kept, dim3(cdiv((size_t)nn, 256)), dim3(256), 0, s, nn, A->n2e_ptr.p, A->n2e.p, dv,
                               P.ebe_ptr.p, P.ebe_lst.p);
            FEM_HIP(hipStreamSynchronize(s));   // cnt, tmp go out of scope
        }
        P.ebe_grid = ebe_blocks(m);
        FEM_HIP(hipGetLastError());
        FEM_HIP(hipStreamSynchronize(s));
        P.ebe_ready = true;
    }
    if (ebe) A->ebe_grid = A->ebe_use_tile ? 0 : P.ebe_grid;
    if (ebe && !A->ebe_use_tile && A->ebe_f.empty()) {   // this Mat's element-force buffers
        A->ebe_f.resize(m->groups.size());
        for (size_t g = 0; g < m->groups.size(); g++)
            A->ebe_f[g].alloc((size_t)m->groups[g].nel * m->groups[g].npe * m->pdim);
    }
    A->mult_ebe = ebe;
    if (A->ebe_use_tile && !A->red_counter.p) { A->red_counter.alloc(1); A->red_counter.zero(s); }
    // the Krylov solver sums A->spmv_partial[0 .. spmv_grid): size it for the product in use
    A->spmv_grid = A->ebe_use_tile ? ebe_tile_partials(*P.ebe_tile) : ebe ? A->ebe_grid : (A->nslices + 3) / 4;
    A->spmv_partial.alloc((size_t)A->spmv_grid);
    if (ebe) A->ebe_partial.alloc((size_t)std::max(A->ebe_grid, A->spmv_grid));
}

// profiling: one event before the element kernels and one after each stage of a sampled product
static void ebe_mark(fem_mat A)
{
    if (!A->sampling) return;
    if (A->pev_used == A->pev.size()) { hipEvent_t e; FEM_HIP(hipEventCreate(&e)); A->pev.push_back(e); }
    FEM_HIP(hipEventRecord(A->pev[A->pev_used++], A->ctx->stream));
}
static EbeViews ebe_views(fem_mat A)
{
    EbeViews ev{};
    for (size_t gi = 0; gi < A->mesh->groups.size(); gi++) { ev.f[gi] = A->ebe_f[gi].p; ev.nel[gi] = A->mesh->groups[gi].nel; }
    return ev;
}
// the element kernels of one product: nodal forces of every element into the buffers, x_e.f_e partials into part[0 .. return)
int ebe_apply_elements(fem_mat A, const double* x, double* part, const int* done)
{
    fem_mesh m = A->mesh;
    hipStream_t s = A->ctx->stream;
    A->sampling = A->profiling > 0 && (A->prod_count++ % (size_t)A->profiling) == 0;
    ebe_mark(A);
    int off = 0;
    // FEM_AMD_EBE_VAR (measurement): 0 = Gauss loop against dN/dxi for hex8 too (round 1), 2 = monomial-basis general path for
    // every hex8 workgroup (no affine kernel, no affine shortcut)
    static const int var = std::getenv("FEM_AMD_EBE_VAR") ? std::atoi(std::getenv("FEM_AMD_EBE_VAR")) : 1;
    for (size_t gi = 0; gi < m->groups.size(); gi++) {
        auto& g = m->groups[gi];
        auto& pl = A->pat->ebe_plan[gi];
        const unsigned nb = cdiv((size_t)g.nel, 256);
        dispatch_type(g.type, [&](auto PD, auto NPE, auto NIP, auto) {
            auto go = [&](auto kern, unsigned nblk, const int* blocks) {
                if (!nblk) return;
                hipLaunchKernelGGL(kern, dim3((nblk + 7) / 8 * 8), dim3(256), 0, s, g.type, g.nel, (int)nblk, blocks, g.conn.p,
                                   m->coords.p, g.E.p, g.nu.p, x, pl.donate.p, A->ebe_f[gi].p, part + off, done);
            };
            if constexpr (NPE() == 8) {
                if (var == 1) {
                    go(k_ebe_apply<PD(), NPE(), NIP(), 3>, (unsigned)pl.n_affine, pl.blk_affine.p);
                    go(k_ebe_apply<PD(), NPE(), NIP(), 1>, (unsigned)pl.n_general, pl.blk_general.p);
                    return;
                }
                if (var == 2) { go(k_ebe_apply<PD(), NPE(), NIP(), 2>, nb, nullptr); return; }
            }
            go(k_ebe_apply<PD(), NPE(), NIP(), 0>, nb, nullptr);
        });
        off += (int)nb;
    }
    ebe_mark(A);
    return off;
}
// the one-pass product in two parts for a partitioned mesh (ksp.hip): phase 0 = the tiles that own nodes shared with other
// domains, phase 1 = the rest; false when this Mat's product cannot be split that way
bool ebe_tile_phase(fem_mat A, const double* x, double* y, double* partial, const int* done, int phase)
{
    if (!A->mult_ebe || !A->ebe_use_tile || !ebe_tile_split_valid(A, *A->pat->ebe_tile)) return false;
    if (phase == 0) {   // (the profile's marks: before the first part, after the second)
        A->sampling = A->profiling > 0 && (A->prod_count++ % (size_t)A->profiling) == 0;
        ebe_mark(A);
    }
    if (phase >= 0) ebe_tile_mult(A, *A->pat->ebe_tile, x, y, partial ? partial : A->ebe_partial.p, done, phase);
    if (phase == 1) { ebe_mark(A); ebe_mark(A); ebe_mark(A); }
    return true;
}

// The product that opens the CG iteration itself (cg_device.h): cg == nullptr asks whether this Mat's product can (quad4 tile plans
// with x in LDS, no penalty-free shortcuts needed); otherwise y = K p with p formed in the stage-in, p.Kp partials as ebe_mult_impl.
bool ebe_mult_fused_impl(fem_mat A, const CgFuseArgs* cg, double* y, double* partial)
{
    if (!A->mult_ebe || !A->ebe_use_tile || !ebe_tile_can_fuse(A, *A->pat->ebe_tile)) return false;
    if (!cg) return true;
    A->sampling = A->profiling > 0 && (A->prod_count++ % (size_t)A->profiling) == 0;
    ebe_mark(A);
    ebe_tile_mult(A, *A->pat->ebe_tile, nullptr, y, partial ? partial : A->ebe_partial.p, nullptr, -1, cg);
    ebe_mark(A); ebe_mark(A); ebe_mark(A);
    return true;
}

// KSPCG / PCJACOBI as one resident launch (ebe_tile.hip): cg == nullptr asks whether this Mat's product qualifies
bool ebe_cg_resident_impl(fem_mat A, const CgResidentArgs* cg)
{
    if (!A->mult_ebe || !A->ebe_use_tile || !A->pat->ebe_tile) return false;
    return ebe_tile_cg_resident(A, *A->pat->ebe_tile, cg);
}

void ebe_mult_impl(fem_mat A, const double* x, double* y, double* partial, const int* done)
{
    fem_mesh m = A->mesh;
    hipStream_t s = A->ctx->stream;
    double* part = partial ? partial : A->ebe_partial.p;
    if (A->ebe_use_tile) {
        A->sampling = A->profiling > 0 && (A->prod_count++ % (size_t)A->profiling) == 0;
        ebe_mark(A);
        ebe_tile_mult(A, *A->pat->ebe_tile, x, y, part, done, -1);
        ebe_mark(A); ebe_mark(A); ebe_mark(A);   // (one stage: the profile keeps its four marks per product)
        return;
    }
    const int off = ebe_apply_elements(A, x, part, done);
    const EbeViews ev = ebe_views(A);
    const int* bcn = A->bc_saved ? A->bc_node.p : nullptr;
    if (m->pdim == 3)
        hipLaunchKernelGGL(k_ebe_gather<3>, dim3(cdiv((size_t)m->nnds, 256)), dim3(256), 0, s, m->nnds, A->pat->ebe_ptr.p,
                           A->pat->ebe_lst.p, ev, bcn, A->bc_dof.p, A->bc_shift.p, x, y, part + off, done);
    else
        hipLaunchKernelGGL(k_ebe_gather<2>, dim3(cdiv((size_t)m->nnds, 256)), dim3(256), 0, s, m->nnds, A->pat->ebe_ptr.p,
                           A->pat->ebe_lst.p, ev, bcn, A->bc_dof.p, A->bc_shift.p, x, y, part + off, done);
    ebe_mark(A);
    ebe_mark(A);   // (the penalty rows used to be a stage of their own: the profile keeps its four marks per product)
}

void ebe_tile_info(const EbeTilePlan& P, long long* tiles, long long* affine_tiles, long long* visits, long long* slots,
                   long long* nodes, double* ms);
void mat_get_mult_info_impl(fem_mat A, int* kind, long long* tiles, long long* affine_tiles, long long* element_visits,
                            long long* lane_slots, long long* tile_nodes, double* setup_ms)
{
    long long z = 0;
    double zd = 0.0;
    if (kind) *kind = !A->mult_ebe ? 0 : A->ebe_use_tile ? 2 : 1;
    if (tiles) *tiles = 0;
    if (affine_tiles) *affine_tiles = 0;
    if (element_visits) *element_visits = 0;
    if (lane_slots) *lane_slots = 0;
    if (tile_nodes) *tile_nodes = 0;
    if (setup_ms) *setup_ms = 0.0;
    if (A->mult_ebe && A->ebe_use_tile)
        ebe_tile_info(*A->pat->ebe_tile, tiles ? tiles : &z, affine_tiles ? affine_tiles : &z, element_visits ? element_visits : &z,
                      lane_slots ? lane_slots : &z, tile_nodes ? tile_nodes : &z, setup_ms ? setup_ms : &zd);
}

// one-pass tiles: how many tiles keep one geometry record for all their elements, whether the product uses that (every tile of
// the plan does) and whether the tile lists are loaded with the streaming hint
void ebe_tile_geometry_info(const EbeTilePlan& P, long long* uniform_tiles, int* per_tile_records, int* streaming_lists);   // ebe_tile.hip
void mat_get_tile_geometry_info_impl(fem_mat A, long long* uniform_tiles, int* per_tile_records, int* streaming_lists)
{
    if (uniform_tiles) *uniform_tiles = 0;
    if (per_tile_records) *per_tile_records = 0;
    if (streaming_lists) *streaming_lists = 0;
    if (!(A->mult_ebe && A->ebe_use_tile)) return;
    ebe_tile_geometry_info(*A->pat->ebe_tile, uniform_tiles, per_tile_records, streaming_lists);
}

void mat_set_profiling_impl(fem_mat A, int on)
{
    A->profiling = on > 0 ? on : 0;
    A->pev_used = 0;
    A->prod_count = 0;
}
// sums over the sampled products among the first `max_products` element-by-element products since profiling
// was switched on (sample j is product j * period)
void mat_get_profile_impl(fem_mat A, int max_products, int* n, double* element_ms, double* gather_ms, double* penalty_ms)
{
    A->ctx->bind();
    const size_t period = (size_t)std::max(A->profiling, 1);
    const size_t have = A->pev_used / 4;
    const size_t allowed = ((size_t)std::max(max_products, 0) + period - 1) / period;
    const size_t take = std::min(have, allowed);
    double t[3] = {0.0, 0.0, 0.0};
    if (take) FEM_HIP(hipEventSynchronize(A->pev[4 * take - 1]));
    for (size_t k = 0; k < take; k++)
        for (int j = 0; j < 3; j++) {
            float ms = 0.f;
            FEM_HIP(hipEventElapsedTime(&ms, A->pev[4 * k + j], A->pev[4 * k + j + 1]));
            t[j] += ms;
        }
    if (n) *n = (int)take;
    if (element_ms) *element_ms = t[0];
    if (gather_ms) *gather_ms = t[1];
    if (penalty_ms) *penalty_ms = t[2];
}

// ---- the small routines of module `local` / `elems`, batched (element-level API) -----------------
// ShapeFuncd (src/m_elems.F90:177-185): dN(pdim,n) column-major at npts arbitrary points
__global__ void __launch_bounds__(128) k_shape_funcd(int type, int npts, int pd, int n, const double* __restrict__ coord,
                                                     double* __restrict__ dN)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npts) return;
    double x[3] = {0, 0, 0}, d[3][8];
    for (int i = 0; i < pd; i++) x[i] = coord[(size_t)p * pd + i];
    elem_shape_d(type, x, d);
    for (int a = 0; a < n; a++)
        for (int i = 0; i < pd; i++) dN[(size_t)p * pd * n + i + (size_t)a * pd] = d[i][a];
}
// FormdNdetJ (src/m_local.F90:175-189): every element at its own point; dN(pdim,n) column-major, signed detj
template <int PD, int NPE>
__global__ void __launch_bounds__(128) k_form_dn_detj(int type, int nel, const double* __restrict__ ipcoord,
                                                      const double* __restrict__ ecoords, double* __restrict__ dN,
                                                      double* __restrict__ detj)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    double x[3] = {0, 0, 0}, d[3][8], X[NPE][PD], g[NPE][PD], dj;
    for (int i = 0; i < PD; i++) x[i] = ipcoord[(size_t)e * PD + i];
    elem_shape_d(type, x, d);
    DenseSrc<PD, NPE>{ecoords}.coordsOf((size_t)e, X);
    dndx_at_gp<PD, NPE>(d, X, g, dj);
    for (int a = 0; a < NPE; a++)
        for (int i = 0; i < PD; i++) dN[(size_t)e * PD * NPE + i + (size_t)a * PD] = g[a][i];
    detj[e] = dj;
}
// BMat (src/m_local.F90:148-171): B(cpdim, n*pdim) column-major from dN(pdim, n)
__global__ void __launch_bounds__(128) k_bmat(int nel, int pd, int n, const double* __restrict__ dN, double* __restrict__ B)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    const int cp = (pd == 2) ? 3 : 6;
    const double* d = dN + (size_t)e * pd * n;
    double* b = B + (size_t)e * cp * n * pd;
    for (int k = 0; k < cp * n * pd; k++) b[k] = 0.0;
    for (int j = 0; j < n; j++) {
        const double dx = d[0 + j * pd], dy = d[1 + j * pd];
        if (pd == 2) {
            b[0 + (2 * j) * cp] = dx; b[1 + (2 * j + 1) * cp] = dy;
            b[2 + (2 * j) * cp] = dy; b[2 + (2 * j + 1) * cp] = dx;
        } else {
            const double dz = d[2 + j * pd];
            b[0 + (3 * j) * cp] = dx; b[1 + (3 * j + 1) * cp] = dy; b[2 + (3 * j + 2) * cp] = dz;
            b[3 + (3 * j) * cp] = dy; b[3 + (3 * j + 1) * cp] = dx;
            b[4 + (3 * j + 1) * cp] = dz; b[4 + (3 * j + 2) * cp] = dy;
            b[5 + (3 * j) * cp] = dz; b[5 + (3 * j + 2) * cp] = dx;
        }
    }
}
// DMat / DMat2d / DMat3d (src/m_local.F90:204-228)
__global__ void __launch_bounds__(128) k_dmat(int nmat, int pd, const double* __restrict__ E, const double* __restrict__ nu,
                                              double* __restrict__ D)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nmat) return;
    const int cp = (pd == 2) ? 3 : 6, nn = (pd == 2) ? 2 : 3;
    const DCoef c = dmat_coef(E[m], nu[m]);
    double* d = D + (size_t)m * cp * cp;
    for (int k = 0; k < cp * cp; k++) d[k] = 0.0;
    for (int i = 0; i < nn; i++)
        for (int j = 0; j < nn; j++) d[i + j * cp] = (i == j) ? c.d11 : c.d12;
    for (int i = nn; i < cp; i++) d[i + i * cp] = c.d44;
}

void shape_funcd_impl(fem_ctx ctx, int type, int npts, const double* coord, double* dN)
{
    if (npts <= 0) return;
    FEM_CHECK(type >= FEM_TRI && type <= FEM_COH, 3, "fem_shape_funcd: unknown element type");
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int pd = type_dim(type), n = type_npe(type);
    DBuf<double> dC((size_t)npts * pd), dD((size_t)npts * pd * n);
    dC.upload(coord, dC.n, s);
    hipLaunchKernelGGL(k_shape_funcd, dim3(cdiv(npts, 128)), dim3(128), 0, s, type, npts, pd, n, dC.p, dD.p);
    FEM_HIP(hipGetLastError());
    dD.download(dN, dD.n, s);
    FEM_HIP(hipStreamSynchronize(s));
}

void form_dn_detj_impl(fem_ctx ctx, int type, int nel, const double* ipcoord, const double* ecoords, double* dN, double* detj)
{
    if (nel <= 0) return;
    FEM_CHECK(type >= FEM_TRI && type <= FEM_HEX, 3, "fem_form_dn_detj: unsupported element type");
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int pd = type_dim(type), n = type_npe(type);
    DBuf<double> dP((size_t)nel * pd), dX((size_t)nel * n * pd), dD((size_t)nel * pd * n), dJ((size_t)nel);
    dP.upload(ipcoord, dP.n, s); dX.upload(ecoords, dX.n, s);
    dispatch_type(type, [&](auto PD, auto NPE, auto, auto) {
        hipLaunchKernelGGL((k_form_dn_detj<PD(), NPE()>), dim3(cdiv(nel, 128)), dim3(128), 0, s, type, nel, dP.p, dX.p, dD.p, dJ.p);
    });
    FEM_HIP(hipGetLastError());
    dD.download(dN, dD.n, s); dJ.download(detj, dJ.n, s);
    FEM_HIP(hipStreamSynchronize(s));
}

void bmat_impl(fem_ctx ctx, int pd, int n, int nel, const double* dN, double* B)
{
    if (nel <= 0) return;
    FEM_CHECK((pd == 2 || pd == 3) && n >= 1 && n <= 8, 2, "fem_bmat: bad pdim / nodecount");
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int cp = (pd == 2) ? 3 : 6;
    DBuf<double> dD((size_t)nel * pd * n), dB((size_t)nel * cp * n * pd);
    dD.upload(dN, dD.n, s);
    hipLaunchKernelGGL(k_bmat, dim3(cdiv(nel, 128)), dim3(128), 0, s, nel, pd, n, dD.p, dB.p);
    FEM_HIP(hipGetLastError());
    dB.download(B, dB.n, s);
    FEM_HIP(hipStreamSynchronize(s));
}

void dmat_impl(fem_ctx ctx, int pd, int nmat, const double* E, const double* nu, double* D)
{
    if (nmat <= 0) return;
    FEM_CHECK(pd == 2 || pd == 3, 2, "fem_dmat: pdim must be 2 or 3");
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int cp = (pd == 2) ? 3 : 6;
    DBuf<double> dE((size_t)nmat), dn((size_t)nmat), dD((size_t)nmat * cp * cp);
    dE.upload(E, nmat, s); dn.upload(nu, nmat, s);
    hipLaunchKernelGGL(k_dmat, dim3(cdiv(nmat, 128)), dim3(128), 0, s, nmat, pd, dE.p, dn.p, dD.p);
    FEM_HIP(hipGetLastError());
    dD.download(D, dD.n, s);
    FEM_HIP(hipStreamSynchronize(s));
}

// FormElKE for a batch of elements given by their coordinates (host in/out)
void form_el_ke_impl(fem_ctx ctx, int type, int nel, const double* ecoords, const double* E, const double* nu, double* k)
{
    if (nel <= 0) return;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int npe = type_npe(type), pd = type_dim(type), nd = npe * pd;
    FEM_CHECK(type >= FEM_TRI && type <= FEM_HEX, 3, "fem_form_el_ke: unsupported element type");
    DBuf<double> dX((size_t)nel * npe * pd), dE(nel), dnu(nel), dK((size_t)nel * nd * nd);
    dX.upload(ecoords, dX.n, s); dE.upload(E, nel, s); dnu.upload(nu, nel, s);
    dispatch_type(type, [&](auto PD, auto NPE, auto NIP, auto ELB) {
        launch_ke<PD(), NPE(), NIP(), ELB()>(type, nel, DenseSrc<PD(), NPE()>{dX.p}, dE.p, dnu.p,
                                             DenseSink<PD(), NPE()>{dK.p}, s);
    });
    FEM_HIP(hipGetLastError());
    dK.download(k, dK.n, s);
    FEM_HIP(hipStreamSynchronize(s));
}

void calc_el_stress_impl(fem_ctx ctx, int type, int nel, const double* ecoords, const double* edisp, const double* E,
                         const double* nu, double* estress)
{
    if (nel <= 0) return;
    ctx->bind();
    hipStream_t s = ctx->stream;
    FEM_CHECK(type >= FEM_TRI && type <= FEM_HEX, 3, "fem_calc_el_stress: unsupported element type");
    const int npe = type_npe(type), pd = type_dim(type), nip = type_nip(type), cp = (pd == 2) ? 3 : 6;
    DBuf<double> dX((size_t)nel * npe * pd), dU((size_t)nel * npe * pd), dE(nel), dnu(nel), dS((size_t)nel * nip * cp);
    dX.upload(ecoords, dX.n, s); dU.upload(edisp, dU.n, s); dE.upload(E, nel, s); dnu.upload(nu, nel, s);
    dispatch_type(type, [&](auto PD, auto NPE, auto NIP, auto ELB) {
        constexpr int TPB = ELB() * (NIP() > NPE() ? NIP() : NPE());
        hipLaunchKernelGGL((k_stress<PD(), NPE(), NIP(), ELB(), false, DenseSrc<PD(), NPE()>, DenseU<PD(), NPE()>>),
                           dim3(cdiv(nel, ELB())), dim3(TPB), 0, s, type, nel, DenseSrc<PD(), NPE()>{dX.p},
                           DenseU<PD(), NPE()>{dU.p}, dE.p, dnu.p, dS.p);
    });
    FEM_HIP(hipGetLastError());
    dS.download(estress, dS.n, s);
    FEM_HIP(hipStreamSynchronize(s));
}

// node -> incident elements of the mesh's pattern (built by the first fem_mat_create_aij; a mesh that never had
// a Mat gets it here)
std::shared_ptr<fem::MatPattern> mesh_pattern(fem_mesh m);

// main.F90:253-292 on the device: S(nnds,cpdim) column-major
void recover_nodal_stress_dev(fem_mesh m, const double* U, double* S)
{
    fem_ctx ctx = m->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const size_t nn = (size_t)m->nnds;
    const auto P = mesh_pattern(m);
    const bool multi = m->halo && m->halo->comm->size > 1;
    auto finish_multi = [&] {   // partitioned meshes sum the partial nodal sums across domains first (replaces Mat_Stress ADD +
                                // assembly, src/main.F90:286-289), then divide
        halo_sum(m, S, m->cpdim, 1, nn);
        hipLaunchKernelGGL(k_divide_by_valence, dim3(std::min<unsigned>(cdiv(nn * m->cpdim, 256), 8192)), dim3(256), 0, s, nn,
                           m->cpdim, S, m->valence_d.p);
    };
    // hex8 mesh with the tile plan of the one-pass product: the same tiles sum the nodal stresses in LDS (no element buffer)
    static const bool tile_env = !(std::getenv("FEM_AMD_STRESS_TILE") && std::atoi(std::getenv("FEM_AMD_STRESS_TILE")) == 0);
    if (tile_env && P->ebe_tile && ebe_tile_stress(m, *P->ebe_tile, U, S, multi ? nullptr : m->valence_d.p)) {
        if (multi) finish_multi();
        FEM_HIP(hipGetLastError());
        return;
    }
    std::vector<DBuf<double>> buf(m->groups.size());
    StressViews sv{};
    for (size_t gi = 0; gi < m->groups.size(); gi++) {
        auto& g = m->groups[gi];
        buf[gi].alloc((size_t)g.nel * g.npe * m->cpdim);
        sv.f[gi] = buf[gi].p; sv.nel[gi] = g.nel;
        dispatch_type(g.type, [&](auto PD, auto NPE, auto NIP, auto ELB) {
            constexpr int TPB = ELB() * (NIP() > NPE() ? NIP() : NPE());
            hipLaunchKernelGGL((k_stress<PD(), NPE(), NIP(), ELB(), true, MeshSrc<PD(), NPE()>, MeshU<PD(), NPE()>>),
                               dim3(cdiv(g.nel, ELB())), dim3(TPB), 0, s, g.type, g.nel,
                               MeshSrc<PD(), NPE()>{g.conn.p, m->coords.p}, MeshU<PD(), NPE()>{g.conn.p, U}, g.E.p,
                               g.nu.p, buf[gi].p);
        });
    }
    // single domain: the division by the valence rides on the gather; partitioned meshes sum the partial nodal
    // sums across domains first (replaces Mat_Stress ADD + assembly, src/main.F90:286-289)
    const double* val = multi ? nullptr : m->valence_d.p;
    if (m->cpdim == 6)
        hipLaunchKernelGGL(k_stress_gather<6>, dim3(cdiv(nn, 256)), dim3(256), 0, s, m->nnds, P->n2e_ptr.p, P->n2e.p, sv, val, S);
    else
        hipLaunchKernelGGL(k_stress_gather<3>, dim3(cdiv(nn, 256)), dim3(256), 0, s, m->nnds, P->n2e_ptr.p, P->n2e.p, sv, val, S);
    if (multi) finish_multi();
    FEM_HIP(hipGetLastError());
    FEM_HIP(hipStreamSynchronize(s));   // the element buffers go back to the pool at scope end
}

// EnforceBCForce (src/m_global.F90:439-455): INSERT PENALTY*val*dt/t on the constrained dofs of F
void enforce_bc_force_impl(fem_mesh m, int nbcs, const int* bcnode, const double* bcval, double dt, double t_total, double* F)
{
    if (nbcs <= 0) return;
    fem_ctx ctx = m->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int pd = m->pdim;
    auto& c = m->bc_force;
    const size_t nb = (size_t)nbcs * (1 + pd), nv = (size_t)nbcs * pd;
    const bool same = c.nm > 0 && c.bcnode.size() == nb && c.bcval.size() == nv && std::memcmp(c.bcnode.data(), bcnode, nb * sizeof(int)) == 0 &&
                      std::memcmp(c.bcval.data(), bcval, nv * sizeof(double)) == 0;
    if (!same) {
        // the reference INSERTs record after record (src/m_global.F90:439-455): where several records
        // constrain the same dof the last one's value stands -- merged here, one record per node
        std::vector<int> pos((size_t)m->nnds, -1), mb;
        std::vector<double> mv;
        for (int i = 0; i < nbcs; i++) {
            const int* rec = bcnode + (size_t)i * (1 + pd);
            FEM_CHECK(rec[0] >= 0 && rec[0] < m->nnds, 3, "fem_form_rhs: BC node out of range");
            int& p = pos[rec[0]];
            if (p < 0) {
                p = (int)(mb.size() / (1 + pd));
                mb.push_back(rec[0]);
                for (int d = 0; d < pd; d++) { mb.push_back(1); mv.push_back(0.0); }
            }
            for (int d = 0; d < pd; d++)
                if (rec[1 + d] == 0) { mb[(size_t)p * (1 + pd) + 1 + d] = 0; mv[(size_t)p * pd + d] = bcval[(size_t)i * pd + d]; }
        }
        c.nm = 0;
        c.d_bc.alloc(mb.size()); c.d_bc.upload(mb.data(), mb.size(), s);
        c.d_bcval.alloc(mv.size()); c.d_bcval.upload(mv.data(), mv.size(), s);
        FEM_HIP(hipStreamSynchronize(s));   // the merged host lists die at scope end
        c.bcnode.assign(bcnode, bcnode + nb); c.bcval.assign(bcval, bcval + nv);
        c.nm = (int)(mb.size() / (1 + pd));
    }
    hipLaunchKernelGGL(k_rhs_bc, dim3(cdiv(c.nm, 256)), dim3(256), 0, s, c.nm, pd, c.d_bc.p, c.d_bcval.p, dt, t_total, F);
    FEM_HIP(hipGetLastError());
}

void form_rhs_impl(fem_mesh m, int nbcs, const int* bcnode, const double* bcval, int nfrcs, const int* fnode,
                   const double* fval, int ntrcs, const int* telsd, const double* tval, double t_total, double t_init,
                   double dt, double* F)
{
    fem_ctx ctx = m->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int pd = m->pdim;
    const double t_end = t_init + dt;
    FEM_HIP(hipMemsetAsync(F, 0, sizeof(double) * (size_t)m->nnds * pd, s));
    DBuf<int> d_fnode, d_tt, d_tn, d_ts;
    DBuf<double> d_fval, d_tval;
    auto add_forces = [&] {
        if (nfrcs <= 0) return;
        for (int i = 0; i < nfrcs; i++)
            FEM_CHECK(fnode[i] >= 0 && fnode[i] < m->nnds, 3, "fem_form_rhs: force node out of range");
        d_fnode.alloc(nfrcs); d_fnode.upload(fnode, nfrcs, s);
        d_fval.alloc((size_t)nfrcs * (pd + 2)); d_fval.upload(fval, d_fval.n, s);
        hipLaunchKernelGGL(k_rhs_forces, dim3(cdiv(nfrcs, 256)), dim3(256), 0, s, nfrcs, pd, d_fnode.p, d_fval.p, t_init,
                           t_end, F);
    };
    auto add_tractions = [&] {
        if (ntrcs <= 0) return;
        std::vector<int> tt(ntrcs), tn((size_t)ntrcs * 8, 0), ts(ntrcs);
        for (int i = 0; i < ntrcs; i++) {
            const int el = telsd[2 * i];
            FEM_CHECK(el >= 0 && el < m->nels, 3, "fem_form_rhs: traction element out of range");
            const int t = m->h_eltype[el], n = m->h_eptr[el + 1] - m->h_eptr[el];
            FEM_CHECK(t >= FEM_TRI && t <= FEM_HEX, 3, "fem_form_rhs: traction on unsupported element type");
            const int nsides = (t == FEM_TRI) ? 3 : (t == FEM_HEX ? 6 : 4);
            FEM_CHECK(telsd[2 * i + 1] >= 1 && telsd[2 * i + 1] <= nsides, 3, "fem_form_rhs: bad side number");
            tt[i] = t; ts[i] = telsd[2 * i + 1];
            for (int a = 0; a < n; a++) tn[(size_t)i * 8 + a] = m->h_conn[m->h_eptr[el] + a];
        }
        d_tt.alloc(ntrcs); d_tt.upload(tt.data(), ntrcs, s);
        d_tn.alloc(tn.size()); d_tn.upload(tn.data(), tn.size(), s);
        d_ts.alloc(ntrcs); d_ts.upload(ts.data(), ntrcs, s);
        d_tval.alloc((size_t)ntrcs * (pd + 2)); d_tval.upload(tval, d_tval.n, s);
        hipLaunchKernelGGL(k_rhs_tractions, dim3(cdiv(ntrcs, 256)), dim3(256), 0, s, ntrcs, pd, d_tt.p, d_tn.p, d_ts.p,
                           d_tval.p, m->coords.p, t_init, t_end, dt, F);
        FEM_HIP(hipStreamSynchronize(s));  // host staging vectors die at scope end
    };
    if (m->halo && m->halo->comm->size > 1) {
        // multi-domain: tractions are per-element (additive) contributions -> halo sum;
        // nodal forces and BC values are applied in full by every domain holding the node
        add_tractions();
        halo_sum(m, F, pd, pd, 1);
        add_forces();
    } else {  // the reference's order (src/m_global.F90:400-426)
        add_forces();
        add_tractions();
    }
    enforce_bc_force_impl(m, nbcs, bcnode, bcval, dt, t_total, F);
    FEM_HIP(hipGetLastError());
    FEM_HIP(hipStreamSynchronize(s));
}
