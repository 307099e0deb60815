// fem_amd/csrc/coh.hip -- cohesive element, element level: the kinematics of the 4-node
// zero-thickness element (getCohValues / getCohRels / getCohGaps, src/m_elems.F90:658-720), the
// Xu-Needleman law (Seplaw_1_Tract / Seplaw_1_Stiff, src/m_seplaw.F90:15-94) and the element vectors and
// matrices the Newton callbacks add (applyTract_1 src/m_global.F90:622-682, applyStiff_1 :762-845), batched,
// one thread per element; and the same on the mesh's own cohesive elements for the Newton callbacks
// (CalcJacobian src/m_global.F90:127-147, CalcResidual :186-205; the SNES loop is snes.cpp).
// Reference behaviour is kept as written and said so: applyTract_1 ASSIGNS per integration point (the last
// point wins), applyStiff_1 scales only its second term by sig1*sig2*N*N*w*det.
#include "objects.h"

using namespace fem;

namespace fem {

struct CohGp {
    double urel[2], vrel[2], gap[2], vgap[2], tract[2], stiff[4];
};

__device__ __forceinline__ void coh_shape(int gp, double (&N)[4])
{
    const double eta = (gp == 0) ? -0.5773502691896260 : 0.5773502691896260;   // SamPtsCoh :607
    N[0] = 0.5 * (1 - eta); N[1] = 0.5 * (1 + eta); N[2] = N[1]; N[3] = N[0];   // ShapeFuncPrecompCoh :640-643
}

__device__ __forceinline__ void coh_point(const double* __restrict__ u8, const double* __restrict__ props, double dt,
                                          const double (&t)[2], const double (&n)[2], int gp, CohGp& o)
{
    double N[4];
    coh_shape(gp, N);
    o.urel[0] = o.urel[1] = 0.0;
    for (int n1 = 0; n1 < 2; n1++) {
        const int n2 = n1 + 2;
        for (int i = 0; i < 2; i++) o.urel[i] = o.urel[i] + N[n2] * u8[2 * n2 + i] - N[n1] * u8[2 * n1 + i];
    }
    o.vrel[0] = o.urel[0] / dt; o.vrel[1] = o.urel[1] / dt;
    o.gap[0] = n[0] * o.urel[0] + n[1] * o.urel[1];
    o.gap[1] = t[0] * o.urel[0] + t[1] * o.urel[1];
    o.vgap[0] = n[0] * o.vrel[0] + n[1] * o.vrel[1];
    o.vgap[1] = t[0] * o.vrel[0] + t[1] * o.vrel[1];
    const double sw = exp(1.0) * props[0] * props[1];
    const double dn = props[1], dtt = props[2], q = props[3], r = props[4], zeta = props[5];
    const double g1 = o.gap[0], g2 = o.gap[1];
    const double et = exp(-g2 * g2 / (dtt * dtt)), en = exp(-g1 / dn);
    double c1, c2;
    // Seplaw_1_Tract
    c1 = (1.0 - et);
    c1 = c1 * (1.0 - q) / (r - 1.0) * (r - g1 / dn);
    c2 = (g1 / dn) * et;
    o.tract[0] = (sw / dn) * en * (c2 + c1);
    c1 = q + (r - q) / (r - 1.0) * (g1 / dn);
    c1 = c1 * en * et;
    c1 = c1 * 2.0 * (dn / dtt) * (sw / dn);
    o.tract[1] = c1 * g2 / dtt;
    o.tract[0] = o.tract[0] + zeta * props[0] * o.vgap[0] / dn;
    // Seplaw_1_Stiff
    c1 = (1.0 - q) / (r - 1) * (1.0 - et);
    c1 = c1 * (r + 1.0 - g1 / dn);
    c1 = (1.0 - g1 / dn) * et - c1;
    double s11 = (sw / (dn * dn)) * en * c1;
    c1 = q + (g1 / dn) * (r - q) / (r - 1.0);
    c1 = c1 * en * et;
    c1 = 2.0 * (sw / (dtt * dtt)) * c1;
    const double s22 = c1 * (1.0 - 2.0 * (g2 * g2 / (dtt * dtt)));
    c1 = -g1 / dn + (1.0 - q) / (r - 1.0) * (r - g1 / dn);
    c1 = c1 * en * et;
    c1 = 2.0 * (sw / (dtt * dn)) * c1;
    const double s12 = (g2 / dtt) * c1;
    s11 = s11 + zeta * props[0] / dn / dt;
    o.stiff[0] = s11; o.stiff[1] = s12; o.stiff[2] = s12; o.stiff[3] = s22;   // (1,1) (2,1) (1,2) (2,2)
}

// element vector (applyTract_1) and matrix (applyStiff_1) of one cohesive element: ec = ecoords(4,2) column-major,
// u = the eight numbers getCohRels reads, pr = the six law parameters
__device__ __forceinline__ void coh_element(const double* __restrict__ ec, const double* __restrict__ u,
                                            const double* __restrict__ pr, double dt, double (&res)[8], double (&K)[64],
                                            double* __restrict__ gp_rec)
{
    double t[2], n[2];
    t[0] = 0.5 * (ec[1] - ec[0] + ec[2] - ec[3]);
    t[1] = 0.5 * (ec[5] - ec[4] + ec[6] - ec[7]);
    const double mag = sqrt(t[0] * t[0] + t[1] * t[1]);
    t[0] /= mag; t[1] /= mag;
    const double det = mag * 0.5;
    n[0] = -t[1]; n[1] = t[0];
    for (int k = 0; k < 8; k++) res[k] = 0.0;
    for (int k = 0; k < 64; k++) K[k] = 0.0;
    for (int gp = 0; gp < 2; gp++) {
        CohGp o;
        coh_point(u, pr, dt, t, n, gp, o);
        if (gp_rec) {
            double* g = gp_rec + (size_t)gp * 14;
            g[0] = o.urel[0]; g[1] = o.urel[1]; g[2] = o.vrel[0]; g[3] = o.vrel[1]; g[4] = o.gap[0]; g[5] = o.gap[1];
            g[6] = o.vgap[0]; g[7] = o.vgap[1]; g[8] = o.tract[0]; g[9] = o.tract[1];
            g[10] = o.stiff[0]; g[11] = o.stiff[1]; g[12] = o.stiff[2]; g[13] = o.stiff[3];
        }
        double N[4];
        coh_shape(gp, N);
        double sig = 1.0;
        for (int node = 0; node < 4; node++) {
            if (node + 1 > 2) sig = -1.0;
            for (int d = 0; d < 2; d++)   // "=": src/m_global.F90:668
                res[node * 2 + d] = sig * N[node] * (o.tract[0] * n[d] + o.tract[1] * t[d]) * 1.0 * det;
        }
        double s1 = 1.0;
        for (int n1 = 0; n1 < 4; n1++) {
            if (n1 + 1 > 2) s1 = -1.0;
            double s2 = 1.0;
            for (int n2 = 0; n2 < 4; n2++) {
                if (n2 + 1 > 2) s2 = -1.0;
                for (int d1 = 0; d1 < 2; d1++)
                    for (int d2 = 0; d2 < 2; d2++) {   // src/m_global.F90:829-831, parentheses as in the reference
                        double& r = K[(n1 * 2 + d1) + 8 * (n2 * 2 + d2)];
                        r = r + ((o.stiff[0] * n[d1] + o.stiff[1] * t[d1]) * n[d2]) +
                            ((o.stiff[2] * n[d1] + o.stiff[3] * t[d1]) * t[d2]) * s1 * s2 * N[n1] * N[n2] * 1.0 * det;
                    }
            }
        }
    }
}

__global__ void __launch_bounds__(128) k_coh_apply_1(int ncoh, const double* __restrict__ ecoords,
                                                     const double* __restrict__ u8, const double* __restrict__ props,
                                                     double dt, double* __restrict__ tract, double* __restrict__ stiff,
                                                     double* __restrict__ gp_out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ncoh) return;
    double res[8], K[64];
    coh_element(ecoords + (size_t)e * 8, u8 + (size_t)e * 8, props + (size_t)e * 6, dt, res, K,
                gp_out ? gp_out + (size_t)e * 28 : (double*)nullptr);
    if (tract) for (int k = 0; k < 8; k++) tract[(size_t)e * 8 + k] = res[k];
    if (stiff) for (int k = 0; k < 64; k++) stiff[(size_t)e * 64 + k] = K[k];
}

struct RelsNodes {
    int v[4];
};
// The mesh's cohesive elements inside the Newton callbacks.  STIFF = false: R(node j) += nodeTraction (CalcResidual,
// src/m_global.F90:200-203: ApplyNodalForce(Residual, ..., .true.)); STIFF = true: Jacobian += appliedStiff, 64
// MatSetValue(ADD_VALUES) per element (:140-146).  getCohRels reads local_u(2(k-1)+i), k = 1..4 -- four fixed nodes of
// the rank (rels_mode 0: `rn`, by default the first four), not the element's (src/m_elems.F90:697-704); rels_mode 1
// hands it the element's own nodes.  One launch per colour of the cohesive elements (no two elements of a colour share a
// node: fem_mesh_set_cohesive): plain adds, the sums into a node in the fixed order of the colours -- bitwise reproducible.
template <bool STIFF>
__global__ void __launch_bounds__(128) k_coh_mesh(int first, int ncoh, const int* __restrict__ conn, const double* __restrict__ props,
                                                  const double* __restrict__ coords, const double* __restrict__ utot, double dt,
                                                  int rels_mode, RelsNodes rn, double* __restrict__ R,
                                                  const int* __restrict__ rowlen, const int* __restrict__ boff,
                                                  const int* __restrict__ bcol, double* __restrict__ val)
{
    // STIFF: sixteen threads per element, one per (row node, column node) pair -- the element's numbers are cheap, the look-up of a block's
    // place in its row is a chain of dependent loads (one thread doing all sixteen: 20 us per launch, round 6)
    const int tid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int e = first + (STIFF ? tid >> 4 : tid);
    if (e >= first + ncoh) return;
    const int* nd = conn + (size_t)e * 4;
    double ec[8], u[8], res[8], K[64];
    for (int a = 0; a < 4; a++) {
        ec[a] = coords[(size_t)nd[a] * 2]; ec[a + 4] = coords[(size_t)nd[a] * 2 + 1];
        const int v = rels_mode ? nd[a] : rn.v[a];
        u[2 * a] = utot[(size_t)v * 2]; u[2 * a + 1] = utot[(size_t)v * 2 + 1];
    }
    coh_element(ec, u, props + (size_t)e * 6, dt, res, K, nullptr);
    if constexpr (!STIFF) {
        for (int j = 0; j < 4; j++)
            for (int d = 0; d < 2; d++) R[(size_t)nd[j] * 2 + d] += res[2 * j + d];
    } else {
        const int n1 = (tid >> 2) & 3, n2 = tid & 3;
        double k4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (a == n1 && b == n2) {
#pragma unroll
                    for (int d1 = 0; d1 < 2; d1++)
#pragma unroll
                        for (int d2 = 0; d2 < 2; d2++) k4[d1 * 2 + d2] = K[(a * 2 + d1) + 8 * (b * 2 + d2)];
                }
        const int row = nd[n1], col = nd[n2];
        const int o0 = boff[row >> 6], lane = row & 63, len = rowlen[row];
        int lo = 0, hi = len;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (bcol[(size_t)(o0 + mid) * 64 + lane] < col) lo = mid + 1; else hi = mid;
        }
        double* blk = val + (size_t)(o0 + lo) * 4 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; q++) blk[(size_t)q * 64] += k4[q];
    }
}

}  // namespace fem

void coh_apply_1_impl(fem_ctx ctx, int ncoh, const double* ecoords, const double* u8, const double* props, double dt,
                      double* tract, double* stiff, double* gp)
{
    if (ncoh <= 0) return;
    FEM_CHECK(ecoords && u8 && props, 1, "fem_coh_apply_1: null input");
    FEM_CHECK(dt > 0.0, 2, "fem_coh_apply_1: dt must be positive");
    ctx->bind();
    hipStream_t s = ctx->stream;
    DBuf<double> dE((size_t)ncoh * 8), dU((size_t)ncoh * 8), dP((size_t)ncoh * 6), dT((size_t)ncoh * 8),
        dK((size_t)ncoh * 64), dG((size_t)ncoh * 28);
    dE.upload(ecoords, dE.n, s); dU.upload(u8, dU.n, s); dP.upload(props, dP.n, s);
    hipLaunchKernelGGL(k_coh_apply_1, dim3(cdiv((size_t)ncoh, 128)), dim3(128), 0, s, ncoh, dE.p, dU.p, dP.p, dt, dT.p, dK.p,
                       gp ? dG.p : (double*)nullptr);
    FEM_HIP(hipGetLastError());
    if (tract) dT.download(tract, dT.n, s);
    if (stiff) dK.download(stiff, dK.n, s);
    if (gp) dG.download(gp, dG.n, s);
    FEM_HIP(hipStreamSynchronize(s));
}

// fem_mesh_set_cohesive: the cohesive elements of the deck (type coh in the element list the mesh was created from)
void mesh_set_cohesive_impl(fem_mesh m, const int* nlmat, int ncohmats, const double* props)
{
    FEM_CHECK(m && nlmat, 1, "fem_mesh_set_cohesive: null argument");
    FEM_CHECK(!m->pattern, 5, "fem_mesh_set_cohesive: call it before the first fem_mat_create_aij (the sparsity holds the "
                              "cohesive node pairs)");
    FEM_CHECK(ncohmats >= 0 && (ncohmats == 0 || props), 2, "fem_mesh_set_cohesive: bad material table");
    m->ctx->bind();
    hipStream_t s = m->ctx->stream;
    std::vector<int> conn;
    std::vector<double> pr;
    for (int e = 0; e < m->nels; e++) {
        if (nlmat[e] == 0) continue;   // el%nlMat == 0: linear element (src/m_global.F90:130)
        FEM_CHECK(m->h_eltype[e] == FEM_COH, 3, "fem_mesh_set_cohesive: nlMat set on an element that is not of type coh");
        FEM_CHECK(m->pdim == 2, 3, "fem_mesh_set_cohesive: cohesive elements are 2-D (4 nodes)");
        FEM_CHECK(nlmat[e] >= 1 && nlmat[e] <= ncohmats, 3, "fem_mesh_set_cohesive: cohesive material out of range");
        for (int a = 0; a < 4; a++) {
            const int v = m->h_conn[m->h_eptr[e] + a];
            FEM_CHECK(v >= 0 && v < m->nnds, 3, "fem_mesh_set_cohesive: node id out of range");
            conn.push_back(v);
        }
        for (int q = 0; q < 6; q++) pr.push_back(props[(size_t)(nlmat[e] - 1) * 6 + q]);
    }
    m->ncoh = (int)(conn.size() / 4);
    {   // colours: greedily, in the order of the deck, the lowest colour none of the element's nodes carries yet; the elements are
        // then kept colour by colour (deck order inside a colour) and every colour is one launch of the element kernels
        const int nc = m->ncoh;
        std::vector<unsigned long long> used((size_t)m->nnds, 0ULL);
        std::vector<int> col((size_t)nc, 0);
        int ncol = 0;
        for (int e = 0; e < nc; e++) {
            unsigned long long u = 0;
            for (int a = 0; a < 4; a++) u |= used[(size_t)conn[(size_t)4 * e + a]];
            int c = 0;
            while (c < 64 && ((u >> c) & 1ULL)) c++;
            FEM_CHECK(c < 64, 3, "fem_mesh_set_cohesive: a node sits in more than 64 cohesive elements");
            col[(size_t)e] = c; ncol = std::max(ncol, c + 1);
            for (int a = 0; a < 4; a++) used[(size_t)conn[(size_t)4 * e + a]] |= 1ULL << c;
        }
        std::vector<int> conn2; std::vector<double> pr2;
        m->coh_colour_ptr.assign(1, 0);
        for (int c = 0; c < ncol; c++) {
            for (int e = 0; e < nc; e++) {
                if (col[(size_t)e] != c) continue;
                conn2.insert(conn2.end(), conn.begin() + (size_t)4 * e, conn.begin() + (size_t)4 * e + 4);
                pr2.insert(pr2.end(), pr.begin() + (size_t)6 * e, pr.begin() + (size_t)6 * e + 6);
            }
            m->coh_colour_ptr.push_back((int)(conn2.size() / 4));
        }
        conn.swap(conn2); pr.swap(pr2);
    }
    m->h_coh_conn = conn;
    if (m->ncoh) {
        m->coh_conn.alloc(conn.size()); m->coh_conn.upload(conn.data(), conn.size(), s);
        m->coh_props.alloc(pr.size()); m->coh_props.upload(pr.data(), pr.size(), s);
        FEM_HIP(hipStreamSynchronize(s));
    }
}

static RelsNodes rels_of(fem_mesh m, int rels_mode, const int* rels_nodes)
{
    RelsNodes rn{{0, 1, 2, 3}};   // "the first four local nodes of the rank"
    if (rels_nodes) for (int a = 0; a < 4; a++) rn.v[a] = rels_nodes[a];
    if (!rels_mode)
        for (int a = 0; a < 4; a++) FEM_CHECK(rn.v[a] >= 0 && rn.v[a] < m->nnds, 3, "cohesive: rels node out of range");
    return rn;
}

// CalcJacobian's element loop (src/m_global.F90:127-147) on Jacobian J
void mat_add_cohesive_impl(fem_mat J, const double* utot, double dt, int rels_mode, const int* rels_nodes)
{
    fem_mesh m = J->mesh;
    if (m->ncoh == 0) return;
    FEM_CHECK(dt > 0.0, 2, "fem_mat_add_cohesive: dt must be positive");
    FEM_CHECK(!J->mult_ebe, 6, "fem_mat_add_cohesive: the element-by-element MatMult recomputes the solid elements only; "
                                "a Mat that carries cohesive tangents needs -mat_mult_type assembled");
    J->ctx->bind();
    hipStream_t s = J->ctx->stream;
    const RelsNodes rn = rels_of(m, rels_mode, rels_nodes);
    double* jval = J->values();
    for (size_t c = 0; c + 1 < m->coh_colour_ptr.size(); c++) {
        const int first = m->coh_colour_ptr[c], cnt = m->coh_colour_ptr[c + 1] - first;
        hipLaunchKernelGGL(k_coh_mesh<true>, dim3(cdiv((size_t)cnt * 16, 128)), dim3(128), 0, s, first, cnt, m->coh_conn.p, m->coh_props.p,
                           m->coords.p, utot, dt, rels_mode, rn, (double*)nullptr, J->rowlen.p, J->boff.p, J->bcol.p, jval);
    }
    J->unsymmetric = true;
    J->bc_saved = false;   // ApplyKBC comes next and sees a new diagonal
    FEM_HIP(hipGetLastError());
}

// CalcResidual's element loop (src/m_global.F90:186-205) on R
void vec_add_cohesive_traction_impl(fem_mesh m, const double* utot, double dt, int rels_mode, const int* rels_nodes, double* R)
{
    if (m->ncoh == 0) return;
    FEM_CHECK(dt > 0.0, 2, "fem_vec_add_cohesive_traction: dt must be positive");
    m->ctx->bind();
    hipStream_t s = m->ctx->stream;
    const RelsNodes rn = rels_of(m, rels_mode, rels_nodes);
    for (size_t c = 0; c + 1 < m->coh_colour_ptr.size(); c++) {
        const int first = m->coh_colour_ptr[c], cnt = m->coh_colour_ptr[c + 1] - first;
        hipLaunchKernelGGL(k_coh_mesh<false>, dim3(cdiv((size_t)cnt, 128)), dim3(128), 0, s, first, cnt, m->coh_conn.p, m->coh_props.p,
                           m->coords.p, utot, dt, rels_mode, rn, R, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr,
                           (double*)nullptr);
    }
    FEM_HIP(hipGetLastError());
}
