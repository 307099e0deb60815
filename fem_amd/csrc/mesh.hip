// fem_amd/csrc/mesh.hip -- device-resident SoA mesh of one domain.
//
// Replaces the reference's AoS `type element` records with per-element heap
// arrays and duplicated coordinates (src/m_elems.F90:6-12, src/m_io.F90:102):
// connectivity conn[nel][npe] per element type, coords[nnds][pdim] once, and
// (E, nu) per element.
#include "objects.h"

using namespace fem;

namespace fem {

__global__ void __launch_bounds__(256) k_count_valence(const int* __restrict__ conn, size_t nent, int* __restrict__ cnt)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < nent; i += st) atomicAdd(&cnt[conn[i]], 1);
}

__global__ void __launch_bounds__(256) k_i32_to_f64(const int* __restrict__ a, double* __restrict__ b, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = (double)a[i];
}

}  // namespace fem

fem_mesh mesh_create_impl(fem_ctx ctx, int pdim, int nels, int nnds, const int* eltype, const int* eptr,
                          const int* conn, const int* mat_id, const double* coords, int nmats, const double* mats)
{
    FEM_CHECK(ctx, 1, "fem_mesh_create: null context");
    FEM_CHECK(pdim == 2 || pdim == 3, 2, "fem_mesh_create: pdim must be 2 or 3");
    FEM_CHECK(nels >= 0 && nnds > 0, 2, "fem_mesh_create: bad sizes");
    ctx->bind();
    hipStream_t s = ctx->stream;
    std::unique_ptr<fem_mesh_s> m(new fem_mesh_s);
    m->ctx = ctx; m->pdim = pdim; m->nels = nels; m->nnds = nnds; m->cpdim = (pdim == 2) ? 3 : 6;
    m->h_eltype.assign(eltype, eltype + nels);
    m->h_eptr.assign(eptr, eptr + nels + 1);
    m->h_conn.assign(conn, conn + eptr[nels]);
    m->elem_group.assign(nels, -1);
    m->elem_pos.assign(nels, -1);

    // validate + bucket by type
    std::vector<int> per_type[6];
    for (int e = 0; e < nels; e++) {
        const int t = eltype[e];
        FEM_CHECK(t >= FEM_TRI && t <= FEM_COH, 3, "fem_mesh_create: unknown element type " + std::to_string(t));
        const int n = eptr[e + 1] - eptr[e];
        FEM_CHECK(n == type_npe(t), 3, "fem_mesh_create: element " + std::to_string(e) + " has wrong node count");
        if (t == FEM_COH) continue;  // cohesive elements are not part of the linear hot path
        FEM_CHECK(type_dim(t) == pdim, 3, "fem_mesh_create: element type does not match pdim");
        FEM_CHECK(mat_id[e] >= 0 && mat_id[e] < nmats, 3, "fem_mesh_create: material id out of range");
        for (int a = eptr[e]; a < eptr[e + 1]; a++)
            FEM_CHECK(conn[a] >= 0 && conn[a] < nnds, 3, "fem_mesh_create: node id out of range");
        per_type[t].push_back(e);
    }
    for (int t = FEM_TRI; t <= FEM_HEX; t++) {
        if (per_type[t].empty()) continue;
        TypeGroup g;
        g.type = t; g.npe = type_npe(t); g.nip = type_nip(t); g.nel = (int)per_type[t].size();
        std::vector<int> hc((size_t)g.nel * g.npe), hid(g.nel);
        std::vector<double> hE(g.nel), hnu(g.nel);
        for (int k = 0; k < g.nel; k++) {
            const int e = per_type[t][k];
            hid[k] = e;
            m->elem_group[e] = (int)m->groups.size();
            m->elem_pos[e] = k;
            for (int a = 0; a < g.npe; a++) hc[(size_t)k * g.npe + a] = conn[eptr[e] + a];
            hE[k] = mats[(size_t)mat_id[e] * 5 + 0];
            hnu[k] = mats[(size_t)mat_id[e] * 5 + 1];
        }
        g.conn.alloc(hc.size()); g.conn.upload(hc.data(), hc.size(), s);
        g.elem_id.alloc(hid.size()); g.elem_id.upload(hid.data(), hid.size(), s);
        g.E.alloc(hE.size()); g.E.upload(hE.data(), hE.size(), s);
        g.nu.alloc(hnu.size()); g.nu.upload(hnu.data(), hnu.size(), s);
        FEM_HIP(hipStreamSynchronize(s));  // host staging vectors die at scope end
        m->groups.push_back(std::move(g));
    }
    m->coords.alloc((size_t)nnds * pdim);
    m->coords.upload(coords, (size_t)nnds * pdim, s);
    m->valence.alloc(nnds);
    m->valence.zero(s);
    for (auto& g : m->groups) {
        const size_t nent = (size_t)g.nel * g.npe;
        hipLaunchKernelGGL(k_count_valence, dim3(std::min<unsigned>(cdiv(nent, 256), 4096)), dim3(256), 0, s,
                           g.conn.p, nent, m->valence.p);
    }
    m->valence_d.alloc(nnds);
    hipLaunchKernelGGL(k_i32_to_f64, dim3(std::min<unsigned>(cdiv(nnds, 256), 4096)), dim3(256), 0, s, m->valence.p,
                       m->valence_d.p, (size_t)nnds);
    FEM_HIP(hipGetLastError());
    FEM_HIP(hipStreamSynchronize(s));
    return m.release();
}
