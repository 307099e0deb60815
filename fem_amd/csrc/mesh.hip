// fem_amd/csrc/mesh.hip -- device-resident SoA mesh of one domain.
//
// Replaces the reference's AoS `type element` records with per-element heap
// arrays and duplicated coordinates (src/m_elems.F90:6-12, src/m_io.F90:102):
// connectivity conn[nel][npe] per element type, coords[nnds][pdim] once, and
// (E, nu) per element.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <sched.h>
#include <thread>

#include "objects.h"

#include <cstdlib>
#include <string>

using namespace fem;

namespace {
// f(first, last) over [0, n) on a few host threads (the hand-over of a mesh of millions of elements: checks and copies)
template <class F>
void host_ranges(long long n, F f)
{
    const int hc = (int)std::thread::hardware_concurrency();
    const int nt = (int)std::max<long long>(1, std::min<long long>(std::min(16, hc > 0 ? hc : 1), n / 65536));
    std::vector<std::thread> th;
    // A worker inherits the CPU mask of the calling thread: a pinning chosen by the user or the launcher (taskset, numactl, per-rank
    // binding under mpirun / torchrun) is respected.  FEM_AMD_HOST_AFFINITY=widen gives the workers every CPU instead -- for a
    // caller whose own thread an OpenMP runtime has pinned to one core (OMP_PROC_BIND; bench.py is such a case and sets it).
    static const bool widen = std::getenv("FEM_AMD_HOST_AFFINITY") && std::string(std::getenv("FEM_AMD_HOST_AFFINITY")) == "widen";
    for (int i = 1; i < nt; i++)
        th.emplace_back([=] {
            if (widen) {
                cpu_set_t all;
                CPU_ZERO(&all);
                for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &all);
                (void)sched_setaffinity(0, sizeof(all), &all);
            }
            f(n * i / nt, n * (i + 1) / nt);
        });
    f(0LL, n / nt);
    for (auto& t : th) t.join();
}
}  // namespace

namespace fem {

__global__ void __launch_bounds__(256) k_count_valence(const int* __restrict__ conn, size_t nent, int* __restrict__ cnt)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < nent; i += st) atomicAdd(&cnt[conn[i]], 1);
}

// E, nu per element from the material table (mat(el%mat,1:2), src/m_global.F90:254-255); element ids 0 .. nel-1
__global__ void __launch_bounds__(256) k_elem_materials(int nel, const int* __restrict__ mat_id, const double* __restrict__ mats,
                                                        double* __restrict__ E, double* __restrict__ nu, int* __restrict__ elem_id)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)nel) return;
    const int mt = mat_id[e];
    E[e] = mats[(size_t)mt * 5]; nu[e] = mats[(size_t)mt * 5 + 1]; elem_id[e] = (int)e;
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
    m->h_conn.resize((size_t)eptr[nels]);
    host_ranges((long long)eptr[nels], [&](long long a, long long b) { std::memcpy(m->h_conn.data() + a, conn + a, (size_t)(b - a) * sizeof(int)); });
    m->elem_group.assign(nels, -1);
    m->elem_pos.assign(nels, -1);

    // The common case -- every element of one solid type, in the caller's order -- needs no bucketing: the checks run on a
    // few host threads, the connectivity goes to the device as the caller holds it, E and nu are looked up there.
    const int t0 = nels > 0 ? eltype[0] : -1;
    bool uniform = nels > 0 && t0 >= FEM_TRI && t0 <= FEM_HEX && type_dim(t0) == pdim && (long long)eptr[nels] == (long long)nels * type_npe(t0);
    if (uniform) {
        std::atomic<int> bad{0};   // 1: not uniform after all (the general path reports), 2: a node id out of range
        const int npe0 = type_npe(t0);
        host_ranges(nels, [&](long long a, long long b) {
            int bd = 0;
            for (long long e = a; e < b && !bd; e++) {
                if (eltype[e] != t0 || eptr[e] != (int)(e * npe0) || mat_id[e] < 0 || mat_id[e] >= nmats) { bd = 1; break; }
                for (int k = 0; k < npe0; k++) {
                    const int v = conn[e * npe0 + k];
                    if (v < 0 || v >= nnds) bd = 2;
                }
                m->elem_group[e] = 0; m->elem_pos[e] = (int)e;
            }
            if (bd) bad.store(bd);
        });
        if (bad.load()) uniform = false;
    }
    if (uniform) {
        TypeGroup g;
        g.type = t0; g.npe = type_npe(t0); g.nip = type_nip(t0); g.nel = nels;
        g.conn.alloc((size_t)eptr[nels]); g.conn.upload(conn, (size_t)eptr[nels], s);
        g.elem_id.alloc((size_t)nels); g.E.alloc((size_t)nels); g.nu.alloc((size_t)nels);
        DBuf<int> dmat((size_t)nels);
        DBuf<double> dtab((size_t)nmats * 5);
        dmat.upload(mat_id, (size_t)nels, s); dtab.upload(mats, (size_t)nmats * 5, s);
        hipLaunchKernelGGL(k_elem_materials, dim3(cdiv((size_t)nels, 256)), dim3(256), 0, s, nels, dmat.p, dtab.p, g.E.p, g.nu.p, g.elem_id.p);
        FEM_HIP(hipStreamSynchronize(s));   // (dmat, dtab go out of scope)
        m->groups.push_back(std::move(g));
    }

    // validate + bucket by type
    std::vector<int> per_type[6];
    if (!uniform) { std::fill(m->elem_group.begin(), m->elem_group.end(), -1); std::fill(m->elem_pos.begin(), m->elem_pos.end(), -1); }
    for (int e = 0; e < nels && !uniform; e++) {
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
