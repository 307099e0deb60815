// fem_amd/csrc/mat.hip -- Mat_K: symbolic phase (sparsity from connectivity, on
// the device), sliced block-ELL storage, MatMult, penalty BC, diagonal, export.
//
// Storage (DESIGN.md "Data layout"): block rows = nodes, block size bs = pdim.
// Rows are cut into slices of 64 = one wavefront; inside a slice everything is
// stored lane-fastest so that every load/store/atomic of a wavefront touches
// one contiguous 512-byte (fp64) or 256-byte (int32) segment:
//     bcol[(boff[s] + j) * 64 + lane]
//     val [((boff[s] + j) * bs*bs + r*bs + c) * 64 + lane]
// Measured on MI355X (tools/spmv_bench.hip, 150^3 nodes): this layout streams at
// 5.5 TB/s (96 % of the box's 5.75 TB/s read ceiling) where scalar CSR reaches
// 3.7 TB/s and row-per-wave BSR 2.9 TB/s.
#include <algorithm>
#include <map>

#include "comm.h"
#include "objects.h"
#include "scan.h"

using namespace fem;

namespace fem {

constexpr int MAXNB = 96;  // max distinct neighbour nodes of one node (27 for hex8 grids)

struct GroupView {
    const int* conn;
    int npe, nel;
};
struct GroupViews {
    GroupView g[4];
    int ng;
};

// ---- symbolic kernels (integer work) -------------------------------------------

__global__ void __launch_bounds__(256) k_fill_n2e(GroupView gv, int gidx, const int* __restrict__ n2e_ptr,
                                                  int* __restrict__ cursor, unsigned* __restrict__ n2e)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nent = (size_t)gv.nel * gv.npe, st = (size_t)gridDim.x * blockDim.x;
    for (; i < nent; i += st) {
        const int v = gv.conn[i];
        const int pos = atomicAdd(&cursor[v], 1);
        n2e[n2e_ptr[v] + pos] = ((unsigned)gidx << 30) | ((unsigned)(i / gv.npe) << 3) | (unsigned)(i % gv.npe);
    }
}

// One thread per node: (1) ascending codes inside the node's incidence list (k_fill_n2e fills it with
// an atomic cursor); (2) the largest valence; (3) the row-owner assembly gives every incidence of a row
// its own lane, `lpr` lanes per row and pass, and lets the lanes of a pass add local column b in the same
// instruction: where two incidences of one pass hold the same node at the same local index b (never on
// structured numberings), the slice is flagged and its lanes take turns instead.
__global__ void __launch_bounds__(256) k_sort_and_flag(int nn, const int* __restrict__ ptr, unsigned* __restrict__ lst,
                                                       GroupViews gvs, int lpr, unsigned char* __restrict__ slice_serial,
                                                       int* __restrict__ max_valence)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    const int p0 = ptr[v], p1 = ptr[v + 1];
    for (int i = p0 + 1; i < p1; i++) {
        const unsigned c = lst[i];
        int j = i - 1;
        while (j >= p0 && lst[j] > c) { lst[j + 1] = lst[j]; j--; }
        lst[j + 1] = c;
    }
    atomicMax(max_valence, p1 - p0);
    bool clash = false;
    for (int q0 = p0; q0 < p1 && !clash; q0 += lpr) {
        const int q1 = min(q0 + lpr, p1);
        for (int i = q0; i < q1 && !clash; i++) {
            const GroupView gi = gvs.g[lst[i] >> 30];
            const int* ci = gi.conn + (size_t)((lst[i] & 0x3fffffffu) >> 3) * gi.npe;
            for (int j = i + 1; j < q1 && !clash; j++) {
                const GroupView gj = gvs.g[lst[j] >> 30];
                const int* cj = gj.conn + (size_t)((lst[j] & 0x3fffffffu) >> 3) * gj.npe;
                const int nb = min(gi.npe, gj.npe);
                for (int b = 0; b < nb; b++) clash = clash || (ci[b] == cj[b]);
            }
        }
    }
    if (clash) slice_serial[v >> 6] = 1;
}

// One thread per node: sorted unique list of the nodes of all incident elements.
// FILL = false: write the list length.  FILL = true: write the slice-ELL column ids.
template <bool FILL>
__global__ void __launch_bounds__(256) k_node_neighbours(int nn, int nrows_padded, const int* __restrict__ n2e_ptr,
                                                         const unsigned* __restrict__ n2e, GroupViews gvs,
                                                         int* __restrict__ rowlen, const int* __restrict__ boff,
                                                         int* __restrict__ bcol, int* __restrict__ err,
                                                         const int* __restrict__ xptr, const int* __restrict__ xadj)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nrows_padded) return;
    int list[MAXNB];
    int len = 0;
    auto insert = [&](int u) {
        int lo = 0, hi = len;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (list[mid] < u) lo = mid + 1; else hi = mid;
        }
        if (lo < len && list[lo] == u) return;
        if (len == MAXNB) { atomicExch(err, 1); return; }
        for (int q = len; q > lo; q--) list[q] = list[q - 1];
        list[lo] = u;
        len++;
    };
    if (v < nn) {
        for (int k = n2e_ptr[v]; k < n2e_ptr[v + 1]; k++) {
            const unsigned code = n2e[k];
            const GroupView gv = gvs.g[code >> 30];
            const int* c = gv.conn + (size_t)((code & 0x3fffffffu) >> 3) * gv.npe;
            for (int a = 0; a < gv.npe; a++) insert(c[a]);
        }
        if (xptr)   // node pairs of the cohesive elements (they are not in the incidence lists)
            for (int k = xptr[v]; k < xptr[v + 1]; k++) insert(xadj[k]);
        if (len == 0) { list[0] = v; len = 1; }  // isolated node: keep a diagonal block
    }
    if (!FILL) {
        if (v < nn) rowlen[v] = len;
    } else {
        const int s = v >> 6, lane = v & 63;
        const int o0 = boff[s], W = boff[s + 1] - o0;
        const int padcol = (v < nn) ? v : nn - 1;
        for (int j = 0; j < W; j++) bcol[(size_t)(o0 + j) * 64 + lane] = (j < len) ? list[j] : padcol;
    }
}

__global__ void __launch_bounds__(256) k_slice_width(int nn, int nslices, const int* __restrict__ rowlen,
                                                     int* __restrict__ width)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslices) return;
    int w = 0;
    for (int l = 0; l < 64; l++) {
        const int v = s * 64 + l;
        if (v < nn) w = max(w, rowlen[v]);
    }
    width[s] = w;
}

// slot[e][a*npe+b] = position of node b in the (sorted) block row of node a
__global__ void __launch_bounds__(256) k_slots(GroupView gv, const int* __restrict__ rowlen,
                                               const int* __restrict__ boff, const int* __restrict__ bcol,
                                               uint16_t* __restrict__ slot)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pp = gv.npe * gv.npe;
    const size_t tot = (size_t)gv.nel * pp, st = (size_t)gridDim.x * blockDim.x;
    for (; i < tot; i += st) {
        const size_t e = i / pp;
        const int ab = (int)(i - e * pp), a = ab / gv.npe, b = ab - a * gv.npe;
        const int row = gv.conn[e * gv.npe + a], u = gv.conn[e * gv.npe + b];
        const int* bc = bcol + (size_t)boff[row >> 6] * 64 + (row & 63);
        int lo = 0, hi = rowlen[row];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (bc[(size_t)mid * 64] < u) lo = mid + 1; else hi = mid;
        }
        slot[i] = (uint16_t)lo;
    }
}

// ---- MatMult ---------------------------------------------------------------------
// One wavefront per slice, lane = block row.  Every wave-level load of `val`
// and `bcol` is one contiguous segment; x is gathered (neighbouring rows read
// neighbouring x blocks, served by L2 / Infinity Cache).  Optionally fuses the
// CG scalar p.(Kp) as one partial per workgroup.
// STREAM: block columns and values with the streaming hint (global_load ... nt) -- a matrix larger than the 256 MB Infinity Cache
// is read once per product and nothing of it is found again, the caches are for x (config 3, 6.5 GB: 1.168 -> 1.093 ms);
// a matrix that fits (config 2, 180 MB) is found there by the next product and keeps the plain loads (0.0323 against 0.0348 ms).
template <bool STREAM, class T>
__device__ __forceinline__ T ld_mat(const T* p) { return STREAM ? __builtin_nontemporal_load(p) : *p; }
template <int BS, bool STREAM>
__global__ void __launch_bounds__(256) k_spmv_sell(int nslices, int nrows, const int* __restrict__ boff,
                                                   const int* __restrict__ bcol, const double* __restrict__ val,
                                                   const double* __restrict__ x, double* __restrict__ y,
                                                   double* __restrict__ partial, const int* __restrict__ done)
{
    if (done && *done) return;
    constexpr int BB = BS * BS, U = 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + wave;
    double dot = 0.0;
    if (s < nslices) {
        const int o0 = boff[s], W = boff[s + 1] - o0;
        const int* bc = bcol + (size_t)o0 * 64 + lane;
        const double* v = val + (size_t)o0 * BB * 64 + lane;
        double acc[BS];
#pragma unroll
        for (int r = 0; r < BS; r++) acc[r] = 0.0;
        int j = 0;
        for (; j + U <= W; j += U) {
            int c[U];
            double a[U][BB], xx[U][BS];
#pragma unroll
            for (int u = 0; u < U; u++) c[u] = ld_mat<STREAM>(bc + (size_t)(j + u) * 64);
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int e = 0; e < BB; e++) a[u][e] = ld_mat<STREAM>(v + ((size_t)(j + u) * BB + e) * 64);
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int q = 0; q < BS; q++) xx[u][q] = x[(size_t)BS * c[u] + q];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int r = 0; r < BS; r++)
#pragma unroll
                    for (int q = 0; q < BS; q++) acc[r] += a[u][r * BS + q] * xx[u][q];
        }
        for (; j < W; j++) {
            const int c = ld_mat<STREAM>(bc + (size_t)j * 64);
            double xx[BS];
#pragma unroll
            for (int q = 0; q < BS; q++) xx[q] = x[(size_t)BS * c + q];
#pragma unroll
            for (int r = 0; r < BS; r++)
#pragma unroll
                for (int q = 0; q < BS; q++) acc[r] += ld_mat<STREAM>(v + ((size_t)j * BB + r * BS + q) * 64) * xx[q];
        }
        const int row = s * 64 + lane;
        if (row < nrows) {
#pragma unroll
            for (int r = 0; r < BS; r++) y[(size_t)BS * row + r] = acc[r];
            if (partial) {
#pragma unroll
                for (int r = 0; r < BS; r++) dot += acc[r] * x[(size_t)BS * row + r];
            }
        }
    }
    if (partial) {
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o);
        __shared__ double red[4];
        if (lane == 0) red[wave] = dot;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

template <int BS>
__global__ void __launch_bounds__(256) k_get_diag(int nrows, const int* __restrict__ rowlen,
                                                  const int* __restrict__ boff, const int* __restrict__ bcol,
                                                  const double* __restrict__ val, double* __restrict__ d)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    const int o0 = boff[row >> 6], lane = row & 63, len = rowlen[row];
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bcol[(size_t)(o0 + mid) * 64 + lane] < row) lo = mid + 1; else hi = mid;
    }
    for (int r = 0; r < BS; r++)
        d[(size_t)BS * row + r] = (lo < len && bcol[(size_t)(o0 + lo) * 64 + lane] == row)
                                      ? val[((size_t)(o0 + lo) * BS * BS + r * BS + r) * 64 + lane] : 0.0;
}

// ApplyKBC / UpdateKMatBcs (src/m_global.F90:264-299): MatSetValue(j,j,PENALTY,INSERT).
// Multi-domain: only the owning domain holds the penalty, the other sharers
// hold 0, so that the halo-summed operator has exactly PENALTY on the diagonal.
template <int BS>
__global__ void __launch_bounds__(256) k_apply_kbc(int nbcs, const int* __restrict__ bcnode,
                                                   const int* __restrict__ rowlen, const int* __restrict__ boff,
                                                   const int* __restrict__ bcol, double* __restrict__ val,
                                                   const unsigned char* __restrict__ owned, double penalty,
                                                   int* __restrict__ bc_dof, double* __restrict__ bc_shift,
                                                   double* __restrict__ bc_diag, int* __restrict__ bc_node)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbcs) return;
    const int row = bcnode[(size_t)i * (1 + BS)];
    const int o0 = boff[row >> 6], lane = row & 63, len = rowlen[row];
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bcol[(size_t)(o0 + mid) * 64 + lane] < row) lo = mid + 1; else hi = mid;
    }
    const double pv = (!owned || owned[row]) ? penalty : 0.0;
    for (int r = 0; r < BS; r++) {
        const bool fixed = bcnode[(size_t)i * (1 + BS) + 1 + r] == 0;
        double* d = &val[((size_t)(o0 + lo) * BS * BS + r * BS + r) * 64 + lane];
        if (bc_dof) {   // first application after an assembly: remember what the INSERT replaces
            bc_dof[(size_t)i * BS + r] = fixed ? row * BS + r : -1;
            bc_shift[(size_t)i * BS + r] = fixed ? pv - *d : 0.0;
            bc_diag[(size_t)i * BS + r] = *d;
            if (r == 0) bc_node[row] = i * BS;
        }
        if (fixed) *d = pv;
    }
}

// Undo of a previous ApplyKBC (a different constraint list arrives without a fresh assembly): put the
// assembled diagonal back on the dofs the old list had replaced.
template <int BS>
__global__ void __launch_bounds__(256) k_restore_kbc(int nbcs, const int* __restrict__ bcnode,
                                                     const int* __restrict__ rowlen, const int* __restrict__ boff,
                                                     const int* __restrict__ bcol, double* __restrict__ val,
                                                     const double* __restrict__ bc_diag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbcs) return;
    const int row = bcnode[(size_t)i * (1 + BS)];
    const int o0 = boff[row >> 6], lane = row & 63, len = rowlen[row];
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bcol[(size_t)(o0 + mid) * 64 + lane] < row) lo = mid + 1; else hi = mid;
    }
    for (int r = 0; r < BS; r++)
        if (bcnode[(size_t)i * (1 + BS) + 1 + r] == 0)
            val[((size_t)(o0 + lo) * BS * BS + r * BS + r) * 64 + lane] = bc_diag[(size_t)i * BS + r];
}

}  // namespace fem

// ---- host side --------------------------------------------------------------------

static GroupViews make_views(fem_mesh m)
{
    GroupViews gv{};
    gv.ng = (int)m->groups.size();
    for (int i = 0; i < gv.ng; i++) gv.g[i] = GroupView{m->groups[i].conn.p, m->groups[i].npe, m->groups[i].nel};
    return gv;
}

// The symbolic phase: everything that follows from the connectivity alone.
static std::shared_ptr<MatPattern> build_pattern(fem_mesh mesh)
{
    fem_ctx ctx = mesh->ctx;
    hipStream_t s = ctx->stream;
    FEM_HIP(hipEventRecord(ctx->t0, s));
    auto P = std::make_shared<MatPattern>();
    P->bs = mesh->pdim; P->nbrows = mesh->nnds;
    const int nn = mesh->nnds;
    P->nslices = (nn + 63) / 64;
    const int nrp = P->nslices * 64;
    const GroupViews gvs = make_views(mesh);

    // node -> incident elements (CSR on device), ascending codes inside every list
    DBuf<int> cursor(nn), tmp(exclusive_scan_tmp_ints((size_t)nrp + 1));
    DBuf<int>& n2e_ptr = P->n2e_ptr;
    DBuf<unsigned>& n2e = P->n2e;
    n2e_ptr.alloc((size_t)nn + 1);
    exclusive_scan_i32(mesh->valence.p, n2e_ptr.p, nn, tmp.p, s);
    int tot = 0;
    FEM_HIP(hipMemcpyAsync(&tot, n2e_ptr.p + nn, sizeof(int), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipStreamSynchronize(s));
    for (auto& g : mesh->groups)
        FEM_CHECK(g.nel < (1 << 27), 4, "fem_mat_create_aij: more than 2^27 elements of one type in a domain");
    n2e.alloc((size_t)std::max(tot, 1));
    cursor.zero(s);
    for (int g = 0; g < gvs.ng; g++) {
        const size_t nent = (size_t)gvs.g[g].nel * gvs.g[g].npe;
        hipLaunchKernelGGL(k_fill_n2e, dim3(std::min<unsigned>(cdiv(nent, 256), 8192)), dim3(256), 0, s, gvs.g[g], g,
                           n2e_ptr.p, cursor.p, n2e.p);
    }
    // node pairs coupled by cohesive elements only (CSR over nodes; absent on decks without them)
    DBuf<int> xptr, xadj;
    if (mesh->ncoh > 0) {
        std::vector<int> hp((size_t)nn + 1, 0), ha;
        for (int e = 0; e < mesh->ncoh; e++)
            for (int a = 0; a < 4; a++) hp[(size_t)mesh->h_coh_conn[4 * e + a] + 1] += 4;
        for (int v = 0; v < nn; v++) hp[v + 1] += hp[v];
        ha.resize((size_t)hp[nn]);
        std::vector<int> fill(hp.begin(), hp.end() - 1);
        for (int e = 0; e < mesh->ncoh; e++)
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++) ha[(size_t)fill[mesh->h_coh_conn[4 * e + a]]++] = mesh->h_coh_conn[4 * e + b];
        xptr.alloc(hp.size()); xptr.upload(hp.data(), hp.size(), s);
        xadj.alloc(std::max<size_t>(ha.size(), 1)); xadj.upload(ha.data(), ha.size(), s);
        FEM_HIP(hipStreamSynchronize(s));
    }
    // row lengths (and the sort of the incidence lists, the serialisation flags of the row-owner
    // assembly, the largest valence), slice widths, slice offsets
    P->rowlen.alloc(nn);
    P->slice_serial.alloc((size_t)P->nslices);
    P->slice_serial.zero(s);
    DBuf<int> err(2), width(P->nslices);
    err.zero(s);
    hipLaunchKernelGGL(k_sort_and_flag, dim3(cdiv(nn, 256)), dim3(256), 0, s, nn, n2e_ptr.p, n2e.p, gvs,
                       mesh->pdim == 3 ? 8 : 4, P->slice_serial.p, err.p + 1);
    hipLaunchKernelGGL(k_node_neighbours<false>, dim3(cdiv(nrp, 256)), dim3(256), 0, s, nn, nrp, n2e_ptr.p, n2e.p, gvs,
                       P->rowlen.p, (const int*)nullptr, (int*)nullptr, err.p, (const int*)xptr.p, (const int*)xadj.p);
    hipLaunchKernelGGL(k_slice_width, dim3(cdiv(P->nslices, 256)), dim3(256), 0, s, nn, P->nslices, P->rowlen.p,
                       width.p);
    P->boff.alloc((size_t)P->nslices + 1);
    exclusive_scan_i32(width.p, P->boff.p, P->nslices, tmp.p, s);
    P->h_boff.resize((size_t)P->nslices + 1);
    P->h_rowlen.resize(nn);
    int herr[2] = {0, 0};
    P->boff.download(P->h_boff.data(), P->h_boff.size(), s);
    P->rowlen.download(P->h_rowlen.data(), nn, s);
    err.download(herr, 2, s);
    FEM_HIP(hipStreamSynchronize(s));
    FEM_CHECK(!herr[0], 4, "fem_mat_create_aij: a node has more than " + std::to_string(MAXNB) + " neighbour nodes");
    P->max_valence = herr[1];
    const long long tb = P->h_boff[P->nslices];
    FEM_CHECK(tb * 64 * P->bs * P->bs < (1LL << 40), 4, "fem_mat_create_aij: matrix too large");
    P->padded_blocks = tb * 64;
    P->nblocks = 0;
    int maxw = 0;
    for (int v = 0; v < nn; v++) { P->nblocks += P->h_rowlen[v]; maxw = std::max(maxw, P->h_rowlen[v]); }
    FEM_CHECK(maxw < 65536, 4, "fem_mat_create_aij: row too long for 16-bit slots");
    P->maxw = maxw;
    P->bcol.alloc((size_t)P->padded_blocks);
    hipLaunchKernelGGL(k_node_neighbours<true>, dim3(cdiv(nrp, 256)), dim3(256), 0, s, nn, nrp, n2e_ptr.p, n2e.p, gvs,
                       P->rowlen.p, P->boff.p, P->bcol.p, err.p, (const int*)xptr.p, (const int*)xadj.p);
    // per-element slot maps (consumed by the assembly kernels)
    for (auto& g : mesh->groups) {
        const size_t tot_s = (size_t)g.nel * g.npe * g.npe;
        g.slot.alloc(tot_s);
        hipLaunchKernelGGL(k_slots, dim3(std::min<unsigned>(cdiv(tot_s, 256), 16384)), dim3(256), 0, s,
                           GroupView{g.conn.p, g.npe, g.nel}, P->rowlen.p, P->boff.p, P->bcol.p, g.slot.p);
    }
    FEM_HIP(hipGetLastError());
    FEM_HIP(hipEventRecord(ctx->t1, s));
    FEM_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    FEM_HIP(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    P->t_symbolic_ms = ms;
    return P;
}

std::shared_ptr<MatPattern> mesh_pattern(fem_mesh mesh)
{
    mesh->ctx->bind();
    if (!mesh->pattern) mesh->pattern = build_pattern(mesh);
    return mesh->pattern;
}

fem_mat mat_create_impl(fem_mesh mesh)
{
    FEM_CHECK(mesh, 1, "fem_mat_create_aij: null mesh");
    fem_ctx ctx = mesh->ctx;
    ctx->bind();
    if (!mesh->pattern) mesh->pattern = build_pattern(mesh);
    std::unique_ptr<fem_mat_s> A(new fem_mat_s(mesh->pattern));
    A->mesh = mesh; A->ctx = ctx;
    A->val.alloc((size_t)A->padded_blocks * A->bs * A->bs);
    A->lazy_zero = true;   // a new Mat is zero (MatCreateAIJ); see fem_mat_s::values()
    A->spmv_grid = (A->nslices + 3) / 4;
    A->spmv_partial.alloc((size_t)A->spmv_grid);
    return A.release();
}

// up to MAXSEG device-to-device copies in one launch (MatCopy: the values and the four penalty-row arrays -- five launches of a
// few microseconds each in front of every Jacobian of a Newton loop); sizes in 4-byte words
struct CopySegs { static constexpr int MAXSEG = 6; unsigned* dst[MAXSEG]; const unsigned* src[MAXSEG]; size_t end[MAXSEG]; int n; };
__global__ void __launch_bounds__(256) k_copy_segments(CopySegs c)
{
    const size_t st = (size_t)gridDim.x * blockDim.x, total = c.end[c.n - 1];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += st) {
        int k = 0;
        while (i >= c.end[k]) k++;
        const size_t o = i - (k ? c.end[k - 1] : 0);
        c.dst[k][o] = c.src[k][o];
    }
}
void device_copy_segments(hipStream_t s, int n, void* const* dst, const void* const* src, const size_t* bytes)
{
    CopySegs c;
    c.n = 0;
    size_t total = 0;
    for (int i = 0; i < n; i++) {
        if (!bytes[i]) continue;
        FEM_CHECK(bytes[i] % 4 == 0 && c.n < CopySegs::MAXSEG, 2, "device_copy_segments: bad segment");
        total += bytes[i] / 4;
        c.dst[c.n] = static_cast<unsigned*>(dst[i]); c.src[c.n] = static_cast<const unsigned*>(src[i]); c.end[c.n] = total;
        c.n++;
    }
    if (!c.n) return;
    hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)std::min<size_t>(cdiv(total, 256), 4096)), dim3(256), 0, s, c);
    FEM_HIP(hipGetLastError());
}

void ebe_mult_impl(fem_mat A, const double* x, double* y, double* partial, const int* done);   // elem_kernels.hip

void mat_mult_raw(fem_mat A, const double* x, double* y, double* partial, const int* done)
{
    if (A->mult_ebe) { ebe_mult_impl(A, x, y, partial, done); return; }
    hipStream_t s = A->ctx->stream;
    // (values + block columns against the Infinity Cache: see ld_mat)
    const bool stream = A->val.bytes() + A->bcol.bytes() > ((size_t)200 << 20);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(A->spmv_grid), dim3(256), 0, s, A->nslices, A->nbrows, A->boff.p, A->bcol.p, A->values(),
                           x, y, partial, done);
    };
    if (A->bs == 3) { if (stream) launch(k_spmv_sell<3, true>); else launch(k_spmv_sell<3, false>); }
    else { if (stream) launch(k_spmv_sell<2, true>); else launch(k_spmv_sell<2, false>); }
}

void mat_get_diagonal_raw(fem_mat A, double* d)
{
    hipStream_t s = A->ctx->stream;
    if (A->bs == 3)
        hipLaunchKernelGGL(k_get_diag<3>, dim3(cdiv(A->nbrows, 256)), dim3(256), 0, s, A->nbrows, A->rowlen.p,
                           A->boff.p, A->bcol.p, A->values(), d);
    else
        hipLaunchKernelGGL(k_get_diag<2>, dim3(cdiv(A->nbrows, 256)), dim3(256), 0, s, A->nbrows, A->rowlen.p,
                           A->boff.p, A->bcol.p, A->values(), d);
}

void mat_apply_kbc_impl(fem_mat A, int nbcs_in, const int* bcnode_in)
{
    if (nbcs_in <= 0) return;
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int pd = A->bs;
    // One record per node.  The reference INSERTs the penalty for every record in turn
    // (src/m_global.F90:264-299), so a dof is constrained as soon as ANY record of its node
    // flags it: records naming the same node are merged before the device sees them (the
    // per-node tables of the element-by-element product hold one entry per node).
    std::vector<int> merged;
    {
        std::vector<int> pos((size_t)A->nbrows, -1);
        merged.reserve((size_t)nbcs_in * (1 + pd));
        for (int i = 0; i < nbcs_in; i++) {
            const int* rec = bcnode_in + (size_t)i * (1 + pd);
            FEM_CHECK(rec[0] >= 0 && rec[0] < A->nbrows, 3, "fem_mat_apply_kbc: node id out of range");
            int& p = pos[rec[0]];
            if (p < 0) {
                p = (int)(merged.size() / (1 + pd));
                merged.push_back(rec[0]);
                for (int d = 0; d < pd; d++) merged.push_back(rec[1 + d] == 0 ? 0 : 1);
            } else {
                for (int d = 0; d < pd; d++)
                    if (rec[1 + d] == 0) merged[(size_t)p * (1 + pd) + 1 + d] = 0;
            }
        }
    }
    const int nbcs = (int)(merged.size() / (1 + pd));
    const unsigned char* owned = A->mesh->halo ? A->mesh->halo->owned.p : nullptr;
    // a different list than the one already applied to this assembly: put the assembled diagonal back first
    if (A->bc_saved && merged != A->h_bc) {
        const int nold = (int)(A->h_bc.size() / (1 + pd));
        DBuf<int> d_old(A->h_bc.size());
        d_old.upload(A->h_bc.data(), A->h_bc.size(), s);
        if (pd == 3)
            hipLaunchKernelGGL(k_restore_kbc<3>, dim3(cdiv(nold, 256)), dim3(256), 0, s, nold, d_old.p, A->rowlen.p,
                               A->boff.p, A->bcol.p, A->values(), A->bc_diag.p);
        else
            hipLaunchKernelGGL(k_restore_kbc<2>, dim3(cdiv(nold, 256)), dim3(256), 0, s, nold, d_old.p, A->rowlen.p,
                               A->boff.p, A->bcol.p, A->values(), A->bc_diag.p);
        FEM_HIP(hipStreamSynchronize(s));
        A->bc_saved = false;
    }
    bool uploaded = false;
    if (merged != A->h_bc_dev) {
        A->h_bc_dev.clear();
        A->d_bc_dev.alloc(merged.size());
        A->d_bc_dev.upload(merged.data(), merged.size(), s);
        uploaded = true;
    }
    const DBuf<int>& d_bc = A->d_bc_dev;
    // the element-by-element product needs (inserted value - assembled diagonal) of every penalty row;
    // only the first ApplyKBC after an assembly sees the assembled diagonal
    int* bcd = nullptr;
    double *bcs = nullptr, *bcg = nullptr;
    if (!A->bc_saved) {
        A->nbc_dof = nbcs * pd;
        A->bc_dof.alloc((size_t)A->nbc_dof); A->bc_shift.alloc((size_t)A->nbc_dof); A->bc_diag.alloc((size_t)A->nbc_dof);
        A->bc_node.alloc((size_t)A->nbrows);
        FEM_HIP(hipMemsetAsync(A->bc_node.p, 0xff, sizeof(int) * (size_t)A->nbrows, s));   // -1 everywhere
        bcd = A->bc_dof.p; bcs = A->bc_shift.p; bcg = A->bc_diag.p;
    }
    if (pd == 3)
        hipLaunchKernelGGL(k_apply_kbc<3>, dim3(cdiv(nbcs, 256)), dim3(256), 0, s, nbcs, d_bc.p, A->rowlen.p, A->boff.p,
                           A->bcol.p, A->values(), owned, PENALTY_PARAM, bcd, bcs, bcg, A->bc_node.p);
    else
        hipLaunchKernelGGL(k_apply_kbc<2>, dim3(cdiv(nbcs, 256)), dim3(256), 0, s, nbcs, d_bc.p, A->rowlen.p, A->boff.p,
                           A->bcol.p, A->values(), owned, PENALTY_PARAM, bcd, bcs, bcg, A->bc_node.p);
    A->bc_saved = true;
    FEM_HIP(hipGetLastError());
    if (uploaded) {
        FEM_HIP(hipStreamSynchronize(s));   // (the upload reads `merged`)
        A->h_bc_dev = merged;
    }
    A->h_bc = std::move(merged);
}

// MatSetValues (src/m_global.F90:257-260 ADD_VALUES from FormLocalK, :296 INSERT_VALUES of the penalty): a logically dense
// m x n block of scalar entries given by 0-based dof indices, row-major as PETSc's default; negative indices are skipped as
// PETSc skips them; an entry outside the sparsity of MatCreateAIJ is an error (PETSc with MAT_NEW_NONZERO_LOCATION_ERR).  The
// host looks the positions up in the block rows concerned, one small kernel writes them: meant for a host that patches a few
// rows (the reference's penalty INSERT, tests), not for the element loop (fem_mat_assemble_elements).
__global__ void __launch_bounds__(256) k_set_values(int n, const long long* __restrict__ pos, const double* __restrict__ v,
                                                    int add, double* __restrict__ val)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (add) val[pos[i]] += v[i];
    else val[pos[i]] = v[i];
}
void mat_set_values_impl(fem_mat A, int m, const int* rows, int n, const int* cols, const double* v, int mode)
{
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int bs = A->bs, nn = A->nbrows;
    const long long nd = (long long)nn * bs;
    std::vector<long long> pos;
    std::vector<double> vals;
    std::vector<int> hc;
    // duplicate positions inside one call: ADD accumulates, INSERT keeps the last one, as a sequential MatSetValues would
    std::map<long long, size_t> seen;
    for (int i = 0; i < m; i++) {
        const int row = rows[i];
        if (row < 0) continue;
        FEM_CHECK(row < nd, 3, "fem_mat_set_values: row index out of range");
        const int node = row / bs, r = row % bs, lane = node & 63, len = A->h_rowlen[node];
        const size_t o0 = (size_t)A->h_boff[node >> 6];
        hc.resize((size_t)std::max(len, 1));
        if (len)
            FEM_HIP(hipMemcpy2DAsync(hc.data(), sizeof(int), A->bcol.p + o0 * 64 + lane, 64 * sizeof(int), sizeof(int), (size_t)len,
                                     hipMemcpyDeviceToHost, s));
        FEM_HIP(hipStreamSynchronize(s));
        for (int jn = 0; jn < n; jn++) {
            const int col = cols[jn];
            if (col < 0) continue;
            FEM_CHECK(col < nd, 3, "fem_mat_set_values: column index out of range");
            const int cnode = col / bs, c = col % bs;
            int j = -1;
            for (int q = 0; q < len; q++)
                if (hc[(size_t)q] == cnode) { j = q; break; }
            FEM_CHECK(j >= 0, 4, "fem_mat_set_values: entry outside the sparsity pattern of MatCreateAIJ");
            const long long p = (long long)(((o0 + (size_t)j) * bs * bs + (size_t)r * bs + c) * 64 + lane);
            const double x = v[(size_t)i * n + jn];
            if (row != col) A->unsymmetric = true;   // (what the host writes off the diagonal is not known to be symmetric: LU, not Cholesky)
            auto it = seen.find(p);
            if (it == seen.end()) { seen[p] = pos.size(); pos.push_back(p); vals.push_back(x); }
            else if (mode == 2) vals[it->second] += x;
            else vals[it->second] = x;
        }
    }
    if (pos.empty()) return;
    DBuf<long long> dpos; DBuf<double> dval;
    dpos.alloc(pos.size()); dval.alloc(vals.size());
    dpos.upload(pos.data(), pos.size(), s); dval.upload(vals.data(), vals.size(), s);
    double* dst = A->values();
    hipLaunchKernelGGL(k_set_values, dim3(cdiv(pos.size(), 256)), dim3(256), 0, s, (int)pos.size(), dpos.p, dval.p, mode == 2 ? 1 : 0, dst);
    FEM_HIP(hipGetLastError());
    FEM_HIP(hipStreamSynchronize(s));
}

// AIJ (scalar CSR) view for MatView-style checks: unpack the slices on the host.
void mat_get_csr_impl(fem_mat A, int* rowptr, int* colind, double* val)
{
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const int bs = A->bs, nn = A->nbrows;
    FEM_CHECK(A->nblocks * bs * bs < (1LL << 31), 4,
              "fem_mat_get_csr: the scalar CSR view has int32 row pointers; this matrix has more than 2^31 entries");
    if (rowptr) {
        long long k = 0;
        rowptr[0] = 0;
        for (int v = 0; v < nn; v++)
            for (int r = 0; r < bs; r++) { k += (long long)A->h_rowlen[v] * bs; rowptr[(size_t)bs * v + r + 1] = (int)k; }
    }
    if (!colind && !val) return;
    std::vector<int> hc((size_t)A->padded_blocks);
    std::vector<double> hv;
    A->bcol.download(hc.data(), hc.size(), s);
    if (val) { hv.resize((size_t)A->padded_blocks * bs * bs); (void)A->values(); A->val.download(hv.data(), hv.size(), s); }
    FEM_HIP(hipStreamSynchronize(s));
    size_t k = 0;
    for (int v = 0; v < nn; v++) {
        const size_t o0 = (size_t)A->h_boff[v >> 6];
        const int lane = v & 63, len = A->h_rowlen[v];
        for (int r = 0; r < bs; r++)
            for (int j = 0; j < len; j++)
                for (int c = 0; c < bs; c++) {
                    if (colind) colind[k] = bs * hc[(o0 + j) * 64 + lane] + c;
                    if (val) val[k] = hv[((o0 + j) * bs * bs + r * bs + c) * 64 + lane];
                    k++;
                }
    }
}
