// fem_amd/csrc/direct.hip -- banded Cholesky direct solve (see direct.h).
#include "direct.h"

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "comm.h"
#include "grid_barrier.h"

using namespace fem;

namespace fem {

// ---- reverse Cuthill-McKee (host; set-up only) -------------------------------------------------------------
std::vector<int> rcm_order(int n, const std::vector<int>& ptr, const std::vector<int>& adj)
{
    std::vector<int> order;           // order[k] = old id of the node numbered k (before reversal)
    order.reserve(n);
    std::vector<int> level(n, -1), deg(n);
    for (int v = 0; v < n; v++) deg[v] = ptr[v + 1] - ptr[v];
    std::vector<char> done(n, 0);
    std::vector<int> by_deg(n);
    for (int v = 0; v < n; v++) by_deg[v] = v;
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return deg[a] < deg[b]; });
    std::vector<int> queue, nb;
    // breadth-first levels from `root` over the not yet numbered nodes; returns the last level's minimum-degree node
    auto bfs = [&](int root, int& depth) {
        queue.clear();
        queue.push_back(root);
        level[root] = 0;
        for (size_t h = 0; h < queue.size(); h++) {
            const int v = queue[h];
            for (int e = ptr[v]; e < ptr[v + 1]; e++) {
                const int u = adj[e];
                if (!done[u] && level[u] < 0) { level[u] = level[v] + 1; queue.push_back(u); }
            }
        }
        depth = level[queue.back()];
        int far = queue.back();
        for (size_t h = queue.size(); h-- > 0 && level[queue[h]] == depth;)
            if (deg[queue[h]] < deg[far]) far = queue[h];
        for (int v : queue) level[v] = -1;
        return far;
    };
    size_t scan = 0;
    while ((int)order.size() < n) {
        while (done[by_deg[scan]]) scan++;
        int root = by_deg[scan];                 // minimum-degree node of the next component
        // pseudo-peripheral node: walk to the far end while the eccentricity grows
        int depth = -1;
        for (int pass = 0; pass < 8; pass++) {
            int d;
            const int far = bfs(root, d);
            if (d <= depth) break;
            depth = d;
            root = far;
        }
        // Cuthill-McKee numbering of the component
        const size_t first = order.size();
        order.push_back(root);
        done[root] = 1;
        for (size_t h = first; h < order.size(); h++) {
            const int v = order[h];
            nb.clear();
            for (int e = ptr[v]; e < ptr[v + 1]; e++)
                if (!done[adj[e]]) { nb.push_back(adj[e]); done[adj[e]] = 1; }
            std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] < deg[b]; });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    std::vector<int> perm(n);          // perm[old] = new, reversed numbering
    for (int k = 0; k < n; k++) perm[order[k]] = n - 1 - k;
    return perm;
}

// ---- kernels ------------------------------------------------------------------------------------
constexpr int NB = 64;
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t band_idx(long long i, long long j, int bw) { return (size_t)(i * bw + j + bw); }

// lower triangle of P K P^T into the band
template <int BS>
__global__ void __launch_bounds__(256) k_band_fill(int nbrows, const int* __restrict__ boff, const int* __restrict__ bcol,
                                                   const double* __restrict__ val, const int* __restrict__ rowlen,
                                                   const int* __restrict__ perm, int bw, double* __restrict__ ab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbrows) return;
    const size_t o0 = (size_t)boff[i >> 6];
    const int ln = i & 63, len = rowlen[i];
    const long long pi = perm[i];
    for (int j = 0; j < len; j++) {
        const long long pc = perm[bcol[(o0 + j) * 64 + ln]];
#pragma unroll
        for (int r = 0; r < BS; r++)
#pragma unroll
            for (int c = 0; c < BS; c++) {
                const long long gi = pi * BS + r, gj = pc * BS + c;
                if (gj <= gi) ab[band_idx(gi, gj, bw)] = val[((o0 + j) * BS * BS + r * BS + c) * 64 + ln];
            }
    }
}

__global__ void __launch_bounds__(256) k_perm_dof(int nn, int bs, const int* __restrict__ perm, int* __restrict__ perm_dof)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    for (int r = 0; r < bs; r++) perm_dof[(size_t)v * bs + r] = perm[v] * bs + r;
}
__global__ void __launch_bounds__(256) k_permute_in(size_t n, const int* __restrict__ perm, const double* __restrict__ b,
                                                    double* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[perm[i]] = b[i];
}
__global__ void __launch_bounds__(256) k_permute_out(size_t n, const int* __restrict__ perm, const double* __restrict__ in,
                                                     double* __restrict__ x)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = in[perm[i]];
}

// Loaders (256 threads).  Masked entries read ab[0] (allocated, unused) and are replaced after the
// load, so that the 16 loads of a thread are issued back to back instead of behind branches.
// the kb x kb diagonal block at k0 (lower triangle, zero outside the band); identity padding
__device__ __forceinline__ void load_diag_block(double (*L)[NB + 1], const double* __restrict__ ab, int bw, long long k0,
                                                int kb, int tid)
{
    double v[NB * NB / 256];
#pragma unroll
    for (int e = 0; e < NB * NB / 256; e++) {
        const int t = tid + e * 256, i = t >> 6, j = t & 63;
        const bool ok = i < kb && j <= i && i - j <= bw;
        v[e] = ab[ok ? band_idx(k0 + i, k0 + j, bw) : 0];
    }
#pragma unroll
    for (int e = 0; e < NB * NB / 256; e++) {
        const int t = tid + e * 256, i = t >> 6, j = t & 63;
        const bool ok = i < kb && j <= i && i - j <= bw;
        L[i][j] = ok ? v[e] : ((i == j) ? 1.0 : 0.0);
    }
}
// 64 rows from R0 of the panel below the diagonal block (columns k0 .. k0+kb-1), zero outside the band
__device__ __forceinline__ void load_panel_tile(double (*T)[NB + 1], const double* __restrict__ ab, int bw, long long k0,
                                                int kb, long long R0, long long lim, int tid)
{
    double v[NB * NB / 256];
#pragma unroll
    for (int e = 0; e < NB * NB / 256; e++) {
        const int t = tid + e * 256, i = t >> 6, j = t & 63;
        const long long row = R0 + i;
        const bool ok = row < lim && j < kb && row - (k0 + j) <= bw;
        v[e] = ab[ok ? band_idx(row, k0 + j, bw) : 0];
    }
#pragma unroll
    for (int e = 0; e < NB * NB / 256; e++) {
        const int t = tid + e * 256, i = t >> 6, j = t & 63;
        const long long row = R0 + i;
        const bool ok = row < lim && j < kb && row - (k0 + j) <= bw;
        T[i][j] = ok ? v[e] : 0.0;
    }
}
__device__ __forceinline__ void load_dense64(double (*L)[NB + 1], const double* __restrict__ src, int tid)
{
    double v[NB * NB / 256];
#pragma unroll
    for (int e = 0; e < NB * NB / 256; e++) v[e] = src[tid + e * 256];
#pragma unroll
    for (int e = 0; e < NB * NB / 256; e++) {
        const int t = tid + e * 256;
        L[t >> 6][t & 63] = v[e];
    }
}

// 64x64 product on the fp64 matrix cores, D = A B^T with A = Ap (rows i), B = Bp (rows j), both
// [64][64+1] in LDS; four wavefronts, each 16 rows x 64 columns = four 16x16 accumulators
// (v_mfma_f64_16x16x4_f64):
//   A operand  lane l holds Ap[16w + (l&15)][k + (l>>4)],  B operand  lane l holds Bp[16c + (l&15)][k + (l>>4)]
//   C/D        register q of lane l is (row 16w + (l>>4) + 4q, col 16c + (l&15))
__device__ __forceinline__ void mfma_tile_abt(const double (*Ap)[NB + 1], const double (*Bp)[NB + 1], int wave, int l,
                                              double4_t (&acc)[4])
{
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < NB; kk += 4) {
        const double a = Ap[16 * wave + (l & 15)][kk + (l >> 4)];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const double b = Bp[16 * c + (l & 15)][kk + (l >> 4)];
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        }
    }
}

// Cholesky of the diagonal block, in place, and the inverse of its factor into linv[64][64]
// (zero above the diagonal, identity beyond kb): the panel solve and both substitutions become products.
// 16 x 16 threads, each keeps a 4x4 sub-block in registers; per column one broadcast of the pivot
// and one of the scaled column through LDS (two barriers).  Blocks shorter than 64 run on their
// identity padding.  The inverse is one unrolled forward substitution per column (one wavefront).
__global__ void __launch_bounds__(256) k_potrf(int bw, long long k0, int kb, double* __restrict__ ab,
                                               double* __restrict__ linv, int* __restrict__ flag)
{
    __shared__ double a[NB][NB + 1], colbuf[NB], dk;
    const int tid = threadIdx.x;
    load_diag_block(a, ab, bw, k0, kb, tid);
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double c[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) c[r][q] = a[4 * ty + r][4 * tx + q];
    for (int kq = 0; kq < NB / 4; kq++) {
#pragma unroll
        for (int kr = 0; kr < 4; kr++) {
            const int k = 4 * kq + kr;
            if (ty == kq && tx == kq) dk = c[kr][kr];
            __syncthreads();
            double d = dk;
            if (!(d > 0.0)) {          // not positive definite (or NaN): flag it and keep the arithmetic finite
                if (tid == 0) atomicExch(flag, 1);
                d = 1.0;
            }
            const double sq = sqrt(d), inv = 1.0 / sq;
            if (tx == kq && ty >= kq) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = 4 * ty + r;
                    if (i > k) c[r][kr] *= inv;
                    else if (i == k) c[r][kr] = sq;
                    colbuf[i] = c[r][kr];
                }
            }
            __syncthreads();
            if (ty >= tx && ty >= kq) {
                double ci[4], cj[4];
#pragma unroll
                for (int r = 0; r < 4; r++) { ci[r] = colbuf[4 * ty + r]; cj[r] = colbuf[4 * tx + r]; }
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (4 * tx + q > k && 4 * ty + r >= 4 * tx + q) c[r][q] -= ci[r] * cj[q];
            }
        }
    }
    __syncthreads();
    if (ty >= tx) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 4 * ty + r, j = 4 * tx + q;
                if (j <= i) {
                    a[i][j] = c[r][q];
                    if (i < kb && i - j <= bw) ab[band_idx(k0 + i, k0 + j, bw)] = c[r][q];
                }
            }
    }
    __syncthreads();
    if (tid < NB) {                // column tid of L^-1: forward substitution on the unit vector
        double x[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            double t = (i == tid) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; k++) t -= a[i][k] * x[k];
            x[i] = t / a[i][i];
            linv[i * NB + tid] = x[i];
        }
    }
}

// panel below the diagonal block:  L21 = A21 L11^-T  as a product with the inverse (64 rows per workgroup)
__global__ void __launch_bounds__(256) k_trsm(int bw, long long k0, int kb, int m, double* __restrict__ ab,
                                              const double* __restrict__ linv)
{
    __shared__ double T[NB][NB + 1], Li[NB][NB + 1];
    const int tid = threadIdx.x;
    const long long R0 = k0 + kb + (long long)blockIdx.x * NB, lim = k0 + kb + m;
    load_panel_tile(T, ab, bw, k0, kb, R0, lim, tid);
    load_dense64(Li, linv, tid);
    __syncthreads();
    const int wave = tid >> 6, l = tid & 63;
    double4_t acc[4];
    mfma_tile_abt(T, Li, wave, l, acc);        // X[i][j] = sum_k T[i][k] Linv[j][k]
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const long long row = R0 + 16 * wave + (l >> 4) + 4 * q;
            const int j = 16 * c + (l & 15);
            if (row < lim && j < kb && row - (k0 + j) <= bw) ab[band_idx(row, k0 + j, bw)] = acc[c][q];
        }
}

// trailing window:  A22 -= L21 L21^T  (lower triangle), one 64x64 tile per workgroup
__global__ void __launch_bounds__(256) k_syrk(int bw, long long k0, int kb, int m, double* __restrict__ ab)
{
    __shared__ double Ap[NB][NB + 1], Bp[NB][NB + 1];
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    int ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > t) ti--;
    while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
    const int tj = t - ti * (ti + 1) / 2;
    const long long base = k0 + kb, lim = base + m;
    const long long I0 = base + (long long)NB * ti, J0 = base + (long long)NB * tj;
    load_panel_tile(Ap, ab, bw, k0, kb, I0, lim, tid);
    load_panel_tile(Bp, ab, bw, k0, kb, J0, lim, tid);
    __syncthreads();
    const int wave = tid >> 6, l = tid & 63;
    double4_t acc[4];
    mfma_tile_abt(Ap, Bp, wave, l, acc);
    double old[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) {     // all sixteen reads in flight before the first write
            const long long row = I0 + 16 * wave + (l >> 4) + 4 * q, col = J0 + 16 * c + (l & 15);
            const bool ok = row < lim && col < lim && col <= row;
            old[c][q] = ab[ok ? band_idx(row, col, bw) : 0];
        }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const long long row = I0 + 16 * wave + (l >> 4) + 4 * q, col = J0 + 16 * c + (l & 15);
            if (row < lim && col < lim && col <= row) ab[band_idx(row, col, bw)] = old[c][q] - acc[c][q];
        }
}

// forward substitution, block k0:  y_k = L11^-1 b_k  (every wavefront for itself, lane = row), then
// b[row] -= L21[row] . y_k for this wavefront's rows below the block
constexpr int FWD_ROWS = 8;    // per wavefront; a workgroup of four covers 32 rows
__global__ void __launch_bounds__(256) k_fwd(int bw, long long k0, int kb, int m, const double* __restrict__ ab,
                                             const double* __restrict__ linv, double* __restrict__ b,
                                             double* __restrict__ y)
{
    __shared__ double Li[NB][NB + 1], bk[NB];
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    load_dense64(Li, linv, tid);
    if (tid < NB) bk[tid] = (tid < kb) ? b[k0 + tid] : 0.0;
    __syncthreads();
    double v = 0.0;
#pragma unroll 8
    for (int j = 0; j < NB; j++) v += Li[l][j] * bk[j];
    if (blockIdx.x == 0 && wave == 0 && l < kb) y[k0 + l] = v;
    const long long lim = k0 + kb + m, row0 = k0 + kb + ((long long)blockIdx.x * 4 + wave) * FWD_ROWS;
    double t[FWD_ROWS];
#pragma unroll
    for (int rr = 0; rr < FWD_ROWS; rr++) {   // independent loads first, reductions after
        const long long row = row0 + rr;
        const bool ok = row < lim && l < kb && row - (k0 + l) <= bw;
        const double a = ab[ok ? band_idx(row, k0 + l, bw) : 0];
        t[rr] = ok ? a * v : 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < FWD_ROWS; rr++)
        for (int o = 32; o > 0; o >>= 1) t[rr] += __shfl_down(t[rr], o);
    if (l == 0) {
#pragma unroll
        for (int rr = 0; rr < FWD_ROWS; rr++)
            if (row0 + rr < lim) b[row0 + rr] -= t[rr];
    }
}

// backward substitution, block k0:  x_k = L11^-T y_k, then y[col] -= sum_r L[k0+r][col] x_k[r] for the
// columns left of the block (thread per column: coalesced along the band rows)
__global__ void __launch_bounds__(256) k_bwd(int bw, long long k0, int kb, const double* __restrict__ ab,
                                             const double* __restrict__ linv, double* __restrict__ y,
                                             double* __restrict__ x)
{
    __shared__ double Li[NB][NB + 1], yk[NB], xs[NB];
    const int tid = threadIdx.x;
    load_dense64(Li, linv, tid);
    if (tid < NB) yk[tid] = (tid < kb) ? y[k0 + tid] : 0.0;
    __syncthreads();
    if (tid < NB) {
        double v = 0.0;
#pragma unroll 8
        for (int j = 0; j < NB; j++) v += Li[j][tid] * yk[j];     // (L^-1)^T
        xs[tid] = v;
        if (blockIdx.x == 0 && tid < kb) x[k0 + tid] = v;
    }
    __syncthreads();
    const long long c_lo = (k0 > bw) ? k0 - bw : 0;
    const long long col = c_lo + (long long)blockIdx.x * 256 + tid;
    if (col < k0) {
        double s = 0.0;
        for (int r0 = 0; r0 < NB; r0 += 16) {
            double a[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {     // sixteen independent loads in flight
                const bool ok = r0 + r < kb && k0 + r0 + r - col <= bw;
                a[r] = ab[ok ? band_idx(k0 + r0 + r, col, bw) : 0];
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const bool ok = r0 + r < kb && k0 + r0 + r - col <= bw;
                s += ok ? a[r] * xs[r0 + r] : 0.0;
            }
        }
        y[col] -= s;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
void BandCholesky::setup(fem_mat A_)
{
    A = A_; bs = A->bs;
    const int nn = A->nbrows;
    n = (long long)nn * bs;
    FEM_CHECK(!(A->mesh->halo && A->mesh->halo->comm->size > 1), 9,
              "pc lu: the direct solve is single-domain (use -ksp_type cg on partitioned meshes)");
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;

    // node graph from the block structure, reverse Cuthill-McKee numbering
    std::vector<int> hc((size_t)A->padded_blocks);
    A->bcol.download(hc.data(), hc.size(), s);
    FEM_HIP(hipStreamSynchronize(s));
    std::vector<int> ptr(nn + 1, 0), adj;
    for (int v = 0; v < nn; v++) ptr[v + 1] = ptr[v] + A->h_rowlen[v];
    adj.resize((size_t)ptr[nn]);
    for (int v = 0; v < nn; v++) {
        const size_t o0 = (size_t)A->h_boff[v >> 6];
        for (int j = 0; j < A->h_rowlen[v]; j++) adj[(size_t)ptr[v] + j] = hc[(o0 + j) * 64 + (v & 63)];
    }
    // numbering: the caller's own (structured meshes arrive well numbered) or reverse
    // Cuthill-McKee, whichever gives the narrower band
    std::vector<int> perm = rcm_order(nn, ptr, adj);
    long long bwn = 0, bw_nat = 0;
    for (int v = 0; v < nn; v++)
        for (int e = ptr[v]; e < ptr[v + 1]; e++) {
            bwn = std::max(bwn, (long long)std::abs(perm[v] - perm[adj[e]]));
            bw_nat = std::max(bw_nat, (long long)std::abs(v - adj[e]));
        }
    if (bw_nat <= bwn) {
        for (int v = 0; v < nn; v++) perm[v] = v;
        bwn = bw_nat;
    }
    bw = (int)std::min<long long>(bwn * bs + bs - 1, n - 1);

    size_t free_b = 0, total_b = 0;
    const double need = (double)n * (bw + 1) * sizeof(double);
    // (a factor of less than 1 GiB fits any state of the device this library runs on; hipMemGetInfo is not free -- round 6: 3-4 ms per
    // call in a process that has allocated and cached tens of GB, five calls per Newton run of config 5 -- and pool_trim empties the cache)
    if (need < 1073741824.0) free_b = (size_t)1 << 40;
    else {
        pool_trim();                      // cached blocks count as free memory for this decision
        FEM_HIP(hipMemGetInfo(&free_b, &total_b));
    }
    FEM_CHECK(need < 0.8 * (double)free_b, 9,
              "pc lu: the band of the factor needs " + std::to_string((long long)(need / 1048576.0)) +
                  " MiB (n = " + std::to_string(n) + ", half bandwidth " + std::to_string(bw) +
                  "), which does not fit the device: use -ksp_type cg");

    DBuf<int> dperm((size_t)nn);
    dperm.upload(perm.data(), perm.size(), s);
    perm_dof.alloc((size_t)n);
    hipLaunchKernelGGL(k_perm_dof, dim3(cdiv(nn, 256)), dim3(256), 0, s, nn, bs, dperm.p, perm_dof.p);
    ab.alloc((size_t)n * (bw + 1));
    ab.zero(s);
    if (bs == 3)
        hipLaunchKernelGGL(k_band_fill<3>, dim3(cdiv(nn, 256)), dim3(256), 0, s, nn, A->boff.p, A->bcol.p, A->values(),
                           A->rowlen.p, dperm.p, bw, ab.p);
    else
        hipLaunchKernelGGL(k_band_fill<2>, dim3(cdiv(nn, 256)), dim3(256), 0, s, nn, A->boff.p, A->bcol.p, A->values(),
                           A->rowlen.p, dperm.p, bw, ab.p);
    linv.alloc((size_t)((n + NB - 1) / NB) * NB * NB);
    v0.alloc((size_t)n); v1.alloc((size_t)n); v2.alloc((size_t)n);
    flag.alloc(1);
    flag.zero(s);

    for (long long k0 = 0; k0 < n; k0 += NB) {
        const int kb = (int)std::min<long long>(NB, n - k0);
        const int m = (int)(std::min<long long>(n, k0 + kb + bw) - (k0 + kb));
        double* li = linv.p + (size_t)(k0 / NB) * NB * NB;
        hipLaunchKernelGGL(k_potrf, dim3(1), dim3(256), 0, s, bw, k0, kb, ab.p, li, flag.p);
        if (m > 0) {
            const int tiles = (m + NB - 1) / NB;
            hipLaunchKernelGGL(k_trsm, dim3(tiles), dim3(256), 0, s, bw, k0, kb, m, ab.p, li);
            hipLaunchKernelGGL(k_syrk, dim3(tiles * (tiles + 1) / 2), dim3(256), 0, s, bw, k0, kb, m, ab.p);
        }
    }
    int hflag = 0;
    flag.download(&hflag, 1, s);
    FEM_HIP(hipGetLastError());
    FEM_HIP(hipStreamSynchronize(s));   // dperm and the host staging go out of scope
    failed = hflag != 0;
}

// ---- general band LU (unsymmetric Jacobians) ------------------------------------------------------------------
// full band of the renumbered matrix: element (i, j) at ab[i*wb + j - i + kl]
template <int BS>
__global__ void __launch_bounds__(256) k_bandfull_fill(int nn, const int* __restrict__ boff, const int* __restrict__ bcol,
                                                       const double* __restrict__ val, const int* __restrict__ rowlen,
                                                       const int* __restrict__ perm, int kl, int wb, double* __restrict__ ab,
                                                       int* __restrict__ flag2, double* __restrict__ amax)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { flag2[0] = 0; flag2[1] = 0; *amax = 0.0; }   // (what the kernels behind this one count into: two fills less per set-up)
    if (i >= nn) return;
    const int ln = i & 63, len = rowlen[i], o0 = boff[i >> 6];
    const long long pi = perm[i];
    for (int j = 0; j < len; j++) {
        const long long pj = perm[bcol[(size_t)(o0 + j) * 64 + ln]];
        for (int r = 0; r < BS; r++)
            for (int c = 0; c < BS; c++) {
                const long long gi = pi * BS + r, gj = pj * BS + c;
                ab[gi * wb + (gj - gi + kl)] = val[((size_t)(o0 + j) * BS * BS + r * BS + c) * 64 + ln];
            }
    }
}

// dgbtf2: one workgroup, column by column
__global__ void __launch_bounds__(256) k_band_lu(long long n, int kl, int wb, double* __restrict__ ab, int* __restrict__ ipiv,
                                                 const double* __restrict__ amax, int* __restrict__ flag)
{
    __shared__ double s_piv;
    __shared__ int s_p;
    const double tiny = 1e-6 * (*amax);
    const int ku2 = 2 * kl;   // upper band incl. fill
    for (long long k = 0; k < n; k++) {
        const int km = (int)min((long long)kl, n - 1 - k);       // rows below
        const int nc = (int)min((long long)ku2, n - 1 - k);      // columns to the right
        if (threadIdx.x == 0) {
            int p = 0;
            double best = fabs(ab[k * wb + kl]);
            for (int r = 1; r <= km; r++) {
                const double a = fabs(ab[(k + r) * wb + (kl - r)]);
                if (a > best) { best = a; p = r; }
            }
            s_p = p; s_piv = best;
            ipiv[k] = (int)(k + p);
        }
        __syncthreads();
        const int p = s_p;
        if (!(s_piv > tiny)) {   // null pivot (flag[0] counts them): the unknown is fixed to zero (row and column dropped);
                                 // a NaN in the column (flag[1]) is treated the same way and reported as a failed factorisation
            if (threadIdx.x == 0) { ipiv[k] = -1 - (int)k; atomicAdd(flag, 1); if (isnan(s_piv)) atomicExch(flag + 1, 1); }
            for (int r = threadIdx.x + 1; r <= km; r += 256) ab[(k + r) * wb + (kl - r)] = 0.0;
            for (int c = threadIdx.x + 1; c <= nc; c += 256) ab[k * wb + kl + c] = 0.0;
            if (threadIdx.x == 0) ab[k * wb + kl] = 1.0;
            __syncthreads();
            continue;
        }
        if (p > 0)   // exchange rows k and k+p over columns k .. k+nc
            for (int c = threadIdx.x; c <= nc; c += 256) {
                double* a = ab + k * wb + kl + c;
                double* b = ab + (k + p) * wb + (kl - p) + c;
                const double t = *a; *a = *b; *b = t;
            }
        __syncthreads();
        const double rp = 1.0 / ab[k * wb + kl];
        for (int r = threadIdx.x + 1; r <= km; r += 256) ab[(k + r) * wb + (kl - r)] *= rp;   // multipliers
        __syncthreads();
        for (int idx = threadIdx.x; idx < km * nc; idx += 256) {
            const int r = idx / nc + 1, c = idx % nc + 1;
            ab[(k + r) * wb + (kl - r) + c] -= ab[(k + r) * wb + (kl - r)] * ab[k * wb + kl + c];
        }
        __syncthreads();
    }
}

// dgbtrs: forward with the row exchanges, then backward; one workgroup
__global__ void __launch_bounds__(256) k_band_lu_solve(long long n, int kl, int wb, const double* __restrict__ ab,
                                                       const int* __restrict__ ipiv, double* __restrict__ x)
{
    const int ku2 = 2 * kl;
    for (long long k = 0; k < n; k++) {
        const int km = (int)min((long long)kl, n - 1 - k);
        const int pv = ipiv[k];
        if (pv < 0) {   // fixed unknown
            if (threadIdx.x == 0) x[k] = 0.0;
            __syncthreads();
            continue;
        }
        if (threadIdx.x == 0 && pv != k) { const double t = x[k]; x[k] = x[pv]; x[pv] = t; }
        __syncthreads();
        const double xk = x[k];
        for (int r = threadIdx.x + 1; r <= km; r += 256) x[k + r] -= ab[(k + r) * wb + (kl - r)] * xk;
        __syncthreads();
    }
    for (long long k = n - 1; k >= 0; k--) {
        if (ipiv[k] < 0) { __syncthreads(); continue; }
        if (threadIdx.x == 0) x[k] = x[k] / ab[k * wb + kl];
        __syncthreads();
        const double xk = x[k];
        const int up = (int)min((long long)ku2, k);
        for (int r = threadIdx.x + 1; r <= up; r += 256) x[k - r] -= ab[(k - r) * wb + kl + r] * xk;
        __syncthreads();
    }
}

// ---- the same factorisation and solve by ONE wavefront with the active window in registers (narrow bands) -------------------
// k_band_lu spends 1.7-3.5 us per column -- a pivot search by one thread, three workgroup barriers and a chain of dependent global
// accesses -- whatever the band: 28 ms for the 8,008 columns of BASELINE config 5 as one connected strip (half bandwidth 9),
// all of a Newton step's time.  A window of an LU step is small there: rows k .. k + kl, columns k .. k + 3 kl.  Lane l of one
// wavefront holds the column j = l (mod 64) of that window, one register per row (3 kl + 1 <= 64, kl <= KL): the pivot column is
// one lane's registers (v_readlane with a uniform index: no LDS, no barrier), the row exchange a register exchange, the rank-1
// update one fma per row and lane; the finished row streams out, the rows that enter the window were asked for sixteen steps
// earlier.  Same arithmetic as k_band_lu (dgbtf2: partial pivoting inside the band, a pivot below 1e-6 of the largest non-penalty
// entry fixed to zero), same storage (the two solve kernels read either's factors).
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
constexpr int WAVE_LU_DEPTH = 16;   // rows asked for ahead of their step
// (KL = the half bandwidth the band was stored with: the host rounds the true one up to the next instantiated value, the extra
// rows and columns are zeros -- a compile-time band keeps every register index static and needs no guard inside a step)
template <int KL>
__global__ void __launch_bounds__(64) k_band_lu_wave(long long n, double* __restrict__ ab, int* __restrict__ ipiv,
                                                     const double* __restrict__ amax, int* __restrict__ flag)
{
    static_assert(3 * KL + 1 <= 64, "the window of a step must fit one wavefront");
    constexpr int D = WAVE_LU_DEPTH, kl = KL, wb = 3 * KL + 1, ku2 = 2 * KL;
    const int lane = threadIdx.x;
    const double tiny = 1e-6 * (*amax);
    // entry (row i, column j) of the band, 0 outside it (and outside the matrix); j = the lane's column in the window that starts at k0
    auto entry = [&](long long i, long long k0) -> double {
        const long long j = k0 + ((lane - (int)(k0 & 63)) & 63);
        const long long o = j - i + kl;
        return (i < n && j < n && o >= 0 && o < wb) ? ab[i * wb + o] : 0.0;
    };
    double W[KL + 1];
#pragma unroll
    for (int r = 0; r <= KL; r++) W[r] = entry(r, 0);
    double nxt[D], fut[D];
    // the row that enters the window behind step k is row k + 1 + kl, seen from the window that starts at k + 1
#pragma unroll
    for (int d = 0; d < D; d++) nxt[d] = entry((long long)d + 1 + kl, (long long)d + 1);
    for (long long kb = 0; kb < n; kb += D) {
#pragma unroll
        for (int d = 0; d < D; d++) fut[d] = entry(kb + D + d + 1 + kl, kb + D + d + 1);
        auto step = [&](long long k, double entering) {
            const int c0 = (int)(k & 63);
            const int km = (int)min((long long)kl, n - 1 - k), nc = (int)min((long long)ku2, n - 1 - k);
            const int rel = (lane - c0) & 63;            // the lane's column is k + rel
            // pivot: the largest of column k's entries in rows k .. k + kl (rows beyond the matrix are zeros).  Every lane looks at
            // its own column, one lane is asked; the diagonal is the largest almost always: the search proper is the rare path
            const double a0 = fabs(W[0]);
            double mx = fabs(W[1]);
#pragma unroll
            for (int r = 2; r <= KL; r++) mx = fmax(mx, fabs(W[r]));
            int p = 0;
            double best = readlane_f64(a0, c0);
            if (readlane_f64(mx, c0) > best) {
                int pl = 0;
                double bl = a0;
#pragma unroll
                for (int r = 1; r <= KL; r++) { const double a = fabs(W[r]); if (a > bl) { bl = a; pl = r; } }
                p = __builtin_amdgcn_readlane(pl, c0);
                best = readlane_f64(bl, c0);
            }
            double lcol = 0.0;                              // what this lane stores of column k: lane r the multiplier of row k + r
            if (!(best > tiny)) {                           // null pivot (or a NaN): the unknown is fixed to zero, row and column dropped
                if (lane == 0) { ipiv[k] = -1 - (int)k; atomicAdd(flag, 1); if (isnan(best)) atomicExch(flag + 1, 1); }
                if (rel == 0) {
#pragma unroll
                    for (int r = 1; r <= KL; r++) W[r] = 0.0;
                    W[0] = 1.0;
                } else W[0] = 0.0;
            } else {
                if (lane == 0) ipiv[k] = (int)(k + p);
                if (p != 0) {                               // (uniform)
#pragma unroll
                    for (int r = 1; r <= KL; r++)
                        if (p == r) { const double t = W[0]; W[0] = W[r]; W[r] = t; }
                }
                const double rp = 1.0 / readlane_f64(W[0], c0);
#pragma unroll
                for (int r = 1; r <= KL; r++) {
                    const double l = readlane_f64(W[r], c0) * rp;
                    W[r] = fma(-l, W[0], W[r]);             // (the pivot column's own lane ends next to zero: cleared with the shift below)
                    if (lane == r) lcol = l;
                }
            }
            // row k is final: columns k .. k + nc; the multipliers of column k go to their rows
            if (rel <= nc) ab[k * wb + kl + rel] = W[0];
            if (lane >= 1 && lane <= km) ab[(k + lane) * wb + (kl - lane)] = lcol;
            // the window moves on: rows shift, the retired column's lane becomes column k + 64 (nothing there yet), row k + 1 + kl enters
#pragma unroll
            for (int r = 0; r < KL; r++) W[r] = (rel == 0) ? 0.0 : W[r + 1];
            W[KL] = entering;
        };
#pragma unroll
        for (int d = 0; d < D; d++)
            if (kb + d < n) step(kb + d, nxt[d]);
#pragma unroll
        for (int d = 0; d < D; d++) nxt[d] = fut[d];
    }
}

// dgbtrs by one wavefront: lane l holds x of the row i = l (mod 64) of the window k .. k + 63; the factor entries a step needs
// (the multipliers of column k in the forward sweep, column k of U in the backward one) are asked for sixteen steps ahead; a lane
// whose row is done keeps the result and takes its next row from a register -- results are stored and the rows after next loaded
// 64 at a time, every 64 steps
__global__ void __launch_bounds__(64) k_band_solve_wave(long long n, int kl, int wb, const double* __restrict__ ab,
                                                        const int* __restrict__ ipiv, double* __restrict__ x)
{
    constexpr int D = WAVE_LU_DEPTH;
    const int lane = threadIdx.x;
    const int ku2 = 2 * kl;
    // ---- forward: row exchanges and L
    {
        // multiplier L(i, k) for the lane's row i in the window that starts at k (0 where i is not in k+1 .. k+kl)
        auto mult = [&](long long k) -> double {
            const int rel = (lane - (int)(k & 63)) & 63;
            const long long i = k + rel;
            return (rel >= 1 && rel <= kl && i < n) ? ab[i * wb + (kl - rel)] : 0.0;
        };
        double xv = (lane < n) ? x[lane] : 0.0, xn = (64 + lane < n) ? x[64 + lane] : 0.0, xd = 0.0;
        int pv = (lane < n) ? ipiv[lane] : 0, pn = (64 + lane < n) ? ipiv[64 + lane] : 0;
        double nxt[D], fut[D];
#pragma unroll
        for (int d = 0; d < D; d++) nxt[d] = mult(d);
        for (long long kb = 0; kb < n; kb += D) {
#pragma unroll
            for (int d = 0; d < D; d++) fut[d] = mult(kb + D + d);
            auto step = [&](long long k, double mk) {
                const int c0 = (int)(k & 63);
                const int pk = __builtin_amdgcn_readlane(pv, c0);
                double xk = 0.0;                            // (a fixed unknown stays 0)
                if (pk >= 0) {
                    const int cp = pk & 63;                 // row exchange k <-> pk (pk <= k + kl: inside the window)
                    const double va = readlane_f64(xv, c0);
                    xk = readlane_f64(xv, cp);
                    if (lane == cp) xv = va;
                    xv = fma(-mk, xk, xv);                  // rows k + 1 .. k + kl
                }
                if (lane == c0) { xd = xk; xv = xn; pv = pn; }   // x[k] is final for this sweep; the lane takes row k + 64
                if (c0 == 63) {                             // (uniform) 64 rows done: out they go, the rows after next come in
                    const long long b0 = k - 63;
                    x[b0 + lane] = xd;
                    xn = (b0 + 128 + lane < n) ? x[b0 + 128 + lane] : 0.0;
                    pn = (b0 + 128 + lane < n) ? ipiv[b0 + 128 + lane] : 0;
                }
            };
#pragma unroll
            for (int d = 0; d < D; d++)
                if (kb + d < n) step(kb + d, nxt[d]);
#pragma unroll
            for (int d = 0; d < D; d++) nxt[d] = fut[d];
        }
        // the last, partial block of 64
        const long long b0 = n & ~63LL;
        if (b0 < n && b0 + lane < n) x[b0 + lane] = xd;
    }
    __builtin_amdgcn_s_waitcnt(0);   // (one wavefront: its own stores are visible to its later loads once they have completed)
    // ---- backward: U.  The window runs down: lane l holds the row i = l (mod 64) of k - 63 .. k
    {
        auto uent = [&](long long k) -> double {   // U(i, k) for the lane's row i in k - 2 kl .. k - 1; the diagonal for i = k
            if (k < 0) return 0.0;
            const int rel = ((int)(k & 63) - lane) & 63;   // i = k - rel
            const long long i = k - rel;
            return (rel <= ku2 && i >= 0) ? ab[i * wb + kl + rel] : 0.0;
        };
        const long long top = n - 1;
        const long long irow = top - (((int)(top & 63) - lane) & 63);   // the lane's row in the first window
        double xv = (irow >= 0) ? x[irow] : 0.0, xn = (irow - 64 >= 0) ? x[irow - 64] : 0.0, xd = 0.0;
        int pv = (irow >= 0) ? ipiv[irow] : 0, pn = (irow - 64 >= 0) ? ipiv[irow - 64] : 0;
        double nxt[D], fut[D];
#pragma unroll
        for (int d = 0; d < D; d++) nxt[d] = uent(top - d);
        const int cend = (int)((top + 1) & 63);            // the lane that holds the lowest row of a full window of 64
        for (long long kb = top; kb >= 0; kb -= D) {
#pragma unroll
            for (int d = 0; d < D; d++) fut[d] = uent(kb - D - d);
            auto step = [&](long long k, double uk) {
                const int c0 = (int)(k & 63);
                const int pk = __builtin_amdgcn_readlane(pv, c0);
                if (pk >= 0) {
                    const double xk = readlane_f64(xv, c0) / readlane_f64(uk, c0);
                    if (lane == c0) xv = xk;
                    else xv = fma(-uk, xk, xv);             // rows k - 1 .. k - 2 kl (0 elsewhere)
                }
                if (lane == c0) { xd = xv; xv = xn; pv = pn; }
                if (c0 == cend) {                           // (uniform) rows k .. k + 63 are done
                    const long long r1 = k + 63 - (((int)((k + 63) & 63) - lane) & 63);   // the lane's row among them
                    x[r1] = xd;
                    xn = (r1 - 128 >= 0) ? x[r1 - 128] : 0.0;
                    pn = (r1 - 128 >= 0) ? ipiv[r1 - 128] : 0;
                }
            };
#pragma unroll
            for (int d = 0; d < D; d++)
                if (kb - d >= 0) step(kb - d, nxt[d]);
#pragma unroll
            for (int d = 0; d < D; d++) nxt[d] = fut[d];
        }
        // the rows of the last, partial window (row 0 upward, fewer than 64 of them)
        const long long rem = (top + 1) % 64;
        if (rem) {
            const long long r1 = (rem - 1) - (((int)((rem - 1) & 63) - lane) & 63);
            if (r1 >= 0) x[r1] = xd;
        }
    }
}

// ---- narrow bands with many columns: block cyclic reduction ----------------------------------------------------------------
// A band of half bandwidth <= B is block tridiagonal in blocks of B rows: L_i x_{i-1} + D_i x_i + U_i x_{i+1} = r_i.  One level of
// cyclic reduction removes every second active block -- x_j = D_j^-1 (r_j - L_j x_{j-s} - U_j x_{j+s}) put into the rows of its two
// neighbours -- and leaves a block tridiagonal system of half the size and twice the stride; log2(nb) levels down one block is
// left, log2(nb) levels up every x_j follows from its two neighbours.  Every level is a few hundred independent B x B problems:
// one wavefront each (lane = column of [D | L U r], one register per row, Gauss-Jordan with partial pivoting INSIDE the block,
// v_readlane for the pivot column), about 3 log2(nb) small launches instead of the chain of n dependent column steps of an LU
// (config 5 as a connected strip, n = 8,008, B = 13: 0.3 ms against 5.6 + 3.2 ms for the band LU and its solve by one wavefront).
// What it gives up is pivoting ACROSS blocks: exact for the matrices it is meant for (Mat_K with its penalties plus the cohesive
// tangents: every reduced diagonal block stays a well-conditioned Schur complement), not for any band matrix.  So it is used
// only where set-up has run the reduction of the matrix once without meeting a small pivot, and every solve checks the
// componentwise backward error of its answer (max_i |b - A x|_i / (|A| |x| + |b|)_i <= 1e-10) -- a solve that fails either
// test goes to the band LU, for this matrix from then on.  FEM_AMD_LU_BCR=0 switches it off.
template <int B>
__global__ void __launch_bounds__(256) k_bcr_blocks(long long n, int nbk, const double* __restrict__ ab, const double* __restrict__ amax,
                                                    double* __restrict__ L, double* __restrict__ D, double* __restrict__ U)
{
    constexpr int kl = B, wb = 3 * B + 1, BB = B * B;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nbk * BB) return;
    const int I = (int)(t / BB), r = (int)(t % BB) / B, c = (int)(t % BB) % B;
    const long long i = (long long)I * B + r;
    auto at = [&](long long j) -> double {     // A(i, j), 0 outside the band and the matrix
        const long long o = j - i + kl;
        return (i < n && j >= 0 && j < n && o >= 0 && o < wb) ? ab[i * wb + o] : 0.0;
    };
    const long long j0 = (long long)I * B + c;
    L[t] = at(j0 - B);
    U[t] = at(j0 + B);
    D[t] = (i < n) ? at(j0) : (r == c ? *amax : 0.0);   // (rows past the matrix in the last block: a diagonal of the matrix's own scale)
}
// one wavefront per eliminated block j = j0 + t jstep: [L_j U_j r_j] <- D_j^-1 [L_j U_j r_j]
template <int B>
__global__ void __launch_bounds__(64) k_bcr_eliminate(int count, int j0, int jstep, double* __restrict__ L, const double* __restrict__ D,
                                                      double* __restrict__ U, double* __restrict__ rv, const double* __restrict__ amax,
                                                      int* __restrict__ flag)
{
    constexpr int BB = B * B;
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= count) return;
    const size_t j = (size_t)j0 + (size_t)t * jstep;
    const double tiny = 1e-6 * (*amax);
    // lane c: column c of [D | L | U | r]
    const int kind = lane < B ? 0 : lane < 2 * B ? 1 : lane < 3 * B ? 2 : lane == 3 * B ? 3 : 4, cc = lane - B * (kind < 3 ? kind : 3);
    double M[B];
#pragma unroll
    for (int r = 0; r < B; r++)
        M[r] = kind == 0 ? D[j * BB + r * B + cc] : kind == 1 ? L[j * BB + r * B + cc] : kind == 2 ? U[j * BB + r * B + cc] : kind == 3 ? rv[j * B + r] : 0.0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < B; k++) {
        int pl = k;
        double bl = fabs(M[k]);
#pragma unroll
        for (int r = k + 1; r < B; r++) { const double a = fabs(M[r]); if (a > bl) { bl = a; pl = r; } }
        const int p = __builtin_amdgcn_readlane(pl, k);
        const double best = readlane_f64(bl, k);
        if (!(best > tiny)) bad = true;                    // (uniform) a small pivot or a NaN: the caller falls back to the band LU
        if (p != k) {
#pragma unroll
            for (int r = k + 1; r < B; r++)
                if (p == r) { const double tt = M[k]; M[k] = M[r]; M[r] = tt; }
        }
        const double rp = 1.0 / readlane_f64(M[k], k);
        M[k] *= rp;
#pragma unroll
        for (int r = 0; r < B; r++)
            if (r != k) { const double f = readlane_f64(M[r], k); M[r] = fma(-f, M[k], M[r]); }
    }
    if (bad && lane == 0) atomicAdd(flag, 1);
#pragma unroll
    for (int r = 0; r < B; r++) {
        if (kind == 1) L[j * BB + r * B + cc] = M[r];
        else if (kind == 2) U[j * BB + r * B + cc] = M[r];
        else if (kind == 3) rv[j * B + r] = M[r];
    }
}
// one wavefront per surviving block i = t istep: its rows take in the two eliminated neighbours i - s, i + s
template <int B>
__global__ void __launch_bounds__(64) k_bcr_update(int count, int istep, int s, int nbk, double* __restrict__ L, double* __restrict__ D,
                                                   double* __restrict__ U, double* __restrict__ rv)
{
    constexpr int BB = B * B;
    __shared__ double Ls[BB], Us[BB];
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= count) return;
    const long long i = (long long)t * istep, jl = i - s, jr = i + s;
    const bool hl = jl >= 0, hr = jr < nbk;
    for (int e = lane; e < BB; e += 64) { Ls[e] = hl ? L[(size_t)i * BB + e] : 0.0; Us[e] = hr ? U[(size_t)i * BB + e] : 0.0; }
    const int kind = lane < B ? 0 : lane < 2 * B ? 1 : lane < 3 * B ? 2 : lane == 3 * B ? 3 : 4, cc = lane - B * (kind < 3 ? kind : 3);
    double Wl[B], Wr[B], out[B];
#pragma unroll
    for (int q = 0; q < B; q++) {
        // from the left neighbour: D^-1 U (into D_i), D^-1 L (the new L_i), D^-1 r; from the right one: D^-1 L (into D_i), D^-1 U (the new U_i), D^-1 r
        Wl[q] = !hl ? 0.0 : kind == 0 ? U[(size_t)jl * BB + q * B + cc] : kind == 1 ? L[(size_t)jl * BB + q * B + cc] : kind == 3 ? rv[(size_t)jl * B + q] : 0.0;
        Wr[q] = !hr ? 0.0 : kind == 0 ? L[(size_t)jr * BB + q * B + cc] : kind == 2 ? U[(size_t)jr * BB + q * B + cc] : kind == 3 ? rv[(size_t)jr * B + q] : 0.0;
        out[q] = kind == 0 ? D[(size_t)i * BB + q * B + cc] : kind == 3 ? rv[(size_t)i * B + q] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < B; r++) {
        double a = out[r];
#pragma unroll
        for (int q = 0; q < B; q++) a = fma(-Ls[r * B + q], Wl[q], fma(-Us[r * B + q], Wr[q], a));
        out[r] = a;
    }
#pragma unroll
    for (int r = 0; r < B; r++) {
        if (kind == 0) D[(size_t)i * BB + r * B + cc] = out[r];
        else if (kind == 1) L[(size_t)i * BB + r * B + cc] = out[r];
        else if (kind == 2) U[(size_t)i * BB + r * B + cc] = out[r];
        else if (kind == 3) rv[(size_t)i * B + r] = out[r];
    }
}
// the way up: x_j = (D^-1 r)_j - (D^-1 L)_j x_{j-s} - (D^-1 U)_j x_{j+s}, one wavefront per block eliminated at this stride
template <int B>
__global__ void __launch_bounds__(64) k_bcr_back(int count, int j0, int jstep, int s, int nbk, const double* __restrict__ L,
                                                 const double* __restrict__ U, double* __restrict__ rv)
{
    constexpr int BB = B * B;
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= count) return;
    const long long j = (long long)j0 + (long long)t * jstep, jl = j - s, jr = j + s;
    const double xl = (lane < B && jl >= 0) ? rv[(size_t)jl * B + lane] : 0.0, xr = (lane < B && jr < nbk) ? rv[(size_t)jr * B + lane] : 0.0;
    const int r = lane < B ? lane : 0;
    double a = rv[(size_t)j * B + r];
#pragma unroll
    for (int q = 0; q < B; q++)
        a = fma(-L[(size_t)j * BB + r * B + q], readlane_f64(xl, q), fma(-U[(size_t)j * BB + r * B + q], readlane_f64(xr, q), a));
    if (lane < B) rv[(size_t)j * B + lane] = a;
}
// ---- the same reduction as ONE resident launch (round 6) -------------------------------------------------------------------------
// The launches above are a chain of 3 log2(nb) + 1 kernels of a few hundred wavefronts each, 5-20 us apiece for 1-2 us of arithmetic
// (config 5's connected strip: 52 launches and 0.6 ms per reduction, ten reductions per run).  Here the levels are loops of one
// kernel with a grid barrier (grid_barrier.h) between them, and a level is ONE step instead of two: the wavefront of a surviving block
// eliminates BOTH its neighbours itself ([D L U r] of a block in 3 B + 1 lanes, one register per row, as k_bcr_eliminate -- the two
// eliminations run side by side in the same instruction stream, so the second one costs issue slots, not latency) and takes them
// into its own rows at once; the elimination of the right-hand neighbour is what it stores for the way up.  Level 0 reads the
// pristine blocks, so no work copies are made.  2 log2(nb) + 1 barriers per solve.
// (partial pivoting inside the block as k_bcr_eliminate does it -- the largest entry of column k among rows k .. B-1, the first one on
// a tie -- but the search for its row runs only where the diagonal entry is not already that largest one: a maximum over the column
// (B - k instructions) and one uniform branch per step instead of three selects per row)
template <int B, int K>
__device__ __forceinline__ void bcr_swap_in_pivot(double (&M)[B])
{
    int pl = K;
    double bl = fabs(M[K]);
#pragma unroll
    for (int r = K + 1; r < B; r++) { const double a = fabs(M[r]); if (a > bl) { bl = a; pl = r; } }
    const int p = __builtin_amdgcn_readlane(pl, K);
#pragma unroll
    for (int r = K + 1; r < B; r++)
        if (p == r) { const double t = M[K]; M[K] = M[r]; M[r] = t; }
}
template <int B, int K>
__device__ __forceinline__ bool bcr_gj_step(double (&Ml)[B], double (&Mr)[B], double tiny)
{
    double ml = 0.0, mr = 0.0;
#pragma unroll
    for (int r = K + 1; r < B; r++) { ml = fmax(ml, fabs(Ml[r])); mr = fmax(mr, fabs(Mr[r])); }
    if (readlane_f64(ml, K) > readlane_f64(fabs(Ml[K]), K)) bcr_swap_in_pivot<B, K>(Ml);   // (uniform; rare for the matrices this is used on)
    if (readlane_f64(mr, K) > readlane_f64(fabs(Mr[K]), K)) bcr_swap_in_pivot<B, K>(Mr);
    const double pvl = readlane_f64(Ml[K], K), pvr = readlane_f64(Mr[K], K);
    const bool bad = !(fabs(pvl) > tiny) || !(fabs(pvr) > tiny);
    const double rl = 1.0 / pvl, rr = 1.0 / pvr;
    Ml[K] *= rl; Mr[K] *= rr;
#pragma unroll
    for (int r = 0; r < B; r++)
        if (r != K) {
            const double fl = readlane_f64(Ml[r], K), fr = readlane_f64(Mr[r], K);
            Ml[r] = fma(-fl, Ml[K], Ml[r]);
            Mr[r] = fma(-fr, Mr[K], Mr[r]);
        }
    if constexpr (K + 1 < B) return bcr_gj_step<B, K + 1>(Ml, Mr, tiny) || bad;
    else return bad;
}
template <int B>
__device__ __forceinline__ bool bcr_gj_pair(double (&Ml)[B], double (&Mr)[B], double tiny)
{
    return bcr_gj_step<B, 0>(Ml, Mr, tiny);
}
// One block's step of a level: block i takes in its neighbours jl = i - s and jr = i + s (hl, hr: they exist), read from `src`; `out`
// = the lane's column of the new [D_i L_i U_i r_i].  STORE_E: D^-1 [L U r] of the right-hand neighbour goes to EL / EU / z.
struct BcrSrc { const double *L, *D, *U, *r; };
template <int B, bool STORE_E>
__device__ __forceinline__ bool bcr_level_block(const BcrSrc& src, long long i, long long jl, long long jr, bool hl, bool hr, int lane,
                                                int kind, int cc, double am, double tiny, double (&out)[B], double* EL, double* EU,
                                                double* z)
{
    constexpr int BB = B * B;
    // one pointer and one stride per lane (its column of D, L or U, or the right-hand side): one load per row and block
    const size_t ocol = kind == 3 ? 0 : (size_t)cc;
    const int sd = kind == 3 ? 1 : B;
    const double* bD = kind == 0 ? src.D : kind == 1 ? src.L : kind == 2 ? src.U : src.r;
    const size_t bsz = kind == 3 ? (size_t)B : (size_t)BB;
    const double *pl = bD + (size_t)(hl ? jl : i) * bsz + ocol, *pr = bD + (size_t)(hr ? jr : i) * bsz + ocol, *po = bD + (size_t)i * bsz + ocol;
    // the block's own L_i (lanes 0 .. B-1) and U_i (lanes B .. 2B-1), a column per lane: their entries reach the update below as
    // uniform values (v_readlane), not through the LDS -- the compiler hoists every LDS read of such a loop in front of it
    // (2 B^2 registers: k_bcr_update<13> spills 800 bytes per lane for that)
    const double* pown = (lane < B ? src.L : src.U) + (size_t)i * BB + (lane < B ? lane : lane - B);
    const bool hown = lane < B ? hl : (lane < 2 * B && hr);
    double Ml[B], Mr[B], own[B];
#pragma unroll
    for (int q = 0; q < B; q++) {
        const double idn = (kind == 0 && q == cc) ? am : 0.0;   // (no neighbour: a diagonal of the matrix's scale goes through the elimination)
        const double vl = kind < 4 ? pl[q * sd] : 0.0, vr = kind < 4 ? pr[q * sd] : 0.0, vo = (kind == 0 || kind == 3) ? po[q * sd] : 0.0;
        Ml[q] = hl ? vl : idn;
        Mr[q] = hr ? vr : idn;
        out[q] = vo;
        own[q] = (lane < 2 * B && hown) ? pown[q * B] : 0.0;
    }
    const bool bad = bcr_gj_pair<B>(Ml, Mr, tiny);
    if constexpr (STORE_E) {
        if (hr && kind >= 1 && kind <= 3) {   // what the way up needs of the right-hand neighbour
            double* pe = (kind == 1 ? EL : kind == 2 ? EU : z) + (size_t)jr * bsz + ocol;
#pragma unroll
            for (int q = 0; q < B; q++) pe[q * sd] = Mr[q];
        }
    }
    // the columns the rows of block i take in: D_i <- D_i - L_i (D^-1 U)_jl - U_i (D^-1 L)_jr, L_i <- -L_i (D^-1 L)_jl,
    // U_i <- -U_i (D^-1 U)_jr, r_i <- r_i - L_i (D^-1 r)_jl - U_i (D^-1 r)_jr
#pragma unroll
    for (int q = 0; q < B; q++) {
        const double ul = __shfl(Ml[q], 2 * B + cc), lr = __shfl(Mr[q], B + cc);
        Ml[q] = !hl ? 0.0 : kind == 0 ? ul : (kind == 1 || kind == 3) ? Ml[q] : 0.0;
        Mr[q] = !hr ? 0.0 : kind == 0 ? lr : (kind == 2 || kind == 3) ? Mr[q] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < B; r++) {
        double a = out[r];
#pragma unroll
        for (int q = 0; q < B; q++) a = fma(-readlane_f64(own[r], q), Ml[q], fma(-readlane_f64(own[r], B + q), Mr[q], a));
        out[r] = a;
    }
    return bad;
}
// the lane's column of [D L U r] of block i to the arrays of `dst`
template <int B>
__device__ __forceinline__ void bcr_store_block(double* L, double* D, double* U, double* r, long long i, int kind, int cc, const double (&out)[B])
{
    if (kind < 4) {
        double* pw = kind == 3 ? r + (size_t)i * B : (kind == 0 ? D : kind == 1 ? L : U) + (size_t)i * B * B + cc;
        const int sd = kind == 3 ? 1 : B;
#pragma unroll
        for (int q = 0; q < B; q++) pw[q * sd] = out[q];
    }
}
// L, D, U: work copies (written and read again inside the launch: no __restrict__, no const -- nothing of them may go through the
// scalar cache); EL, EU, z: D^-1 [L U r] of every eliminated block; rv: right-hand side by blocks in, solution out
template <int B>
__global__ void __launch_bounds__(256) k_bcr_resident(int nbk, const double* __restrict__ L0, const double* __restrict__ D0,
                                                      const double* __restrict__ U0, double* L, double* D, double* U, double* EL,
                                                      double* EU, double* z, double* rv, const double* __restrict__ amax, int* flag,
                                                      GridBarrier bar)
{
    static_assert(3 * B + 1 <= 64, "[D L U r] of a block must fit one wavefront");
    constexpr int BB = B * B;
    __shared__ unsigned shu[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wv), nw = (int)gridDim.x * 4;
    GridBarrierLane gl = grid_barrier_census(bar, gridDim.x, shu);
    const double am = *amax, tiny = 1e-6 * am;
    const int kind = lane < B ? 0 : lane < 2 * B ? 1 : lane < 3 * B ? 2 : lane == 3 * B ? 3 : 4, cc = lane - B * (kind < 3 ? kind : 3);
    bool bad = false;
    int smax = 0;
    for (int st = 1; st < nbk; st *= 2) {
        const BcrSrc src{st == 1 ? L0 : L, st == 1 ? D0 : D, st == 1 ? U0 : U, rv};
        const int count = (nbk + 2 * st - 1) / (2 * st);
        for (int t = gw; t < count; t += nw) {
            const long long i = (long long)t * 2 * st, jl = i - st, jr = i + st;
            double out[B];
            bad |= bcr_level_block<B, true>(src, i, jl, jr, jl >= 0, jr < nbk, lane, kind, cc, am, tiny, out, EL, EU, z);
            bcr_store_block<B>(L, D, U, rv, i, kind, cc, out);
        }
        smax = st;
        if (!grid_barrier(bar, gl, shu)) return;
    }
    if (gw == 0) {   // the one block left: x_0 = D_0^-1 r_0
        const double* sD = nbk == 1 ? D0 : D;
        double Ml[B], Mr[B];
#pragma unroll
        for (int q = 0; q < B; q++) {
            Ml[q] = kind == 0 ? sD[q * B + cc] : kind == 3 ? rv[q] : 0.0;
            Mr[q] = (kind == 0 && q == cc) ? am : 0.0;
        }
        bad |= bcr_gj_pair<B>(Ml, Mr, tiny);
        if (kind == 3) {
#pragma unroll
            for (int q = 0; q < B; q++) rv[q] = Ml[q];
        }
    }
    if (bad && lane == 0) atomicAdd(flag, 1);
    for (int st = smax; st >= 1; st /= 2) {
        if (!grid_barrier(bar, gl, shu)) return;
        const int count = (nbk - st + 2 * st - 1) / (2 * st);
        for (int t = gw; t < count; t += nw) {
            const long long j = (long long)st + (long long)t * 2 * st, jl = j - st, jr = j + st;
            const double xl = (lane < B) ? rv[(size_t)jl * B + lane] : 0.0, xr = (lane < B && jr < nbk) ? rv[(size_t)jr * B + lane] : 0.0;
            const int r = lane < B ? lane : 0;
            double a = z[(size_t)j * B + r];
#pragma unroll
            for (int q = 0; q < B; q++)
                a = fma(-EL[(size_t)j * BB + r * B + q], readlane_f64(xl, q), fma(-EU[(size_t)j * BB + r * B + q], readlane_f64(xr, q), a));
            if (lane < B) rv[(size_t)j * B + lane] = a;
        }
    }
}
// Few blocks (one wavefront each fits the chip): PARALLEL cyclic reduction -- every block takes in its neighbours at every level, so
// that after log2(nb) levels each one stands alone and is solved where it is: no way up, half the barriers (config 5's strip, 616
// blocks: 10 instead of 21).  log2(nb) times the arithmetic of the reduction above, which is why it is for few blocks only.  Levels
// alternate between two sets of arrays (A, Bq): a level reads what every block of the level before wrote.  x: the solution by blocks.
template <int B>
__global__ void __launch_bounds__(256) k_pcr_resident(int nbk, const double* __restrict__ L0, const double* __restrict__ D0,
                                                      const double* __restrict__ U0, const double* __restrict__ r0, double* LA, double* DA,
                                                      double* UA, double* rA, double* LB, double* DB, double* UB, double* rB, double* x,
                                                      const double* __restrict__ amax, int* flag, GridBarrier bar)
{
    static_assert(3 * B + 1 <= 64, "[D L U r] of a block must fit one wavefront");
    __shared__ unsigned shu[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wv), nw = (int)gridDim.x * 4;
    GridBarrierLane gl = grid_barrier_census(bar, gridDim.x, shu);
    const double am = *amax, tiny = 1e-6 * am;
    const int kind = lane < B ? 0 : lane < 2 * B ? 1 : lane < 3 * B ? 2 : lane == 3 * B ? 3 : 4, cc = lane - B * (kind < 3 ? kind : 3);
    bool bad = false;
    int lvl = 0;
    for (int st = 1;; st *= 2, lvl++) {
        const bool last = st >= nbk;   // (no neighbour left at this distance: the blocks stand alone)
        const BcrSrc src = lvl == 0 ? BcrSrc{L0, D0, U0, r0} : (lvl & 1) ? BcrSrc{LA, DA, UA, rA} : BcrSrc{LB, DB, UB, rB};
        for (int i = gw; i < nbk; i += nw) {
            double out[B];
            if (!last) {
                const long long jl = (long long)i - st, jr = (long long)i + st;
                bad |= bcr_level_block<B, false>(src, i, jl, jr, jl >= 0, jr < nbk, lane, kind, cc, am, tiny, out, nullptr, nullptr, nullptr);
            }
            if (last || 2 * st >= nbk) {   // x_i = D_i^-1 r_i, from the registers where the level has just been taken in
                double Ml[B], Mr[B];
#pragma unroll
                for (int q = 0; q < B; q++) {
                    const double own = last ? (kind == 0 ? src.D[(size_t)i * B * B + q * B + cc] : kind == 3 ? src.r[(size_t)i * B + q] : 0.0) : out[q];
                    Ml[q] = (kind == 0 || kind == 3) ? own : 0.0;
                    Mr[q] = (kind == 0 && q == cc) ? am : 0.0;
                }
                bad |= bcr_gj_pair<B>(Ml, Mr, tiny);
                if (kind == 3) {
#pragma unroll
                    for (int q = 0; q < B; q++) x[(size_t)i * B + q] = Ml[q];
                }
            } else if (lvl & 1) bcr_store_block<B>(LB, DB, UB, rB, i, kind, cc, out);
            else bcr_store_block<B>(LA, DA, UA, rA, i, kind, cc, out);
        }
        if (last || 2 * st >= nbk) break;
        if (!grid_barrier(bar, gl, shu)) return;
    }
    if (bad && lane == 0) atomicAdd(flag, 1);
}
// what a solve by the resident reduction needs in front of it, in one launch: the right-hand side in the band's numbering (twice: the
// reduction overwrites one, the backward-error test reads the other), zeros in the rows past the matrix, the barrier's words and the
// status words behind them cleared
__global__ void __launch_bounds__(256) k_bcr_prep(size_t n, size_t npad, const int* __restrict__ perm, const double* __restrict__ b,
                                                  double* __restrict__ rv, double* __restrict__ keep, unsigned* __restrict__ words, int nwords)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const double v = b[i]; const int p = perm[i]; rv[p] = v; keep[p] = v; }
    else if (i < npad) rv[i] = 0.0;
    if (i < (size_t)nwords) words[i] = 0u;
}
// componentwise backward error of x for the band matrix (renumbered system): max_i |b - A x|_i / (|A| |x| + |b|)_i
__global__ void __launch_bounds__(256) k_band_backward_error(long long n, int kl, int wb, const double* __restrict__ ab,
                                                             const double* __restrict__ x, const double* __restrict__ b,
                                                             unsigned long long* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0;
    if (i < n) {
        double sum = b[i], den = fabs(b[i]);
        for (int o = 0; o < wb; o++) {
            const long long j = i - kl + o;
            if (j < 0 || j >= n) continue;
            const double a = ab[i * wb + o];
            sum = fma(-a, x[j], sum); den = fma(fabs(a), fabs(x[j]), den);
        }
        e = (den > 0.0) ? fabs(sum) / den : 0.0;
        if (isnan(e)) e = 1.0;
    }
    for (int o = 32; o > 0; o >>= 1) e = fmax(e, __shfl_down(e, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(e));
}

// ---- a matrix that falls apart into small independent blocks (BASELINE config 5: a thousand copies of the reference's
// cohesive deck side by side share no node) is factorised block by block: one wavefront per connected component of the node
// graph, dense LU with partial pivoting in LDS -- instead of one workgroup walking all 16,000 columns of the band one after the
// other (28 ms per factorisation, most of a Newton step's time).  Same pivot rule as the band LU (a pivot below 1e-6 of
// the largest non-penalty entry of the whole matrix is fixed to zero, a NaN is reported).
constexpr int COMP_MAX = 64;   // dofs of a component (one lane per row)
__global__ void __launch_bounds__(256) k_val_amax(size_t n, const double* __restrict__ v, unsigned long long* __restrict__ out)
{
    double m = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const double a = fabs(v[i]); if (a < 1e29 && a > m) m = a; }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));   // (non-negative doubles order like their bits)
}
template <int BS>
__global__ void __launch_bounds__(64) k_comp_lu(int ncomp, const int* __restrict__ comp_ptr, const int* __restrict__ comp_nodes,
                                                const int* __restrict__ node_loc, const int* __restrict__ boff,
                                                const int* __restrict__ bcol, const double* __restrict__ val,
                                                const int* __restrict__ rowlen, const unsigned long long* __restrict__ amax_bits,
                                                const long long* __restrict__ fac_off, double* __restrict__ fac, int* __restrict__ piv,
                                                int* __restrict__ flag)
{
    __shared__ double a[COMP_MAX][COMP_MAX + 1];
    __shared__ int s_p;
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= ncomp) return;
    const int p0 = comp_ptr[c], m = (comp_ptr[c + 1] - p0) * BS;
    const double tiny = 1e-6 * __longlong_as_double((long long)*amax_bits);
    for (int j = 0; j < m; j++) a[lane][j] = 0.0;
    __syncthreads();
    if (lane < m) {   // row `lane` of the block from the node's block row of the Mat
        const int node = comp_nodes[p0 + lane / BS], r = lane % BS;
        const int ln = node & 63, len = rowlen[node], o0 = boff[node >> 6];
        for (int j = 0; j < len; j++) {
            const int cl = node_loc[bcol[(size_t)(o0 + j) * 64 + ln]] * BS;
            for (int cc = 0; cc < BS; cc++) a[lane][cl + cc] = val[((size_t)(o0 + j) * BS * BS + r * BS + cc) * 64 + ln];
        }
    }
    __syncthreads();
    const size_t dof0 = (size_t)p0 * BS;   // position of the block's first dof in the component-ordered numbering
    for (int k = 0; k < m; k++) {
        if (lane == 0) {
            int p = k;
            double best = fabs(a[k][k]);
            for (int i = k + 1; i < m; i++) { const double v = fabs(a[i][k]); if (v > best) { best = v; p = i; } }
            if (!(best > tiny)) {
                p = -1;
                atomicAdd(flag, 1);
                if (isnan(best)) atomicExch(flag + 1, 1);
            }
            s_p = p;
            piv[dof0 + k] = p;
        }
        __syncthreads();
        const int p = s_p;
        if (p < 0) {   // null pivot: the unknown is fixed to zero (row and column dropped)
            if (lane < m && lane != k) { a[lane][k] = 0.0; a[k][lane] = 0.0; }
            if (lane == 0) a[k][k] = 1.0;
            __syncthreads();
            continue;
        }
        // only the active part of the two rows changes places (as k_band_lu and LAPACK's dgbtf2 do): the multipliers already stored in
        // columns < k stay where the step that made them put them, which is what k_comp_solve's interleaved exchanges assume
        if (p != k && lane >= k && lane < m) { const double t = a[k][lane]; a[k][lane] = a[p][lane]; a[p][lane] = t; }
        __syncthreads();
        if (lane > k && lane < m) {
            const double l = a[lane][k] / a[k][k];
            a[lane][k] = l;
            for (int j = k + 1; j < m; j++) a[lane][j] -= l * a[k][j];
        }
        __syncthreads();
    }
    if (lane < m)
        for (int j = 0; j < m; j++) fac[fac_off[c] + (size_t)lane * m + j] = a[lane][j];
}
// the solves of the same blocks: b and x in the Mat's dof numbering
template <int BS>
__global__ void __launch_bounds__(64) k_comp_solve(int ncomp, const int* __restrict__ comp_ptr, const int* __restrict__ comp_nodes,
                                                   const long long* __restrict__ fac_off, const double* __restrict__ fac,
                                                   const int* __restrict__ piv, const double* __restrict__ b, double* __restrict__ x)
{
    __shared__ double xs[COMP_MAX];
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= ncomp) return;
    const int p0 = comp_ptr[c], m = (comp_ptr[c + 1] - p0) * BS;
    const size_t dof0 = (size_t)p0 * BS;
    const double* f = fac + fac_off[c];
    size_t gdof = 0;
    if (lane < m) { gdof = (size_t)comp_nodes[p0 + lane / BS] * BS + lane % BS; xs[lane] = b[gdof]; }
    __syncthreads();
    for (int k = 0; k < m; k++) {   // row exchanges and L
        const int p = piv[dof0 + k];
        if (p < 0) { if (lane == 0) xs[k] = 0.0; __syncthreads(); continue; }
        if (lane == 0 && p != k) { const double t = xs[k]; xs[k] = xs[p]; xs[p] = t; }
        __syncthreads();
        const double xk = xs[k];
        if (lane > k && lane < m) xs[lane] -= f[(size_t)lane * m + k] * xk;
        __syncthreads();
    }
    for (int k = m - 1; k >= 0; k--) {   // U
        if (piv[dof0 + k] < 0) { __syncthreads(); continue; }
        if (lane == 0) xs[k] /= f[(size_t)k * m + k];
        __syncthreads();
        const double xk = xs[k];
        if (lane < k) xs[lane] -= f[(size_t)lane * m + k] * xk;
        __syncthreads();
    }
    if (lane < m) x[gdof] = xs[lane];
}

void BandLU::setup(fem_mat A_)
{
    A = A_; bs = A->bs;
    const int nn = A->nbrows;
    n = (long long)nn * bs;
    FEM_CHECK(!(A->mesh->halo && A->mesh->halo->comm->size > 1), 9, "pc lu: the direct solve is single-domain");
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const bool comp_env = !(std::getenv("FEM_AMD_LU_COMPONENTS") && std::atoi(std::getenv("FEM_AMD_LU_COMPONENTS")) == 0);   // (read at every set-up: the tests switch it)
    // the host-side symbolic work depends on the sparsity alone: done once per pattern (a Newton loop comes here at every iteration)
    if (!A->pat->lu_sym || A->pat->lu_sym->comp_env != comp_env) {
        auto sym = std::make_shared<LuSymbolic>();
        sym->comp_env = comp_env;
        std::vector<int> hc((size_t)A->padded_blocks);
        A->bcol.download(hc.data(), hc.size(), s);
        FEM_HIP(hipStreamSynchronize(s));
        std::vector<int> ptr(nn + 1, 0), adj;
        for (int v = 0; v < nn; v++) ptr[v + 1] = ptr[v] + A->h_rowlen[v];
        adj.resize((size_t)ptr[nn]);
        for (int v = 0; v < nn; v++) {
            const size_t o0 = (size_t)A->h_boff[v >> 6];
            for (int j = 0; j < A->h_rowlen[v]; j++) adj[(size_t)ptr[v] + j] = hc[(o0 + j) * 64 + (v & 63)];
        }
        // independent blocks?  connected components of the node graph (the pattern is symmetric), small ones only
        {
            std::vector<int> comp((size_t)nn, -1), stack, cptr{0}, cnodes;
            int nc = 0, biggest = 0;
            for (int v0_ = 0; v0_ < nn && comp_env; v0_++) {
                if (comp[(size_t)v0_] >= 0) continue;
                const size_t first = cnodes.size();
                stack.assign(1, v0_); comp[(size_t)v0_] = nc;
                while (!stack.empty()) {
                    const int v = stack.back(); stack.pop_back();
                    cnodes.push_back(v);
                    for (int e = ptr[v]; e < ptr[v + 1]; e++) {
                        const int u = adj[(size_t)e];
                        if (comp[(size_t)u] < 0) { comp[(size_t)u] = nc; stack.push_back(u); }
                    }
                }
                std::sort(cnodes.begin() + (long)first, cnodes.end());
                biggest = std::max(biggest, (int)(cnodes.size() - first));
                cptr.push_back((int)cnodes.size());
                nc++;
            }
            if (comp_env && nc >= 1 && biggest * bs <= COMP_MAX) {   // (also a single small deck: the same arithmetic as its tilings)
                sym->batched = true;
                sym->nc = nc;
                sym->loc.resize((size_t)nn);
                sym->foff.assign((size_t)nc + 1, 0);
                for (int c = 0; c < nc; c++) {
                    for (int k = cptr[c]; k < cptr[c + 1]; k++) sym->loc[(size_t)cnodes[(size_t)k]] = k - cptr[c];
                    const long long mm = (long long)(cptr[c + 1] - cptr[c]) * bs;
                    sym->foff[(size_t)c + 1] = sym->foff[(size_t)c] + mm * mm;
                }
                sym->cptr = std::move(cptr); sym->cnodes = std::move(cnodes);
            }
        }
        if (!sym->batched) {
            sym->perm = rcm_order(nn, ptr, adj);
            long long bwn = 0, bw_nat = 0;
            for (int v = 0; v < nn; v++)
                for (int e = ptr[v]; e < ptr[v + 1]; e++) {
                    bwn = std::max(bwn, (long long)std::abs(sym->perm[v] - sym->perm[adj[e]]));
                    bw_nat = std::max(bw_nat, (long long)std::abs(v - adj[e]));
                }
            if (bw_nat <= bwn) { for (int v = 0; v < nn; v++) sym->perm[v] = v; bwn = bw_nat; }
            sym->bwn = bwn;
        }
        A->pat->lu_sym = sym;
    }
    LuSymbolic& sym = *A->pat->lu_sym;
    batched = sym.batched;
    if (batched) {
        const int nc = sym.nc;
        ncomp = nc;
        if (sym.d_cptr.n != sym.cptr.size()) {   // (once per sparsity: four uploads and their waits in front of every Jacobian's factorisation otherwise)
            sym.d_cptr.alloc(sym.cptr.size()); sym.d_cptr.upload(sym.cptr.data(), sym.cptr.size(), s);
            sym.d_cnodes.alloc(sym.cnodes.size()); sym.d_cnodes.upload(sym.cnodes.data(), sym.cnodes.size(), s);
            sym.d_loc.alloc(sym.loc.size()); sym.d_loc.upload(sym.loc.data(), sym.loc.size(), s);
            sym.d_foff.alloc(sym.foff.size()); sym.d_foff.upload(sym.foff.data(), sym.foff.size(), s);
        }
        comp_ptr = sym.d_cptr.p; comp_nodes = sym.d_cnodes.p; node_loc = sym.d_loc.p; fac_off = sym.d_foff.p;
        fac.alloc((size_t)std::max<long long>(sym.foff[(size_t)nc], 1));
        ipiv.alloc((size_t)n); flag.alloc(2); flag.zero(s);
        DBuf<unsigned long long> amax(1);
        amax.zero(s);
        hipLaunchKernelGGL(k_val_amax, dim3(std::min<unsigned>(cdiv(A->val.n, 256), 128)), dim3(256), 0, s, A->val.n, A->values(), amax.p);   // (one atomic per wavefront on one word: 4096 of them took 47 us)
        if (bs == 3)
            hipLaunchKernelGGL(k_comp_lu<3>, dim3(nc), dim3(64), 0, s, nc, comp_ptr, comp_nodes, node_loc, A->boff.p, A->bcol.p,
                               A->values(), A->rowlen.p, amax.p, fac_off, fac.p, ipiv.p, flag.p);
        else
            hipLaunchKernelGGL(k_comp_lu<2>, dim3(nc), dim3(64), 0, s, nc, comp_ptr, comp_nodes, node_loc, A->boff.p, A->bcol.p,
                               A->values(), A->rowlen.p, amax.p, fac_off, fac.p, ipiv.p, flag.p);
        FEM_HIP(hipGetLastError());
        int hflag[2] = {0, 0};
        flag.download(hflag, 2, s);
        FEM_HIP(hipStreamSynchronize(s));
        null_pivots = hflag[0];
        failed = hflag[1] != 0;
        return;
    }
    const std::vector<int>& perm = sym.perm;
    const long long bwn = sym.bwn;
    kl = (int)std::min<long long>(bwn * bs + bs - 1, std::max<long long>(n - 1, 0));
    // narrow bands: one wavefront with the window of a step in registers (k_band_lu_wave; the half bandwidth is rounded up to the
    // next instantiated one, the extra diagonals are zeros); FEM_AMD_LU_WAVE=0: the column walk of one workgroup.  The two write the
    // same factors up to the order of roundings inside an fma.
    const char* we = std::getenv("FEM_AMD_LU_WAVE");
    wave = !(we && std::atoi(we) == 0) && kl <= 21;
    const int kl_true = kl;
    if (wave) kl = kl <= 5 ? 5 : kl <= 9 ? 9 : kl <= 13 ? 13 : kl <= 17 ? 17 : 21;
    if (std::getenv("FEM_AMD_DIRECT_VERBOSE"))
        fprintf(stderr, "[fem_amd lu] band LU: n = %lld, half bandwidth %d%s, %s\n", n, kl_true, wave ? (" (stored as " + std::to_string(kl) + ")").c_str() : "",
                wave ? "one wavefront, window in registers" : "one workgroup, column walk");
    wb = 3 * kl + 1;
    size_t free_b = 0, total_b = 0;
    const double need = (double)n * wb * sizeof(double);
    if (need < 1073741824.0) free_b = (size_t)1 << 40;   // (see BandCholesky::setup: no hipMemGetInfo for a factor under 1 GiB)
    else {
        FEM_HIP(hipMemGetInfo(&free_b, &total_b));
        if (!(need < 0.8 * ((double)free_b + (double)pool_cached_bytes()))) {   // (only a factor that may not fit empties the pool)
            pool_trim();
            FEM_HIP(hipMemGetInfo(&free_b, &total_b));
        } else free_b += pool_cached_bytes();
    }
    FEM_CHECK(need < 0.8 * (double)free_b, 9, "pc lu: the band of the unsymmetric factor needs " +
                                                  std::to_string((long long)(need / 1048576.0)) + " MiB: use -ksp_type bcgs");
    if (sym.d_perm_dof.n != (size_t)n) {   // (once per sparsity: the numbering does not change from one Jacobian to the next)
        sym.d_perm.alloc((size_t)nn);
        sym.d_perm.upload(perm.data(), perm.size(), s);
        sym.d_perm_dof.alloc((size_t)n);
        hipLaunchKernelGGL(k_perm_dof, dim3(cdiv(nn, 256)), dim3(256), 0, s, nn, bs, sym.d_perm.p, sym.d_perm_dof.p);
    }
    const DBuf<int>& dperm = sym.d_perm;
    perm_dof = sym.d_perm_dof.p;
    ab.alloc((size_t)n * wb);
    ab.zero(s);
    ipiv.alloc((size_t)n); v0.alloc((size_t)n + 2); flag.alloc(2);
    if (bs == 3)
        hipLaunchKernelGGL(k_bandfull_fill<3>, dim3(cdiv(nn, 256)), dim3(256), 0, s, nn, A->boff.p, A->bcol.p, A->values(),
                           A->rowlen.p, dperm.p, kl, wb, ab.p, flag.p, v0.p + n);
    else
        hipLaunchKernelGGL(k_bandfull_fill<2>, dim3(cdiv(nn, 256)), dim3(256), 0, s, nn, A->boff.p, A->bcol.p, A->values(),
                           A->rowlen.p, dperm.p, kl, wb, ab.p, flag.p, v0.p + n);
    // (the largest non-penalty entry: non-negative doubles order like their bits -- many workgroups and one atomic max each)
    hipLaunchKernelGGL(k_val_amax, dim3(64), dim3(256), 0, s, (size_t)n * wb, ab.p, reinterpret_cast<unsigned long long*>(v0.p + n));
    band_ready = false; bcr = false;
    const char* be = std::getenv("FEM_AMD_LU_BCR");
    if (wave && !(be && std::atoi(be) == 0) && n >= 32LL * kl) {
        // block cyclic reduction (see k_bcr_*): the matrix alone is reduced once; no small pivot on the way -> the solves use it
        nbk = (int)((n + kl - 1) / kl);
        const size_t nb2 = (size_t)nbk * kl * kl;
        for (DBuf<double>* a : {&cL0, &cD0, &cU0, &cL, &cD, &cU}) if (a->n != nb2) a->alloc(nb2);
        if (cr.n != (size_t)nbk * kl) cr.alloc((size_t)nbk * kl);
        const unsigned gb = cdiv(nb2, 256);
        switch (kl) {
        case 5: hipLaunchKernelGGL(k_bcr_blocks<5>, dim3(gb), dim3(256), 0, s, n, nbk, ab.p, v0.p + n, cL0.p, cD0.p, cU0.p); break;
        case 9: hipLaunchKernelGGL(k_bcr_blocks<9>, dim3(gb), dim3(256), 0, s, n, nbk, ab.p, v0.p + n, cL0.p, cD0.p, cU0.p); break;
        case 13: hipLaunchKernelGGL(k_bcr_blocks<13>, dim3(gb), dim3(256), 0, s, n, nbk, ab.p, v0.p + n, cL0.p, cD0.p, cU0.p); break;
        case 17: hipLaunchKernelGGL(k_bcr_blocks<17>, dim3(gb), dim3(256), 0, s, n, nbk, ab.p, v0.p + n, cL0.p, cD0.p, cU0.p); break;
        default: hipLaunchKernelGGL(k_bcr_blocks<21>, dim3(gb), dim3(256), 0, s, n, nbk, ab.p, v0.p + n, cL0.p, cD0.p, cU0.p); break;
        }
        // One resident launch per solve (k_bcr_resident; FEM_AMD_LU_BCR_RESIDENT=0: the launch per level).  It reduces matrix and
        // right-hand side together, so set-up does not run the reduction for the pivots' sake any more: the first solve meets a small
        // pivot as well as this would have, and goes to the band LU the same way (a Newton loop sets up and solves once per Jacobian:
        // half the reductions).
        const char* re = std::getenv("FEM_AMD_LU_BCR_RESIDENT");
        bcr_resident = !(re && std::atoi(re) == 0);
        if (bcr_resident) {
            for (DBuf<double>* a : {&cEL, &cEU}) if (a->n != nb2) a->alloc(nb2);
            if (cz.n != (size_t)nbk * kl) cz.alloc((size_t)nbk * kl);
            if (bcr_words.n != (size_t)GridBarrier::WORDS + 32) bcr_words.alloc((size_t)GridBarrier::WORDS + 32);
            // few blocks (a wavefront each is resident at once): the parallel form of the reduction (FEM_AMD_LU_BCR_PCR=0: never)
            const char* pe = std::getenv("FEM_AMD_LU_BCR_PCR");
            int occ = 0;
            const void* kp = kl == 5 ? (const void*)k_pcr_resident<5> : kl == 9 ? (const void*)k_pcr_resident<9> : kl == 13 ? (const void*)k_pcr_resident<13>
                           : kl == 17 ? (const void*)k_pcr_resident<17> : (const void*)k_pcr_resident<21>;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kp, 256, 0) != hipSuccess) { (void)hipGetLastError(); occ = 0; }
            bcr_pcr = !(pe && std::atoi(pe) == 0) && !sym.pcr_failed && (long long)(nbk + 3) / 4 <= (long long)std::min(occ, 2) * ctx->num_cu;
            if (bcr_pcr && pcr_extra.n != nb2 + 2 * (size_t)nbk * kl) pcr_extra.alloc(nb2 + 2 * (size_t)nbk * kl);
            bcr = true;
        } else
            bcr = bcr_reduce(false);
        if (std::getenv("FEM_AMD_DIRECT_VERBOSE"))
            fprintf(stderr, "[fem_amd lu] block cyclic reduction: %d blocks of %d rows: %s\n", nbk, kl,
                    bcr_resident ? (bcr_pcr ? "used (parallel cyclic reduction, one resident launch per solve, which also looks at the pivots)" : "used (one resident launch per solve, which also looks at the pivots)") : bcr ? "used" : "a small pivot on the way: band LU");
        if (bcr) { null_pivots = 0; failed = false; return; }
        flag.zero(s);
    }
    factor_band();
}

// levels of the block cyclic reduction on the work copies; with_rhs: cr holds the right-hand side by blocks and receives the solution
bool BandLU::bcr_reduce(bool with_rhs)
{
    fem_ctx ctx = A->ctx;
    hipStream_t s = ctx->stream;
    const size_t nb2 = (size_t)nbk * kl * kl;
    FEM_HIP(hipMemcpyAsync(cL.p, cL0.p, nb2 * sizeof(double), hipMemcpyDeviceToDevice, s));
    FEM_HIP(hipMemcpyAsync(cD.p, cD0.p, nb2 * sizeof(double), hipMemcpyDeviceToDevice, s));
    FEM_HIP(hipMemcpyAsync(cU.p, cU0.p, nb2 * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (!with_rhs) cr.zero(s);
    flag.zero(s);
    const double* am = v0.p + n;
    auto elim = [&](int count, int j0, int jstep) {
        if (count <= 0) return;
        switch (kl) {
        case 5: hipLaunchKernelGGL(k_bcr_eliminate<5>, dim3(count), dim3(64), 0, s, count, j0, jstep, cL.p, cD.p, cU.p, cr.p, am, flag.p); break;
        case 9: hipLaunchKernelGGL(k_bcr_eliminate<9>, dim3(count), dim3(64), 0, s, count, j0, jstep, cL.p, cD.p, cU.p, cr.p, am, flag.p); break;
        case 13: hipLaunchKernelGGL(k_bcr_eliminate<13>, dim3(count), dim3(64), 0, s, count, j0, jstep, cL.p, cD.p, cU.p, cr.p, am, flag.p); break;
        case 17: hipLaunchKernelGGL(k_bcr_eliminate<17>, dim3(count), dim3(64), 0, s, count, j0, jstep, cL.p, cD.p, cU.p, cr.p, am, flag.p); break;
        default: hipLaunchKernelGGL(k_bcr_eliminate<21>, dim3(count), dim3(64), 0, s, count, j0, jstep, cL.p, cD.p, cU.p, cr.p, am, flag.p); break;
        }
    };
    auto update = [&](int count, int istep, int st) {
        if (count <= 0) return;
        switch (kl) {
        case 5: hipLaunchKernelGGL(k_bcr_update<5>, dim3(count), dim3(64), 0, s, count, istep, st, nbk, cL.p, cD.p, cU.p, cr.p); break;
        case 9: hipLaunchKernelGGL(k_bcr_update<9>, dim3(count), dim3(64), 0, s, count, istep, st, nbk, cL.p, cD.p, cU.p, cr.p); break;
        case 13: hipLaunchKernelGGL(k_bcr_update<13>, dim3(count), dim3(64), 0, s, count, istep, st, nbk, cL.p, cD.p, cU.p, cr.p); break;
        case 17: hipLaunchKernelGGL(k_bcr_update<17>, dim3(count), dim3(64), 0, s, count, istep, st, nbk, cL.p, cD.p, cU.p, cr.p); break;
        default: hipLaunchKernelGGL(k_bcr_update<21>, dim3(count), dim3(64), 0, s, count, istep, st, nbk, cL.p, cD.p, cU.p, cr.p); break;
        }
    };
    auto back = [&](int count, int j0, int jstep, int st) {
        if (count <= 0) return;
        switch (kl) {
        case 5: hipLaunchKernelGGL(k_bcr_back<5>, dim3(count), dim3(64), 0, s, count, j0, jstep, st, nbk, cL.p, cU.p, cr.p); break;
        case 9: hipLaunchKernelGGL(k_bcr_back<9>, dim3(count), dim3(64), 0, s, count, j0, jstep, st, nbk, cL.p, cU.p, cr.p); break;
        case 13: hipLaunchKernelGGL(k_bcr_back<13>, dim3(count), dim3(64), 0, s, count, j0, jstep, st, nbk, cL.p, cU.p, cr.p); break;
        case 17: hipLaunchKernelGGL(k_bcr_back<17>, dim3(count), dim3(64), 0, s, count, j0, jstep, st, nbk, cL.p, cU.p, cr.p); break;
        default: hipLaunchKernelGGL(k_bcr_back<21>, dim3(count), dim3(64), 0, s, count, j0, jstep, st, nbk, cL.p, cU.p, cr.p); break;
        }
    };
    int smax = 0;
    for (int st = 1; st < nbk; st *= 2) {
        elim((nbk - st + 2 * st - 1) / (2 * st), st, 2 * st);          // blocks st, 3 st, 5 st, ...
        update((nbk + 2 * st - 1) / (2 * st), 2 * st, st);             // blocks 0, 2 st, 4 st, ...
        smax = st;
    }
    elim(1, 0, 0);                                                     // the one block left
    if (with_rhs)
        for (int st = smax; st >= 1; st /= 2) back((nbk - st + 2 * st - 1) / (2 * st), st, 2 * st, st);
    FEM_HIP(hipGetLastError());
    int hflag = 0;
    flag.download(&hflag, 1, s);
    FEM_HIP(hipStreamSynchronize(s));
    return hflag == 0;
}

// the band LU proper: at set-up, or when a solve by block cyclic reduction has failed its backward-error test
void BandLU::factor_band()
{
    fem_ctx ctx = A->ctx;
    hipStream_t s = ctx->stream;
    flag.zero(s);
    if (wave) {
        double* am = v0.p + n;
        switch (kl) {
        case 5: hipLaunchKernelGGL(k_band_lu_wave<5>, dim3(1), dim3(64), 0, s, n, ab.p, ipiv.p, am, flag.p); break;
        case 9: hipLaunchKernelGGL(k_band_lu_wave<9>, dim3(1), dim3(64), 0, s, n, ab.p, ipiv.p, am, flag.p); break;
        case 13: hipLaunchKernelGGL(k_band_lu_wave<13>, dim3(1), dim3(64), 0, s, n, ab.p, ipiv.p, am, flag.p); break;
        case 17: hipLaunchKernelGGL(k_band_lu_wave<17>, dim3(1), dim3(64), 0, s, n, ab.p, ipiv.p, am, flag.p); break;
        default: hipLaunchKernelGGL(k_band_lu_wave<21>, dim3(1), dim3(64), 0, s, n, ab.p, ipiv.p, am, flag.p); break;
        }
    } else
        hipLaunchKernelGGL(k_band_lu, dim3(1), dim3(256), 0, s, n, kl, wb, ab.p, ipiv.p, v0.p + n, flag.p);
    FEM_HIP(hipGetLastError());
    int hflag[2] = {0, 0};
    flag.download(hflag, 2, s);
    FEM_HIP(hipStreamSynchronize(s));
    null_pivots = hflag[0];      // (MUMPS ICNTL(24) = 1 fixes them the same way, src/main.F90:368)
    failed = hflag[1] != 0;      // a NaN reached a pivot column: KSP_DIVERGED_PC_FAILED
    band_ready = true;
}

void BandLU::solve(const double* b, double* x)
{
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    if (batched) {
        if (bs == 3) hipLaunchKernelGGL(k_comp_solve<3>, dim3(ncomp), dim3(64), 0, s, ncomp, comp_ptr, comp_nodes, fac_off, fac.p, ipiv.p, b, x);
        else hipLaunchKernelGGL(k_comp_solve<2>, dim3(ncomp), dim3(64), 0, s, ncomp, comp_ptr, comp_nodes, fac_off, fac.p, ipiv.p, b, x);
        FEM_HIP(hipGetLastError());
        return;
    }
    const unsigned g = cdiv((size_t)n, 256);
    if (bcr && bcr_resident) {
        // prep, reduction, backward error, one read-back of (barrier gave up, small pivots, error)
        const int nwords = GridBarrier::WORDS + 32;
        unsigned* status = bcr_words.p + GridBarrier::WORDS;            // [0] barrier  [1] pivots  [2..3] the error's bits
        const size_t npad = (size_t)nbk * kl;
        hipLaunchKernelGGL(k_bcr_prep, dim3(cdiv(std::max(npad, (size_t)nwords), 256)), dim3(256), 0, s, (size_t)n, npad, perm_dof, b,
                           cr.p, v0.p, bcr_words.p, nwords);
        GridBarrier bar{bcr_words.p, reinterpret_cast<int*>(status)};
        int* pflag = reinterpret_cast<int*>(status + 1);
        const double* am = v0.p + n;
        const double* xblocks = cr.p;
        if (bcr_pcr) {
            // few blocks: parallel cyclic reduction, a wavefront per block at every level (k_pcr_resident)
            const unsigned nwg = (unsigned)((nbk + 3) / 4);
            const size_t nb2 = (size_t)nbk * kl * kl, nr = (size_t)nbk * kl;
            double *DB = pcr_extra.p, *rA = DB + nb2, *rB = rA + nr;
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, s, nbk, cL0.p, cD0.p, cU0.p, cr.p, cL.p, cD.p, cU.p, rA, cEL.p, DB, cEU.p, rB,
                                   cz.p, am, pflag, bar);
            };
            switch (kl) {
            case 5: go(k_pcr_resident<5>); break;
            case 9: go(k_pcr_resident<9>); break;
            case 13: go(k_pcr_resident<13>); break;
            case 17: go(k_pcr_resident<17>); break;
            default: go(k_pcr_resident<21>); break;
            }
            xblocks = cz.p;
        } else {
            const unsigned nwg = (unsigned)std::min<long long>(((nbk + 1) / 2 + 3) / 4, std::max(ctx->num_cu, 1));
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, s, nbk, cL0.p, cD0.p, cU0.p, cL.p, cD.p, cU.p, cEL.p, cEU.p, cz.p, cr.p, am, pflag, bar);
            };
            switch (kl) {
            case 5: go(k_bcr_resident<5>); break;
            case 9: go(k_bcr_resident<9>); break;
            case 13: go(k_bcr_resident<13>); break;
            case 17: go(k_bcr_resident<17>); break;
            default: go(k_bcr_resident<21>); break;
            }
        }
        hipLaunchKernelGGL(k_band_backward_error, dim3(g), dim3(256), 0, s, n, kl, wb, ab.p, xblocks, v0.p,
                           reinterpret_cast<unsigned long long*>(status + 2));
        FEM_HIP(hipGetLastError());
        unsigned h[4] = {0, 0, 0, 0};
        FEM_HIP(hipMemcpyAsync(h, status, sizeof(h), hipMemcpyDeviceToHost, s));
        FEM_HIP(hipStreamSynchronize(s));
        double err;
        std::memcpy(&err, h + 2, sizeof(double));
        const char* te = std::getenv("FEM_AMD_LU_BCR_TOL");
        const bool ok = h[0] == 0 && h[1] == 0 && err <= (te ? std::atof(te) : 1e-10);
        if (ok) {
            hipLaunchKernelGGL(k_permute_out, dim3(g), dim3(256), 0, s, (size_t)n, perm_dof, xblocks, x);
            FEM_HIP(hipGetLastError());
            return;
        }
        const bool from_pcr = bcr_pcr && !h[0];
        if (std::getenv("FEM_AMD_DIRECT_VERBOSE")) {
            if (h[0]) fprintf(stderr, "[fem_amd lu] block cyclic reduction: the grid barrier of the resident launch gave up: a launch per level from here on\n");
            else if (from_pcr && h[1]) fprintf(stderr, "[fem_amd lu] block cyclic reduction, parallel form: small pivots on the way: the form with a way up on this sparsity from here on\n");
            else if (from_pcr) fprintf(stderr, "[fem_amd lu] block cyclic reduction, parallel form: backward error %.2e: the form with a way up on this sparsity from here on\n", err);
            else if (h[1]) fprintf(stderr, "[fem_amd lu] block cyclic reduction: a small pivot on the way: band LU from here on\n");
            else fprintf(stderr, "[fem_amd lu] block cyclic reduction: backward error %.2e: band LU from here on\n", err);
        }
        if (h[0]) bcr_resident = false;      // (the reduction itself is sound: its launches, below)
        else if (from_pcr) {
            // every unknown of the parallel form comes out of the Schur complement of all the others onto its block (on a slender
            // cantilever: pivots of the order (h / L)^3 of the entries); the reduction with a way up gets most unknowns from their
            // neighbours' values: it is the more accurate of the two on such strips.  Remembered on the sparsity (a Newton loop builds a new factorisation per Jacobian).
            bcr_pcr = false;
            A->pat->lu_sym->pcr_failed = true;
            solve(b, x);
            return;
        }
        else { bcr = false; if (!band_ready) factor_band(); }
    }
    if (bcr) {
        cr.zero(s);                                                     // (the rows past the matrix in the last block)
        hipLaunchKernelGGL(k_permute_in, dim3(g), dim3(256), 0, s, (size_t)n, perm_dof, b, cr.p);
        hipLaunchKernelGGL(k_permute_in, dim3(g), dim3(256), 0, s, (size_t)n, perm_dof, b, v0.p);   // (b in the band's numbering, for the test)
        bool ok = bcr_reduce(true);
        if (ok) {
            unsigned long long* eb = reinterpret_cast<unsigned long long*>(v0.p + n + 1);
            FEM_HIP(hipMemsetAsync(eb, 0, sizeof(double), s));
            hipLaunchKernelGGL(k_band_backward_error, dim3(g), dim3(256), 0, s, n, kl, wb, ab.p, cr.p, v0.p, eb);
            double err = 1.0;
            FEM_HIP(hipMemcpyAsync(&err, eb, sizeof(double), hipMemcpyDeviceToHost, s));
            FEM_HIP(hipStreamSynchronize(s));
            // (FEM_AMD_LU_BCR_TOL: the bound of the test -- 0 sends every solve to the band LU, which is how the tests reach that way)
            const char* te = std::getenv("FEM_AMD_LU_BCR_TOL");
            ok = err <= (te ? std::atof(te) : 1e-10);
            if (!ok && std::getenv("FEM_AMD_DIRECT_VERBOSE"))
                fprintf(stderr, "[fem_amd lu] block cyclic reduction: backward error %.2e: band LU from here on\n", err);
        }
        if (ok) {
            hipLaunchKernelGGL(k_permute_out, dim3(g), dim3(256), 0, s, (size_t)n, perm_dof, cr.p, x);
            FEM_HIP(hipGetLastError());
            return;
        }
        bcr = false;
        if (!band_ready) factor_band();
    }
    hipLaunchKernelGGL(k_permute_in, dim3(g), dim3(256), 0, s, (size_t)n, perm_dof, b, v0.p);
    if (wave) hipLaunchKernelGGL(k_band_solve_wave, dim3(1), dim3(64), 0, s, n, kl, wb, ab.p, ipiv.p, v0.p);
    else
        hipLaunchKernelGGL(k_band_lu_solve, dim3(1), dim3(256), 0, s, n, kl, wb, ab.p, ipiv.p, v0.p);
    hipLaunchKernelGGL(k_permute_out, dim3(g), dim3(256), 0, s, (size_t)n, perm_dof, v0.p, x);
    FEM_HIP(hipGetLastError());
}

void BandCholesky::solve(const double* b, double* x)
{
    fem_ctx ctx = A->ctx;
    ctx->bind();
    hipStream_t s = ctx->stream;
    const unsigned g = cdiv((size_t)n, 256);
    hipLaunchKernelGGL(k_permute_in, dim3(g), dim3(256), 0, s, (size_t)n, perm_dof.p, b, v0.p);
    for (long long k0 = 0; k0 < n; k0 += NB) {
        const int kb = (int)std::min<long long>(NB, n - k0);
        const int m = (int)(std::min<long long>(n, k0 + kb + bw) - (k0 + kb));
        hipLaunchKernelGGL(k_fwd, dim3(std::max(1, (m + 4 * FWD_ROWS - 1) / (4 * FWD_ROWS))), dim3(256), 0, s, bw, k0, kb, m,
                           ab.p, linv.p + (size_t)(k0 / NB) * NB * NB, v0.p, v1.p);
    }
    const long long nblk = (n + NB - 1) / NB;
    for (long long kbk = nblk - 1; kbk >= 0; kbk--) {
        const long long k0 = kbk * NB;
        const int kb = (int)std::min<long long>(NB, n - k0);
        const long long ncols = std::min<long long>(k0, bw);
        hipLaunchKernelGGL(k_bwd, dim3((unsigned)std::max<long long>(1, (ncols + 255) / 256)), dim3(256), 0, s, bw, k0, kb, ab.p,
                           linv.p + (size_t)(k0 / NB) * NB * NB, v1.p, v2.p);
    }
    hipLaunchKernelGGL(k_permute_out, dim3(g), dim3(256), 0, s, (size_t)n, perm_dof.p, v2.p, x);
    FEM_HIP(hipGetLastError());
}

}  // namespace fem
