// fem_amd/csrc/mg.hip -- box-aggregation multigrid preconditioner (see mg.h).
#include "mg.h"

#include <algorithm>
#include <cmath>

#include "comm.h"
#include "scan.h"

using namespace fem;

void mat_mult_raw(fem_mat A, const double* x, double* y, double* partial, const int* done);
void mat_get_diagonal_raw(fem_mat A, double* d);

namespace fem {

struct Grid3 {
    int nx, ny, nz;
    __host__ __device__ int nb() const { return nx * ny * nz; }
};

// ---- set-up kernels (integer / one-off work) ----------------------------------------------------
__global__ void __launch_bounds__(256) k_bbox_partial(size_t nn, int pd, const double* __restrict__ coords,
                                                      double* __restrict__ out)
{
    __shared__ double smin[3][4], smax[3][4];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < nn; i += st)
        for (int d = 0; d < pd; d++) { const double c = coords[i * pd + d]; mn[d] = fmin(mn[d], c); mx[d] = fmax(mx[d], c); }
    for (int d = 0; d < 3; d++)
        for (int o = 32; o > 0; o >>= 1) { mn[d] = fmin(mn[d], __shfl_down(mn[d], o)); mx[d] = fmax(mx[d], __shfl_down(mx[d], o)); }
    if ((threadIdx.x & 63) == 0)
        for (int d = 0; d < 3; d++) { smin[d][threadIdx.x >> 6] = mn[d]; smax[d][threadIdx.x >> 6] = mx[d]; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int d = 0; d < 3; d++) {
            out[blockIdx.x * 6 + d] = fmin(fmin(smin[d][0], smin[d][1]), fmin(smin[d][2], smin[d][3]));
            out[blockIdx.x * 6 + 3 + d] = fmax(fmax(smax[d][0], smax[d][1]), fmax(smax[d][2], smax[d][3]));
        }
}

__global__ void __launch_bounds__(256) k_assign_boxes(int nn, int pd, const double* __restrict__ coords, double lo0,
                                                      double lo1, double lo2, double invH, Grid3 g,
                                                      int* __restrict__ agg, int* __restrict__ cnt)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    const double lo[3] = {lo0, lo1, lo2};
    const int dim[3] = {g.nx, g.ny, g.nz};
    int idx[3] = {0, 0, 0};
    for (int d = 0; d < pd; d++) idx[d] = min(max((int)floor((coords[(size_t)v * pd + d] - lo[d]) * invH), 0), dim[d] - 1);
    const int b = idx[0] + g.nx * (idx[1] + g.ny * idx[2]);
    agg[v] = b;
    atomicAdd(&cnt[b], 1);
}

__global__ void __launch_bounds__(256) k_fill_boxes(int nn, const int* __restrict__ agg, const int* __restrict__ box_ptr,
                                                    int* __restrict__ cursor, int* __restrict__ box_nodes)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    const int b = agg[v];
    box_nodes[box_ptr[b] + atomicAdd(&cursor[b], 1)] = v;
}

// ascending node order inside every box: fixes the summation order of the Galerkin
// product and of the restriction (bitwise reproducible set-up and cycles)
__global__ void __launch_bounds__(256) k_sort_boxes(int nb, const int* __restrict__ box_ptr, int* __restrict__ box_nodes)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int p0 = box_ptr[b], p1 = box_ptr[b + 1];
    for (int i = p0 + 1; i < p1; i++) {
        const int v = box_nodes[i];
        int j = i - 1;
        while (j >= p0 && box_nodes[j] > v) { box_nodes[j + 1] = box_nodes[j]; j--; }
        box_nodes[j + 1] = v;
    }
}

__global__ void __launch_bounds__(256) k_free_weights(size_t n, const double* __restrict__ diag, double* __restrict__ w,
                                                      double* __restrict__ dinv)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        const double d = diag[i];
        w[i] = (d >= 1.0e29) ? 0.0 : 1.0;   // penalty dofs (PENALTY_PARAM = 1e30) stay out of the coarse space
        dinv[i] = (d != 0.0) ? 1.0 / d : 1.0;
    }
}

__device__ __forceinline__ bool neighbour_box(const Grid3& g, int b, int slot, int& out)
{
    const int ix = b % g.nx, iy = (b / g.nx) % g.ny, iz = b / (g.nx * g.ny);
    const int jx = ix + slot % 3 - 1, jy = iy + (slot / 3) % 3 - 1, jz = iz + slot / 9 - 1;
    if (jx < 0 || jy < 0 || jz < 0 || jx >= g.nx || jy >= g.ny || jz >= g.nz) return false;
    out = jx + g.nx * (jy + g.ny * jz);
    return true;
}

// level-1 operator  A_1 = P^T K P : thread (box A, slot s) gathers every fine block
// K_ij with i in A and j in the neighbour box A+s, in fixed (node, column) order.
template <int BS>
__global__ void __launch_bounds__(256) k_galerkin_fine(Grid3 g, const int* __restrict__ box_ptr,
                                                       const int* __restrict__ box_nodes, const int* __restrict__ agg,
                                                       const int* __restrict__ rowlen, const int* __restrict__ boff,
                                                       const int* __restrict__ bcol, const double* __restrict__ val,
                                                       const double* __restrict__ w, double* __restrict__ st,
                                                       int* __restrict__ err)
{
    constexpr int BB = BS * BS;
    const int nb = g.nb();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 27 * nb) return;
    const int slot = t / nb, A = t - slot * nb;
    int T = -1;
    const bool inside = neighbour_box(g, A, slot, T);
    double acc[BB];
#pragma unroll
    for (int e = 0; e < BB; e++) acc[e] = 0.0;
    if (inside) {
        const int ax = A % g.nx, ay = (A / g.nx) % g.ny, az = A / (g.nx * g.ny);
        for (int p = box_ptr[A]; p < box_ptr[A + 1]; p++) {
            const int i = box_nodes[p];
            const size_t o0 = (size_t)boff[i >> 6];
            const int lane = i & 63, len = rowlen[i];
            for (int j = 0; j < len; j++) {
                const int c = bcol[(o0 + j) * 64 + lane];
                const int B = agg[c];
                if (slot == 13) {  // the centre thread also checks that coupled nodes sit in adjacent boxes
                    const int bx = B % g.nx, by = (B / g.nx) % g.ny, bz = B / (g.nx * g.ny);
                    if (abs(bx - ax) > 1 || abs(by - ay) > 1 || abs(bz - az) > 1) atomicExch(err, 1);
                }
                if (B != T) continue;
#pragma unroll
                for (int r = 0; r < BS; r++)
#pragma unroll
                    for (int q = 0; q < BS; q++)
                        acc[r * BS + q] += w[(size_t)BS * i + r] * w[(size_t)BS * c + q] *
                                           val[((o0 + j) * BB + r * BS + q) * 64 + lane];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < BB; e++) st[((size_t)slot * BB + e) * nb + A] = acc[e];
}

// The same product with one wavefront per box: lane = node of the box.  For each of the 27 slots
// every lane sums its own matching blocks (the row's column ids are re-read per slot from L1/L2,
// the 72-byte values only when they match, i.e. once), then a fixed shuffle tree adds the lanes:
// bitwise reproducible, no atomics, 64x more parallel than the thread-per-(box,slot) form.
template <int BS>
__global__ void __launch_bounds__(256) k_galerkin_fine_wave(Grid3 g, const int* __restrict__ box_ptr,
                                                            const int* __restrict__ box_nodes, const int* __restrict__ agg,
                                                            const int* __restrict__ rowlen, const int* __restrict__ boff,
                                                            const int* __restrict__ bcol, const double* __restrict__ val,
                                                            const double* __restrict__ w, double* __restrict__ st,
                                                            int* __restrict__ err)
{
    constexpr int BB = BS * BS;
    const int nb = g.nb();
    const int A = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (A >= nb) return;
    const int ax = A % g.nx, ay = (A / g.nx) % g.ny, az = A / (g.nx * g.ny);
    const int p0 = box_ptr[A], p1 = box_ptr[A + 1];
    // pass 1: slot id of each block of the lane's row, 5 bits each, packed in two words per 12 entries
    // (only the first node chunk of a box is cached; boxes with more than 64 nodes loop below)
    for (int s = 0; s < 27; s++) {
        double acc[BB];
#pragma unroll
        for (int e = 0; e < BB; e++) acc[e] = 0.0;
        for (int p = p0 + lane; p < p1; p += 64) {
            const int i = box_nodes[p];
            const size_t o0 = (size_t)boff[i >> 6];
            const int ln = i & 63, len = rowlen[i];
            for (int j = 0; j < len; j++) {
                const int c = bcol[(o0 + j) * 64 + ln];
                const int B = agg[c];
                const int dx = B % g.nx - ax, dy = (B / g.nx) % g.ny - ay, dz = B / (g.nx * g.ny) - az;
                if (abs(dx) > 1 || abs(dy) > 1 || abs(dz) > 1) { if (s == 13) atomicExch(err, 1); continue; }
                if ((dx + 1) + 3 * (dy + 1) + 9 * (dz + 1) != s) continue;
#pragma unroll
                for (int r = 0; r < BS; r++)
#pragma unroll
                    for (int q = 0; q < BS; q++)
                        acc[r * BS + q] += w[(size_t)BS * i + r] * w[(size_t)BS * c + q] *
                                           val[((o0 + j) * BB + r * BS + q) * 64 + ln];
            }
        }
#pragma unroll
        for (int e = 0; e < BB; e++) {
            double a = acc[e];
            for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
            if (lane == 0) st[((size_t)s * BB + e) * nb + A] = a;
        }
    }
}

// A_{l+1} = P^T A_l P between two box levels (children = 2x2x2 boxes)
template <int BS>
__global__ void __launch_bounds__(256) k_galerkin_box(Grid3 gf, Grid3 gc, const double* __restrict__ stf,
                                                      double* __restrict__ stc)
{
    constexpr int BB = BS * BS;
    const int nbc = gc.nb(), nbf = gf.nb();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 27 * nbc) return;
    const int slot = t / nbc, A = t - slot * nbc;
    int T = -1;
    const bool inside = neighbour_box(gc, A, slot, T);
    double acc[BB];
#pragma unroll
    for (int e = 0; e < BB; e++) acc[e] = 0.0;
    if (inside) {
        const int ax = A % gc.nx, ay = (A / gc.nx) % gc.ny, az = A / (gc.nx * gc.ny);
        for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
            const int fx = 2 * ax + cx, fy = 2 * ay + cy, fz = 2 * az + cz;
            if (fx >= gf.nx || fy >= gf.ny || fz >= gf.nz) continue;
            const int a = fx + gf.nx * (fy + gf.ny * fz);
            for (int s = 0; s < 27; s++) {
                int b;
                if (!neighbour_box(gf, a, s, b)) continue;
                const int bx = (b % gf.nx) >> 1, by = ((b / gf.nx) % gf.ny) >> 1, bz = (b / (gf.nx * gf.ny)) >> 1;
                if (bx + gc.nx * (by + gc.ny * bz) != T) continue;
#pragma unroll
                for (int e = 0; e < BB; e++) acc[e] += stf[((size_t)s * BB + e) * nbf + a];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < BB; e++) stc[((size_t)slot * BB + e) * nbc + A] = acc[e];
}

template <int BS>
__global__ void __launch_bounds__(256) k_level_dinv(int nb, const double* __restrict__ st, double* __restrict__ dinv)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    for (int r = 0; r < BS; r++) {
        const double d = st[((size_t)13 * BS * BS + r * BS + r) * nb + b];
        dinv[(size_t)b * BS + r] = (d != 0.0) ? 1.0 / d : 1.0;
    }
}

// ---- cycle kernels ------------------------------------------------------------------------------------
// t = r - A x on a box level (27-point block stencil, thread per box)
template <int BS>
__global__ void __launch_bounds__(256) k_box_residual(Grid3 g, const double* __restrict__ st, const double* __restrict__ x,
                                                      const double* __restrict__ r, double* __restrict__ t)
{
    constexpr int BB = BS * BS;
    const int nb = g.nb();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    double acc[BS];
#pragma unroll
    for (int q = 0; q < BS; q++) acc[q] = r[(size_t)b * BS + q];
    for (int s = 0; s < 27; s++) {
        int c;
        if (!neighbour_box(g, b, s, c)) continue;
#pragma unroll
        for (int q = 0; q < BS; q++)
#pragma unroll
            for (int p = 0; p < BS; p++) acc[q] -= st[((size_t)s * BB + q * BS + p) * nb + b] * x[(size_t)c * BS + p];
    }
#pragma unroll
    for (int q = 0; q < BS; q++) t[(size_t)b * BS + q] = acc[q];
}

// same for the small levels, which are latency-bound with one thread per box: 32 lanes per
// box (one stencil slot each, 27 active), fixed shuffle tree over the slots
template <int BS>
__global__ void __launch_bounds__(256) k_box_residual_slots(Grid3 g, const double* __restrict__ st,
                                                            const double* __restrict__ x, const double* __restrict__ r,
                                                            double* __restrict__ t)
{
    constexpr int BB = BS * BS;
    const int nb = g.nb();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = gid >> 5, s = gid & 31;
    double acc[BS];
#pragma unroll
    for (int q = 0; q < BS; q++) acc[q] = 0.0;
    int c;
    if (b < nb && s < 27 && neighbour_box(g, b, s, c)) {
#pragma unroll
        for (int q = 0; q < BS; q++)
#pragma unroll
            for (int p = 0; p < BS; p++) acc[q] -= st[((size_t)s * BB + q * BS + p) * nb + b] * x[(size_t)c * BS + p];
    }
#pragma unroll
    for (int q = 0; q < BS; q++)
        for (int o = 16; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, 32);
    if (b < nb && s == 0) {
#pragma unroll
        for (int q = 0; q < BS; q++) t[(size_t)b * BS + q] = r[(size_t)b * BS + q] + acc[q];
    }
}

// x = omega * dinv * r
__global__ void __launch_bounds__(256) k_smooth_first(size_t n, double omega, const double* __restrict__ dinv,
                                                      const double* __restrict__ r, double* __restrict__ x)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) x[i] = omega * dinv[i] * r[i];
}
// x += omega * dinv * t          (t already holds r - A x)
__global__ void __launch_bounds__(256) k_smooth_add(size_t n, double omega, const double* __restrict__ dinv,
                                                    const double* __restrict__ t, double* __restrict__ x)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) x[i] += omega * dinv[i] * t[i];
}
// z += omega * dinv * (r - Kz)   (fine level; kz = K z)
__global__ void __launch_bounds__(256) k_fine_post(size_t n, double omega, const double* __restrict__ dinv,
                                                   const double* __restrict__ r, const double* __restrict__ kz,
                                                   double* __restrict__ z)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) z[i] += omega * dinv[i] * (r[i] - kz[i]);
}

// r_1[B] = sum_{i in B} w_i (r_i - (K z)_i)      one wavefront per box, lanes over the box's
// nodes (ascending), fixed shuffle tree -> bitwise reproducible
template <int BS>
__global__ void __launch_bounds__(256) k_restrict_fine(int nb, const int* __restrict__ box_ptr,
                                                       const int* __restrict__ box_nodes, const double* __restrict__ w,
                                                       const double* __restrict__ r, const double* __restrict__ kz,
                                                       double* __restrict__ rc)
{
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (b >= nb) return;
    double acc[BS];
#pragma unroll
    for (int q = 0; q < BS; q++) acc[q] = 0.0;
    for (int p = box_ptr[b] + lane; p < box_ptr[b + 1]; p += 64) {
        const size_t i = (size_t)box_nodes[p] * BS;
#pragma unroll
        for (int q = 0; q < BS; q++) acc[q] += w[i + q] * (kz ? r[i + q] - kz[i + q] : r[i + q]);
    }
#pragma unroll
    for (int q = 0; q < BS; q++)
        for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < BS; q++) rc[(size_t)b * BS + q] = acc[q];
    }
}

// z_i += alpha * w_i * x_1[agg(i)]
template <int BS>
__global__ void __launch_bounds__(256) k_prolong_fine(int nn, double alpha, const int* __restrict__ agg,
                                                      const double* __restrict__ w, const double* __restrict__ xc,
                                                      double* __restrict__ z)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    const size_t b = (size_t)agg[v] * BS, i = (size_t)v * BS;
#pragma unroll
    for (int q = 0; q < BS; q++) z[i + q] += alpha * w[i + q] * xc[b + q];
}

// additive fine level, one pass: z_i = dinv_i r_i + alpha w_i x_1[agg(i)], plus the partials of
// z.z and z.r that the CG needs next (same fixed-order two-stage reduction as the Jacobi path)
template <int BS>
__global__ void __launch_bounds__(256) k_additive_finish(int nn, double alpha, const int* __restrict__ agg,
                                                         const double* __restrict__ w, const double* __restrict__ dinv,
                                                         const double* __restrict__ r, const double* __restrict__ xc,
                                                         double* __restrict__ z, double* __restrict__ pzz,
                                                         double* __restrict__ pzr, const int* __restrict__ done)
{
    if (done && *done) return;
    __shared__ double sh[2][4];
    double zz = 0.0, zr = 0.0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nn; v += gridDim.x * blockDim.x) {
        const size_t b = (size_t)agg[v] * BS, i = (size_t)v * BS;
#pragma unroll
        for (int q = 0; q < BS; q++) {
            const double ri = r[i + q];
            const double zi = dinv[i + q] * ri + alpha * w[i + q] * xc[b + q];
            z[i + q] = zi;
            zz += zi * zi; zr += zi * ri;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { zz += __shfl_down(zz, o); zr += __shfl_down(zr, o); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = zz; sh[1][threadIdx.x >> 6] = zr; }
    __syncthreads();
    if (threadIdx.x == 0) {
        pzz[blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        pzr[blockIdx.x] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

template <int BS>
__global__ void __launch_bounds__(256) k_restrict_box(Grid3 gf, Grid3 gc, const double* __restrict__ tf,
                                                      double* __restrict__ rc)
{
    const int A = blockIdx.x * blockDim.x + threadIdx.x;
    if (A >= gc.nb()) return;
    const int ax = A % gc.nx, ay = (A / gc.nx) % gc.ny, az = A / (gc.nx * gc.ny);
    double acc[BS];
#pragma unroll
    for (int q = 0; q < BS; q++) acc[q] = 0.0;
    for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
        const int fx = 2 * ax + cx, fy = 2 * ay + cy, fz = 2 * az + cz;
        if (fx >= gf.nx || fy >= gf.ny || fz >= gf.nz) continue;
        const size_t a = (size_t)(fx + gf.nx * (fy + gf.ny * fz)) * BS;
#pragma unroll
        for (int q = 0; q < BS; q++) acc[q] += tf[a + q];
    }
#pragma unroll
    for (int q = 0; q < BS; q++) rc[(size_t)A * BS + q] = acc[q];
}

template <int BS>
__global__ void __launch_bounds__(256) k_prolong_box(Grid3 gf, Grid3 gc, double alpha, const double* __restrict__ xc,
                                                     double* __restrict__ xf)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= gf.nb()) return;
    const int px = (a % gf.nx) >> 1, py = ((a / gf.nx) % gf.ny) >> 1, pz = (a / (gf.nx * gf.ny)) >> 1;
    const size_t A = (size_t)(px + gc.nx * (py + gc.ny * pz)) * BS;
#pragma unroll
    for (int q = 0; q < BS; q++) xf[(size_t)a * BS + q] += alpha * xc[A + q];
}

// ---- coarsest grid: dense inverse by Gauss-Jordan (one workgroup) -------------------------------------
template <int BS>
__global__ void __launch_bounds__(256) k_dense_from_stencil(Grid3 g, const double* __restrict__ st, double* __restrict__ M)
{
    constexpr int BB = BS * BS;
    const int nb = g.nb(), n = nb * BS;
    for (int idx = threadIdx.x + blockIdx.x * blockDim.x; idx < n * 2 * n; idx += gridDim.x * blockDim.x) M[idx] = 0.0;
    __syncthreads();
    if (blockIdx.x != 0) return;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        for (int s = 0; s < 27; s++) {
            int c;
            if (!neighbour_box(g, b, s, c)) continue;
            for (int q = 0; q < BS; q++)
                for (int p = 0; p < BS; p++)
                    M[(size_t)(b * BS + q) * 2 * n + c * BS + p] += st[((size_t)s * BB + q * BS + p) * nb + b];
        }
        for (int q = 0; q < BS; q++) {
            const int i = b * BS + q;
            if (M[(size_t)i * 2 * n + i] == 0.0) M[(size_t)i * 2 * n + i] = 1.0;  // empty / fully constrained box
            M[(size_t)i * 2 * n + n + i] = 1.0;
        }
    }
}

// Gauss-Jordan on [A | I] (n x 2n, row-major), two small launches per pivot so that the
// rank-1 update runs on the whole chip (a single workgroup needs ~50 ms for n = 375).  A is
// symmetric positive (semi-)definite with unit rows for empty boxes: no pivot search.
__global__ void __launch_bounds__(256) k_gj_pivot(int n, int k, const double* __restrict__ M, double* __restrict__ rowk,
                                                  double* __restrict__ colk)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const double piv = 1.0 / M[(size_t)k * 2 * n + k];
    if (t < 2 * n) rowk[t] = M[(size_t)k * 2 * n + t] * piv;
    if (t < n) colk[t] = M[(size_t)t * 2 * n + k];
}
__global__ void __launch_bounds__(256) k_gj_update(int n, int k, double* __restrict__ M, const double* __restrict__ rowk,
                                                   const double* __restrict__ colk)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 2 * n) return;
    const int i = idx / (2 * n), j = idx - i * 2 * n;
    M[idx] = (i == k) ? rowk[j] : M[idx] - colk[i] * rowk[j];
}

// x = A^-1 r on the coarsest grid: one wavefront per row, fixed shuffle tree
__global__ void __launch_bounds__(256) k_dense_apply(int n, const double* __restrict__ M, const double* __restrict__ r,
                                                     double* __restrict__ x)
{
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n) return;
    double s = 0.0;
    for (int j = lane; j < n; j += 64) s += M[(size_t)i * 2 * n + n + j] * r[j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) x[i] = s;
}

// ---- host side ----------------------------------------------------------------------------------------------
static unsigned g1(size_t n) { return std::max(1u, cdiv(n, 256)); }
static unsigned gs(size_t n) { return std::min<unsigned>(std::max(1u, cdiv(n, 256)), 2048); }

template <int BS>
static void setup_t(Multigrid& mg)
{
    fem_mat A = mg.A;
    fem_mesh mesh = A->mesh;
    fem_ctx ctx = A->ctx;
    hipStream_t s = ctx->stream;
    const int nn = mesh->nnds, pd = mesh->pdim;
    const size_t ndof = (size_t)nn * BS;
    // bounding box
    const int pg = 256;
    DBuf<double> dbb((size_t)pg * 6);
    hipLaunchKernelGGL(k_bbox_partial, dim3(pg), dim3(256), 0, s, (size_t)nn, pd, mesh->coords.p, dbb.p);
    std::vector<double> hbb((size_t)pg * 6);
    dbb.download(hbb.data(), hbb.size(), s);
    FEM_HIP(hipStreamSynchronize(s));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int b = 0; b < pg; b++)
        for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], hbb[b * 6 + d]); hi[d] = std::max(hi[d], hbb[b * 6 + 3 + d]); }
    double vol = 1.0;
    for (int d = 0; d < pd; d++) vol *= std::max(hi[d] - lo[d], 1e-300);
    const double hnode = std::pow(vol / std::max(nn, 1), 1.0 / pd);  // mean node spacing
    const double H = 4.0 * hnode;                                    // ~4 nodes per box edge
    Grid3 g{1, 1, 1};
    int* gd[3] = {&g.nx, &g.ny, &g.nz};
    for (int d = 0; d < pd; d++) *gd[d] = std::max(1, (int)std::floor((hi[d] - lo[d]) / H + 1e-9) + 1);
    for (int d = pd; d < 3; d++) lo[d] = 0.0;
    FEM_CHECK((long long)g.nx * g.ny * g.nz < (1LL << 26), 9, "pc mg: box grid too large");

    // level-1 aggregates and their node lists
    mg.agg.alloc(nn);
    int nb = g.nb();
    DBuf<int> cnt((size_t)nb), cursor((size_t)nb), tmp(exclusive_scan_tmp_ints((size_t)nb + 1));
    cnt.zero(s); cursor.zero(s);
    hipLaunchKernelGGL(k_assign_boxes, dim3(g1(nn)), dim3(256), 0, s, nn, pd, mesh->coords.p, lo[0], lo[1], lo[2], 1.0 / H,
                       g, mg.agg.p, cnt.p);
    mg.box_ptr.alloc((size_t)nb + 1);
    exclusive_scan_i32(cnt.p, mg.box_ptr.p, nb, tmp.p, s);
    mg.box_nodes.alloc((size_t)nn);
    hipLaunchKernelGGL(k_fill_boxes, dim3(g1(nn)), dim3(256), 0, s, nn, mg.agg.p, mg.box_ptr.p, cursor.p, mg.box_nodes.p);
    hipLaunchKernelGGL(k_sort_boxes, dim3(g1(nb)), dim3(256), 0, s, nb, mg.box_ptr.p, mg.box_nodes.p);

    // free-dof weights and fine Jacobi diagonal
    mg.w.alloc(ndof); mg.dinv0.alloc(ndof); mg.t0.alloc(ndof);
    mat_get_diagonal_raw(A, mg.t0.p);
    hipLaunchKernelGGL(k_free_weights, dim3(gs(ndof)), dim3(256), 0, s, ndof, mg.t0.p, mg.w.p, mg.dinv0.p);

    // level 1
    DBuf<int> err(1);
    err.zero(s);
    mg.lev.clear();
    auto new_level = [&](Grid3 gg) {
        mg.lev.emplace_back();
        MgLevel& L = mg.lev.back();
        L.nx = gg.nx; L.ny = gg.ny; L.nz = gg.nz; L.nb = gg.nb();
        L.st.alloc((size_t)27 * BS * BS * L.nb);
        L.dinv.alloc((size_t)L.nb * BS); L.x.alloc((size_t)L.nb * BS); L.r.alloc((size_t)L.nb * BS);
        L.t.alloc((size_t)L.nb * BS);
    };
    new_level(g);
    hipLaunchKernelGGL(k_galerkin_fine_wave<BS>, dim3(g1((size_t)64 * nb)), dim3(256), 0, s, g, mg.box_ptr.p, mg.box_nodes.p,
                       mg.agg.p, A->rowlen.p, A->boff.p, A->bcol.p, A->val.p, mg.w.p, mg.lev[0].st.p, err.p);
    hipLaunchKernelGGL(k_level_dinv<BS>, dim3(g1(nb)), dim3(256), 0, s, nb, mg.lev[0].st.p, mg.lev[0].dinv.p);
    // coarser levels: merge 2x2x2 boxes until the grid is at most 5 boxes per direction
    // (<= 375 dofs: inverted densely; every extra tiny level would cost seven latency-bound launches)
    while (std::max(g.nx, std::max(g.ny, g.nz)) > 5) {
        Grid3 gc{(g.nx + 1) / 2, (g.ny + 1) / 2, (g.nz + 1) / 2};
        new_level(gc);
        MgLevel& Lf = mg.lev[mg.lev.size() - 2];
        MgLevel& Lc = mg.lev.back();
        hipLaunchKernelGGL(k_galerkin_box<BS>, dim3(g1((size_t)27 * gc.nb())), dim3(256), 0, s, g, gc, Lf.st.p, Lc.st.p);
        hipLaunchKernelGGL(k_level_dinv<BS>, dim3(g1(gc.nb())), dim3(256), 0, s, gc.nb(), Lc.st.p, Lc.dinv.p);
        g = gc;
    }
    // coarsest: dense inverse
    mg.ncoarse = g.nb() * BS;
    mg.cinv.alloc((size_t)mg.ncoarse * 2 * mg.ncoarse);
    hipLaunchKernelGGL(k_dense_from_stencil<BS>, dim3(1), dim3(256), 0, s, g, mg.lev.back().st.p, mg.cinv.p);
    {
        const int n = mg.ncoarse;
        DBuf<double> rowk((size_t)2 * n), colk((size_t)n);
        for (int k = 0; k < n; k++) {
            hipLaunchKernelGGL(k_gj_pivot, dim3(g1((size_t)2 * n)), dim3(256), 0, s, n, k, mg.cinv.p, rowk.p, colk.p);
            hipLaunchKernelGGL(k_gj_update, dim3(g1((size_t)n * 2 * n)), dim3(256), 0, s, n, k, mg.cinv.p, rowk.p, colk.p);
        }
        FEM_HIP(hipStreamSynchronize(s));   // rowk/colk go out of scope
    }
    int herr = 0;
    err.download(&herr, 1, s);
    FEM_HIP(hipGetLastError());
    FEM_HIP(hipStreamSynchronize(s));
    FEM_CHECK(!herr, 9, "pc mg: an element spans more than two boxes of the aggregation grid (mesh too graded)");
}

template <int BS>
static void cycle_t(Multigrid& mg, size_t l)
{
    hipStream_t s = mg.A->ctx->stream;
    MgLevel& L = mg.lev[l];
    const Grid3 g{L.nx, L.ny, L.nz};
    const size_t n = (size_t)L.nb * BS;
    if (l + 1 == mg.lev.size()) {
        hipLaunchKernelGGL(k_dense_apply, dim3(g1(n * 64)), dim3(256), 0, s, (int)n, mg.cinv.p, L.r.p, L.x.p);
        return;
    }
    auto residual = [&] {   // small levels: 32 lanes per box (latency), large ones: one thread per box (bandwidth)
        if (L.nb <= 16384)
            hipLaunchKernelGGL(k_box_residual_slots<BS>, dim3(g1((size_t)L.nb * 32)), dim3(256), 0, s, g, L.st.p, L.x.p,
                               L.r.p, L.t.p);
        else
            hipLaunchKernelGGL(k_box_residual<BS>, dim3(g1(L.nb)), dim3(256), 0, s, g, L.st.p, L.x.p, L.r.p, L.t.p);
    };
    MgLevel& C = mg.lev[l + 1];
    const Grid3 gc{C.nx, C.ny, C.nz};
    hipLaunchKernelGGL(k_smooth_first, dim3(gs(n)), dim3(256), 0, s, n, mg.omega, L.dinv.p, L.r.p, L.x.p);
    residual();
    hipLaunchKernelGGL(k_restrict_box<BS>, dim3(g1(C.nb)), dim3(256), 0, s, g, gc, L.t.p, C.r.p);
    cycle_t<BS>(mg, l + 1);
    hipLaunchKernelGGL(k_prolong_box<BS>, dim3(g1(L.nb)), dim3(256), 0, s, g, gc, mg.alpha, C.x.p, L.x.p);
    residual();
    hipLaunchKernelGGL(k_smooth_add, dim3(gs(n)), dim3(256), 0, s, n, mg.omega, L.dinv.p, L.t.p, L.x.p);
}

template <int BS>
static void apply_t(Multigrid& mg, const double* r, double* z, const int* done, double* pzz, double* pzr, int npart)
{
    fem_mat A = mg.A;
    hipStream_t s = A->ctx->stream;
    const int nn = mg.nn;
    const size_t ndof = (size_t)nn * BS;
    MgLevel& L1 = mg.lev[0];
    if (mg.fine_additive) {
        // fine level additive (Jacobi + coarse-grid correction of the un-smoothed residual): the
        // preconditioner costs no fine-level product; the coarse hierarchy is still a V-cycle
        hipLaunchKernelGGL(k_restrict_fine<BS>, dim3(g1((size_t)L1.nb * 64)), dim3(256), 0, s, L1.nb, mg.box_ptr.p,
                           mg.box_nodes.p, mg.w.p, r, (const double*)nullptr, L1.r.p);
        cycle_t<BS>(mg, 0);
        if (pzz) {   // the caller wants z.z and z.r as well: one fused pass
            hipLaunchKernelGGL(k_additive_finish<BS>, dim3(npart), dim3(256), 0, s, nn, mg.alpha, mg.agg.p, mg.w.p,
                               mg.dinv0.p, r, L1.x.p, z, pzz, pzr, done);
        } else {
            hipLaunchKernelGGL(k_smooth_first, dim3(gs(ndof)), dim3(256), 0, s, ndof, 1.0, mg.dinv0.p, r, z);
            hipLaunchKernelGGL(k_prolong_fine<BS>, dim3(g1(nn)), dim3(256), 0, s, nn, mg.alpha, mg.agg.p, mg.w.p, L1.x.p, z);
        }
        return;
    }
    hipLaunchKernelGGL(k_smooth_first, dim3(gs(ndof)), dim3(256), 0, s, ndof, mg.omega, mg.dinv0.p, r, z);
    mat_mult_raw(A, z, mg.t0.p, nullptr, done);
    hipLaunchKernelGGL(k_restrict_fine<BS>, dim3(g1((size_t)L1.nb * 64)), dim3(256), 0, s, L1.nb, mg.box_ptr.p, mg.box_nodes.p,
                       mg.w.p, r, mg.t0.p, L1.r.p);
    cycle_t<BS>(mg, 0);
    hipLaunchKernelGGL(k_prolong_fine<BS>, dim3(g1(nn)), dim3(256), 0, s, nn, mg.alpha, mg.agg.p, mg.w.p, L1.x.p, z);
    mat_mult_raw(A, z, mg.t0.p, nullptr, done);
    hipLaunchKernelGGL(k_fine_post, dim3(gs(ndof)), dim3(256), 0, s, ndof, mg.omega, mg.dinv0.p, r, mg.t0.p, z);
}

void Multigrid::setup(fem_mat A_)
{
    A = A_; bs = A->bs; nn = A->nbrows;
    FEM_CHECK(!(A->mesh->halo && A->mesh->halo->comm->size > 1), 9, "pc mg: multi-domain runs are not supported yet");
    A->ctx->bind();
    if (bs == 3) setup_t<3>(*this); else setup_t<2>(*this);
}

void Multigrid::apply(const double* r, double* z, const int* done, double* pzz, double* pzr, int npart)
{
    if (bs == 3) apply_t<3>(*this, r, z, done, pzz, pzr, npart); else apply_t<2>(*this, r, z, done, pzz, pzr, npart);
}

bool Multigrid::fuses_dots() const { return fine_additive; }

size_t Multigrid::device_bytes() const
{
    size_t b = agg.bytes() + box_ptr.bytes() + box_nodes.bytes() + w.bytes() + dinv0.bytes() + t0.bytes() + cinv.bytes();
    for (auto& L : lev) b += L.st.bytes() + L.dinv.bytes() + L.x.bytes() + L.r.bytes() + L.t.bytes();
    return b;
}

}  // namespace fem
