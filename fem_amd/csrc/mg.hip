// fem_amd/csrc/mg.hip -- box-aggregation multigrid preconditioner (see mg.h).
#include "mg.h"
#include "grid_barrier.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

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
// Between a box level and the next coarser one: parent index = (child index + shift) >> 1 per direction.  Zero on an undivided
// mesh; on a partitioned mesh every domain keeps, per level, only the sub-grid of the global box grid its nodes reach, and the
// shift is what is left of the sub-grid's origin (its parity -- or all of it below the replicated coarsest level).
struct Shift3 {
    int x, y, z;
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

// (gg: the global box grid, lo its corner -- every domain of a partitioned mesh evaluates the same expression on the same
// coordinates, so all copies of a shared node land in the same box; g / off: this domain's sub-grid and its origin in gg)
__global__ void __launch_bounds__(256) k_assign_boxes(int nn, int pd, const double* __restrict__ coords, double lo0,
                                                      double lo1, double lo2, double invH, Grid3 gg, Grid3 g, Shift3 off,
                                                      int* __restrict__ agg, int* __restrict__ cnt)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    const double lo[3] = {lo0, lo1, lo2};
    const int dim[3] = {gg.nx, gg.ny, gg.nz};
    const int o[3] = {off.x, off.y, off.z};
    int idx[3] = {0, 0, 0};
    for (int d = 0; d < pd; d++) idx[d] = min(max((int)floor((coords[(size_t)v * pd + d] - lo[d]) * invH), 0), dim[d] - 1) - o[d];
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

__global__ void __launch_bounds__(256) k_free_weights(size_t n, const double* __restrict__ diag, unsigned char* __restrict__ w,
                                                      double* __restrict__ dinv)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        const double d = diag[i];
        w[i] = (d >= 1.0e29) ? 0 : 1;       // penalty dofs (PENALTY_PARAM = 1e30) stay out of the coarse space
        dinv[i] = (d != 0.0) ? 1.0 / d : 1.0;
    }
}

// restriction weights of a partitioned mesh: every shared node is counted by its owner only
__global__ void __launch_bounds__(256) k_owner_weights(int nn, int bs, const unsigned char* __restrict__ w,
                                                       const unsigned char* __restrict__ owned, unsigned char* __restrict__ wr)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    for (int r = 0; r < bs; r++) wr[(size_t)v * bs + r] = owned[v] ? w[(size_t)v * bs + r] : 0;
}
__global__ void __launch_bounds__(256) k_count_owned(int nn, const unsigned char* __restrict__ owned, double* __restrict__ out)
{
    __shared__ int sh[4];
    int c = 0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nn; v += gridDim.x * blockDim.x) c += owned[v] ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (double)(sh[0] + sh[1] + sh[2] + sh[3]));   // integers: exact in any order
}

__device__ __forceinline__ bool neighbour_box(const Grid3& g, int b, int slot, int& out)
{
    const int ix = b % g.nx, iy = (b / g.nx) % g.ny, iz = b / (g.nx * g.ny);
    const int jx = ix + slot % 3 - 1, jy = iy + (slot / 3) % 3 - 1, jz = iz + slot / 9 - 1;
    if (jx < 0 || jy < 0 || jz < 0 || jx >= g.nx || jy >= g.ny || jz >= g.nz) return false;
    out = jx + g.nx * (jy + g.ny * jz);
    return true;
}

// ---- rigid-body-mode interpolation -----------------------------------------------------------------
// Coarse dofs of a box: translations and infinitesimal rotations about the box centre
// (3-D: 3 + 3, 2-D: 2 + 1).  A fine node at offset d from the centre moves by  u = U + Theta x d.
template <int BS> struct CoarseDofs { static constexpr int value = (BS == 3) ? 6 : 3; };

// u[BS] = P_d * xc[CS]
template <int BS, int CS>
__device__ __forceinline__ void rbm_apply(const double (&d)[3], const double* xc, double (&u)[BS])
{
    if constexpr (BS == 3) {
        u[0] = xc[0] + xc[4] * d[2] - xc[5] * d[1];
        u[1] = xc[1] + xc[5] * d[0] - xc[3] * d[2];
        u[2] = xc[2] + xc[3] * d[1] - xc[4] * d[0];
    } else {
        u[0] = xc[0] - xc[2] * d[1];
        u[1] = xc[1] + xc[2] * d[0];
    }
}
// out[CS] = P_d^T * f[BS]   (forces and their moments about the centre)
template <int BS, int CS>
__device__ __forceinline__ void rbm_apply_t(const double (&d)[3], const double (&f)[BS], double (&out)[CS])
{
    if constexpr (BS == 3) {
        out[0] = f[0]; out[1] = f[1]; out[2] = f[2];
        out[3] = d[1] * f[2] - d[2] * f[1];
        out[4] = d[2] * f[0] - d[0] * f[2];
        out[5] = d[0] * f[1] - d[1] * f[0];
    } else {
        out[0] = f[0]; out[1] = f[1];
        out[2] = d[0] * f[1] - d[1] * f[0];
    }
}
// box -> parent box: child dofs (u, theta) = T * parent dofs (U, Theta),  T = [I, R(d); 0, I]
template <int CS>
__device__ __forceinline__ void box_apply(const double (&d)[3], const double* X, double (&x)[CS])
{
    if constexpr (CS == 6) {
        x[0] = X[0] + X[4] * d[2] - X[5] * d[1];
        x[1] = X[1] + X[5] * d[0] - X[3] * d[2];
        x[2] = X[2] + X[3] * d[1] - X[4] * d[0];
        x[3] = X[3]; x[4] = X[4]; x[5] = X[5];
    } else {
        x[0] = X[0] - X[2] * d[1];
        x[1] = X[1] + X[2] * d[0];
        x[2] = X[2];
    }
}
template <int CS>
__device__ __forceinline__ void box_apply_t(const double (&d)[3], const double* f, double (&F)[CS])
{
    if constexpr (CS == 6) {
        F[0] = f[0]; F[1] = f[1]; F[2] = f[2];
        F[3] = f[3] + d[1] * f[2] - d[2] * f[1];
        F[4] = f[4] + d[2] * f[0] - d[0] * f[2];
        F[5] = f[5] + d[0] * f[1] - d[1] * f[0];
    } else {
        F[0] = f[0]; F[1] = f[1];
        F[2] = f[2] + d[0] * f[1] - d[1] * f[0];
    }
}
template <int CS>
__device__ __forceinline__ void box_matrix(const double (&d)[3], double (&T)[CS * CS])
{
#pragma unroll
    for (int k = 0; k < CS * CS; k++) T[k] = 0.0;
#pragma unroll
    for (int r = 0; r < CS; r++) T[r * CS + r] = 1.0;
    if constexpr (CS == 6) {
        T[0 * 6 + 4] = d[2];  T[0 * 6 + 5] = -d[1];
        T[1 * 6 + 3] = -d[2]; T[1 * 6 + 5] = d[0];
        T[2 * 6 + 3] = d[1];  T[2 * 6 + 4] = -d[0];
    } else {
        T[0 * 3 + 2] = -d[1];
        T[1 * 3 + 2] = d[0];
    }
}
// geometry of a box level: centre of box b
// (lo: corner of the GLOBAL grid; off: origin of this domain's sub-grid in it -- the centre of a box comes out of the same
// expression on every domain that holds the box, bit for bit)
struct BoxGeom {
    double lo[3], H;
    int off[3];
};
__device__ __forceinline__ void box_centre(const Grid3& g, const BoxGeom& geo, int b, double (&c)[3])
{
    c[0] = geo.lo[0] + ((b % g.nx + geo.off[0]) + 0.5) * geo.H;
    c[1] = geo.lo[1] + (((b / g.nx) % g.ny + geo.off[1]) + 0.5) * geo.H;
    c[2] = geo.lo[2] + ((b / (g.nx * g.ny) + geo.off[2]) + 0.5) * geo.H;
}
template <int PD>
__device__ __forceinline__ void node_offset(const double* coords, int v, const double (&c)[3], double (&d)[3])
{
    d[0] = coords[(size_t)v * PD] - c[0];
    d[1] = coords[(size_t)v * PD + 1] - c[1];
    d[2] = (PD == 3) ? coords[(size_t)v * PD + 2] - c[2] : 0.0;
}

// level-1 operator  A_1 = P^T K P, one wavefront per box: lane = node of the box.  A first sweep
// over the lane's block row classifies every block by the stencil slot of its column's box (one
// gather of agg[] per block) and parks the slot ids in LDS, four to a word, lane-private.  Then for
// each of the 27 slots every lane sums P_i^T (w K_ij w) P_j over its matching blocks (the values
// are read exactly once) and a fixed shuffle tree adds the lanes: bitwise reproducible, no atomics.
// Boxes with more than 64 nodes and rows longer than MG_ROW_CHUNK blocks take further passes that
// add to the stencil entries already written (the wavefront owns its box).
constexpr int MG_ROW_CHUNK = 128;
template <int BS, int CS, int PD>
__global__ void __launch_bounds__(256) k_galerkin_fine_wave(Grid3 g, BoxGeom geo, const double* __restrict__ coords,
                                                            const int* __restrict__ box_ptr,
                                                            const int* __restrict__ box_nodes, const int* __restrict__ agg,
                                                            const int* __restrict__ rowlen, const int* __restrict__ boff,
                                                            const int* __restrict__ bcol, const double* __restrict__ val,
                                                            const unsigned char* __restrict__ w, double* __restrict__ st,
                                                            int* __restrict__ err, const int* __restrict__ box_list, int nlist)
{
    constexpr int BB = BS * BS, CC = CS * CS;
    __shared__ unsigned int sslot[4][MG_ROW_CHUNK / 4][64];
    const int nb = g.nb();
    const int widx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    // (box_list: the boxes this launch handles -- the ones k_galerkin_fine_quad does not; null: every box)
    if (widx >= (box_list ? nlist : nb)) return;   // wave-uniform
    const int A = box_list ? box_list[widx] : widx;
    unsigned int (*my)[64] = sslot[threadIdx.x >> 6];
    const int ax = A % g.nx, ay = (A / g.nx) % g.ny, az = A / (g.nx * g.ny);
    const int p0 = box_ptr[A], p1 = box_ptr[A + 1];
    double cA[3];
    box_centre(g, geo, A, cA);
    bool first = true;
    for (int pbase = p0; pbase < p1 || first; pbase += 64) {
        const bool have = pbase + lane < p1;
        const int i = have ? box_nodes[pbase + lane] : 0;
        const int len = have ? rowlen[i] : 0, ln = i & 63;
        const size_t o0 = (size_t)boff[i >> 6];
        double di[3] = {0.0, 0.0, 0.0}, wi[BS];
        if (have) node_offset<PD>(coords, i, cA, di);
#pragma unroll
        for (int r = 0; r < BS; r++) wi[r] = (have && w[(size_t)BS * i + r]) ? 1.0 : 0.0;
        int maxlen = len;
        for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
        for (int j0 = 0; j0 < maxlen || first; j0 += MG_ROW_CHUNK) {
            const int j1 = min(len, j0 + MG_ROW_CHUNK);
            // classify
            for (int jw = j0; jw < j1; jw += 4) {
                unsigned int word = 0xffffffffu;
                for (int k = 0; k < 4 && jw + k < j1; k++) {
                    const int B = agg[bcol[(o0 + jw + k) * 64 + ln]];
                    const int dx = B % g.nx - ax, dy = (B / g.nx) % g.ny - ay, dz = B / (g.nx * g.ny) - az;
                    unsigned int sl = 0xffu;
                    if (abs(dx) > 1 || abs(dy) > 1 || abs(dz) > 1) atomicExch(err, 1);   // coupled nodes must sit in adjacent boxes
                    else sl = (unsigned)((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1));
                    word = (word & ~(0xffu << (8 * k))) | (sl << (8 * k));
                }
                my[(jw - j0) >> 2][lane] = word;
            }
            for (int s = 13; s < 27; s++) {   // the centre and the 13 "upper" neighbours; k_galerkin_mirror fills the other half
                int T = -1;
                const bool inside = neighbour_box(g, A, s, T);   // wave-uniform
                double acc[CC];
#pragma unroll
                for (int e = 0; e < CC; e++) acc[e] = 0.0;
                if (inside) {
                    double cT[3];
                    box_centre(g, geo, T, cT);
                    for (int jw = j0; jw < j1; jw += 4) {
                        const unsigned int word = my[(jw - j0) >> 2][lane];
                        for (int k = 0; k < 4; k++) {
                            if (((word >> (8 * k)) & 0xffu) != (unsigned)s) continue;
                            const int j = jw + k;
                            const int c = bcol[(o0 + j) * 64 + ln];
                            double dj[3], KP[BS * CS], kw[BB];
                            node_offset<PD>(coords, c, cT, dj);
#pragma unroll
                            for (int r = 0; r < BS; r++)
#pragma unroll
                                for (int q = 0; q < BS; q++)
                                    kw[r * BS + q] = wi[r] * (double)w[(size_t)BS * c + q] * val[((o0 + j) * BB + r * BS + q) * 64 + ln];
                            // KP = (w K w) P_j : row r of K applied to the modes of node j
#pragma unroll
                            for (int r = 0; r < BS; r++) {
                                double f[BS], m[CS];
#pragma unroll
                                for (int q = 0; q < BS; q++) f[q] = kw[r * BS + q];
                                rbm_apply_t<BS, CS>(dj, f, m);
#pragma unroll
                                for (int cc = 0; cc < CS; cc++) KP[r * CS + cc] = m[cc];
                            }
                            // acc += P_i^T KP : column cc of KP seen as a force on node i
#pragma unroll
                            for (int cc = 0; cc < CS; cc++) {
                                double f[BS], m[CS];
#pragma unroll
                                for (int r = 0; r < BS; r++) f[r] = KP[r * CS + cc];
                                rbm_apply_t<BS, CS>(di, f, m);
#pragma unroll
                                for (int aa = 0; aa < CS; aa++) acc[aa * CS + cc] += m[aa];
                            }
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < CC; e++) {
                    double a = acc[e];
                    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
                    if (lane == 0) {
                        double* dst = &st[((size_t)s * CC + e) * nb + A];
                        *dst = first ? a : *dst + a;
                    }
                }
            }
            first = false;
        }
    }
}

// ---- the same product with full-line reads of the matrix (round 6) ---------------------------------------------------------------
// A box's rows sit in as many 64-row slices as it has lines of nodes, four rows to a line on a regular grid: the wavefront-per-box
// kernel above uses a quarter of every 128-B line it touches and the four wavefronts that share a line drift apart -- 16.9 GB fetched
// for the 6.8 GB matrix of config 3, 6.9 ms (profiles/r06_pmc_mg_setup.txt).  Here a workgroup takes FOUR consecutive boxes whose 256
// nodes, sorted by id, are runs of 16 consecutive ids of which every box holds four in order (k_galerkin_groups finds those groups
// once per mesh; the other boxes keep the kernel above): lane = position in the sorted list, so 16 consecutive lanes read 16
// consecutive rows -- whole lines -- and a box's lanes are the same four of every 16.  Per slot the 36 sums of a lane are added over
// its box's lanes by a fixed tree (quad, then the four lines of the wavefront, then the four wavefronts through LDS): bitwise
// reproducible, no atomics.  gnodes[group * 256 + position] = node id.
constexpr int MG_GROUP = 4;
__global__ void __launch_bounds__(256) k_galerkin_groups(int ngroups_cand, int groups_per_row, int nx, const int* __restrict__ box_ptr,
                                                         const int* __restrict__ box_nodes, int* __restrict__ gnodes,
                                                         unsigned char* __restrict__ ok)
{
    const int gi = blockIdx.x;
    if (gi >= ngroups_cand) return;
    const int row = gi / groups_per_row, gq = gi % groups_per_row, A0 = row * nx + MG_GROUP * gq;
    __shared__ int nodes[256], sorted[256];
    __shared__ int bad;
    const int t = threadIdx.x, b = t >> 6, k = t & 63;
    if (t == 0) bad = 0;
    __syncthreads();
    const int p0 = box_ptr[A0 + b], cnt = box_ptr[A0 + b + 1] - p0;
    if (cnt != 64) bad = 1;
    __syncthreads();
    if (bad) { if (t == 0) ok[gi] = 0; return; }
    const int v = box_nodes[p0 + k];
    nodes[t] = v;
    __syncthreads();
    int rank = k;                                   // (every box list is ascending)
    for (int ob = 0; ob < MG_GROUP; ob++) {
        if (ob == b) continue;
        int lo = 0, hi = 64;                        // lower bound of v in box ob's list
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (nodes[ob * 64 + mid] < v) lo = mid + 1; else hi = mid; }
        rank += lo;
    }
    sorted[rank] = v;
    __syncthreads();
    {   // position t: the run of 16 it belongs to is consecutive ids, its quarter of the run is box (t >> 2) & 3's
        const int q0 = t & ~15, r = t & 15;
        if (sorted[t] != sorted[q0] + r) bad = 1;
        if (rank >= 0 && ((rank >> 2) & 3) != b) bad = 1;
    }
    __syncthreads();
    if (bad) { if (t == 0) ok[gi] = 0; return; }
    gnodes[(size_t)gi * 256 + t] = sorted[t];
    if (t == 0) ok[gi] = 1;
}

constexpr int MG_QUAD_CHUNK = 64;   // blocks of a row per round of k_galerkin_fine_quad (hex8 rows have 27)
template <int BS, int CS, int PD>
__global__ void __launch_bounds__(256) k_galerkin_fine_quad(Grid3 g, BoxGeom geo, const double* __restrict__ coords, int ngroups,
                                                            const int* __restrict__ group_box0, const int* __restrict__ group_slot,
                                                            const int* __restrict__ gnodes, const int* __restrict__ agg,
                                                            const int* __restrict__ rowlen, const int* __restrict__ boff,
                                                            const int* __restrict__ bcol, const double* __restrict__ val,
                                                            const unsigned char* __restrict__ w, double* __restrict__ st,
                                                            int* __restrict__ err)
{
    constexpr int BB = BS * BS, CC = CS * CS, JC = MG_QUAD_CHUNK;
    // per lane (column = thread): the slot of every block of the chunk, then the blocks ordered by slot and where every slot starts --
    // a slot's pass walks only its own blocks (the wavefront-per-box kernel scans all of them in every pass and runs the body for
    // every block that ANY lane matches: 14 x 15 bodies per wavefront on a regular grid against 76 here)
    __shared__ unsigned char sl[JC][256], ord[JC][256], beg[16][256];
    __shared__ double red[4][MG_GROUP][CC];
    __shared__ int s_maxlen[4];
    const int gi = blockIdx.x;
    if (gi >= ngroups) return;
    const int nb = g.nb();
    const int t = threadIdx.x, wv = t >> 6, lane = t & 63, bq = (lane >> 2) & 3;
    const int A = group_box0[gi] + bq;                               // this lane's box
    const int ax = A % g.nx, ay = (A / g.nx) % g.ny, az = A / (g.nx * g.ny);
    double cA[3];
    box_centre(g, geo, A, cA);
    const int i = gnodes[(size_t)group_slot[gi] * 256 + t];
    const int len = rowlen[i], ln = i & 63;
    const size_t o0 = (size_t)boff[i >> 6];
    double di[3], wi[BS];
    node_offset<PD>(coords, i, cA, di);
#pragma unroll
    for (int r = 0; r < BS; r++) wi[r] = w[(size_t)BS * i + r] ? 1.0 : 0.0;
    int maxlen = len;
    for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
    if (lane == 0) s_maxlen[wv] = maxlen;
    __syncthreads();
    maxlen = max(max(s_maxlen[0], s_maxlen[1]), max(s_maxlen[2], s_maxlen[3]));   // (the four wavefronts walk the chunks together)
    bool first = true;
    for (int j0 = 0; j0 < maxlen || first; j0 += JC) {
        const int nj = max(0, min(len - j0, JC));
        // classify: the stencil slot of every block's box (slots below 13 are the mirror's: marked 0xff with the far ones)
        // (nine blocks at a time: their column ids, then the boxes of those columns -- two round trips per nine blocks, not per block;
        // the kernel is a chain of latencies, not of bytes)
        const int njw = min(maxlen - j0, JC);   // workgroup-uniform
        for (int kb = 0; kb < njw; kb += 9) {
            int cc[9], Bx[9];
#pragma unroll
            for (int u = 0; u < 9; u++) cc[u] = (kb + u < nj) ? bcol[(o0 + j0 + kb + u) * 64 + ln] : -1;
#pragma unroll
            for (int u = 0; u < 9; u++) Bx[u] = (cc[u] >= 0) ? agg[cc[u]] : 0;
#pragma unroll
            for (int u = 0; u < 9; u++) {
                if (kb + u >= nj) continue;
                const int B = Bx[u];
                const int dx = B % g.nx - ax, dy = (B / g.nx) % g.ny - ay, dz = B / (g.nx * g.ny) - az;
                unsigned sv = 0xffu;
                if (abs(dx) > 1 || abs(dy) > 1 || abs(dz) > 1) atomicExch(err, 1);   // coupled nodes must sit in adjacent boxes
                else sv = (unsigned)((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1));
                sl[kb + u][t] = (unsigned char)sv;
            }
        }
        // the lane's blocks in slot order
        int pos = 0;
        for (int s = 13; s < 27; s++) {
            beg[s - 13][t] = (unsigned char)pos;
            for (int k = 0; k < nj; k++)
                if (sl[k][t] == (unsigned char)s) ord[pos++][t] = (unsigned char)k;
        }
        beg[14][t] = (unsigned char)pos;
        for (int s = 13; s < 27; s++) {
            int T = -1;
            const bool inside = neighbour_box(g, A, s, T);   // per lane (the four boxes differ in x)
            // S = sum_j (w K_ij w) P_j over the slot's blocks (BS x CS): P_i^T is applied once, behind the loop -- 18 running sums per
            // lane instead of 36, and the kernel, a chain of latencies, fits four wavefronts per SIMD instead of two
            double S[BS * CS];
#pragma unroll
            for (int e = 0; e < BS * CS; e++) S[e] = 0.0;
            double cT[3] = {0.0, 0.0, 0.0};
            if (inside) box_centre(g, geo, T, cT);
            const int m0 = beg[s - 13][t], m1 = inside ? beg[s - 12][t] : m0;
            // (round 6 tried the loads of two blocks at a time and a software pipeline that asks for block m + 1 before block m is worked
            // on: both push the kernel to 200-256 registers, two wavefronts per SIMD or one, and lose -- 15.4 and 11.6 ms of set-up
            // against 10.6 with this plain loop at three wavefronts per SIMD)
            for (int m = m0; __any(m < m1); m++) {
                if (m >= m1) continue;
                const int j = j0 + ord[m][t];
                const int c = bcol[(o0 + j) * 64 + ln];
                double dj[3];
                node_offset<PD>(coords, c, cT, dj);
#pragma unroll
                for (int r = 0; r < BS; r++) {
                    double f[BS], mm[CS];
#pragma unroll
                    for (int q = 0; q < BS; q++) f[q] = wi[r] * (double)w[(size_t)BS * c + q] * val[((o0 + j) * BB + r * BS + q) * 64 + ln];
                    rbm_apply_t<BS, CS>(dj, f, mm);   // row r of (w K w) applied to the modes of node j
#pragma unroll
                    for (int c2 = 0; c2 < CS; c2++) S[r * CS + c2] += mm[c2];
                }
            }
            // P_i^T S column by column; the box's lanes of this wavefront added by a fixed tree: lanes (lane & 0x33) == 0 end up with
            // the sums of boxes 0..3 (lanes 0, 4, 8, 12)
#pragma unroll
            for (int c2 = 0; c2 < CS; c2++) {
                double f[BS], mm[CS];
#pragma unroll
                for (int r = 0; r < BS; r++) f[r] = S[r * CS + c2];
                rbm_apply_t<BS, CS>(di, f, mm);
#pragma unroll
                for (int aa = 0; aa < CS; aa++) {
                    double a = mm[aa];
                    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                    if ((lane & 0x33) == 0) red[wv][bq][aa * CS + c2] = a;
                }
            }
            __syncthreads();
            if (t < MG_GROUP * CC) {
                const int b = t / CC, e = t % CC;
                const double a = (red[0][b][e] + red[1][b][e]) + (red[2][b][e] + red[3][b][e]);
                double* dst = &st[((size_t)s * CC + e) * nb + group_box0[gi] + b];
                *dst = first ? a : *dst + a;
            }
            __syncthreads();
        }
        first = false;
    }
}

// K is symmetric, so A_1(T, A) = A_1(A, T)^T: stencil slot s < 13 of box A is the transpose of slot 26 - s of the neighbour
// that slot points at (zero where there is none) -- the fine-level product computes 14 of the 27 slots
template <int CS>
__global__ void __launch_bounds__(256) k_galerkin_mirror(Grid3 g, double* __restrict__ st)
{
    constexpr int CC = CS * CS;
    const int nb = g.nb();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)13 * CC * nb) return;
    const int A = (int)(t % nb), e = (int)((t / nb) % CC), s = (int)(t / ((size_t)nb * CC));
    int T = -1;
    double v = 0.0;
    if (neighbour_box(g, A, s, T)) v = st[((size_t)(26 - s) * CC + (e % CS) * CS + e / CS) * nb + T];
    st[((size_t)s * CC + e) * nb + A] = v;
}

// A_{l+1} = T^T A_l T between two box levels (children = 2x2x2 boxes), T = [I, R(d); 0, I]
template <int CS>
__global__ void __launch_bounds__(256) k_galerkin_box(Grid3 gf, BoxGeom geof, Grid3 gc, BoxGeom geoc, Shift3 sh,
                                                      const double* __restrict__ stf, double* __restrict__ stc)
{
    constexpr int CC = CS * CS;
    const int nbc = gc.nb(), nbf = gf.nb();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 27 * nbc) return;
    const int slot = t / nbc, A = t - slot * nbc;
    int T = -1;
    const bool inside = neighbour_box(gc, A, slot, T);
    double acc[CC];
#pragma unroll
    for (int e = 0; e < CC; e++) acc[e] = 0.0;
    if (inside) {
        const int ax = A % gc.nx, ay = (A / gc.nx) % gc.ny, az = A / (gc.nx * gc.ny);
        double cA[3], cT[3];
        box_centre(gc, geoc, A, cA);
        box_centre(gc, geoc, T, cT);
        for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
            const int fx = 2 * ax + cx - sh.x, fy = 2 * ay + cy - sh.y, fz = 2 * az + cz - sh.z;
            if (fx < 0 || fy < 0 || fz < 0 || fx >= gf.nx || fy >= gf.ny || fz >= gf.nz) continue;
            const int a = fx + gf.nx * (fy + gf.ny * fz);
            double ca[3], da[3], Ta[CC];
            box_centre(gf, geof, a, ca);
            for (int k = 0; k < 3; k++) da[k] = ca[k] - cA[k];
            box_matrix<CS>(da, Ta);
            for (int s = 0; s < 27; s++) {
                int b;
                if (!neighbour_box(gf, a, s, b)) continue;
                const int bx = (b % gf.nx + sh.x) >> 1, by = ((b / gf.nx) % gf.ny + sh.y) >> 1, bz = (b / (gf.nx * gf.ny) + sh.z) >> 1;
                if (bx + gc.nx * (by + gc.ny * bz) != T) continue;
                double cb[3], db[3], Tb[CC], ST[CC];
                box_centre(gf, geof, b, cb);
                for (int k = 0; k < 3; k++) db[k] = cb[k] - cT[k];
                box_matrix<CS>(db, Tb);
                for (int r = 0; r < CS; r++)
                    for (int c = 0; c < CS; c++) {
                        double v = 0.0;
                        for (int q = 0; q < CS; q++) v += stf[((size_t)s * CC + r * CS + q) * nbf + a] * Tb[q * CS + c];
                        ST[r * CS + c] = v;
                    }
                for (int r = 0; r < CS; r++)
                    for (int c = 0; c < CS; c++) {
                        double v = 0.0;
                        for (int q = 0; q < CS; q++) v += Ta[q * CS + r] * ST[q * CS + c];
                        acc[r * CS + c] += v;
                    }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < CC; e++) stc[((size_t)slot * CC + e) * nbc + A] = acc[e];
}

// inverse of the CS x CS diagonal block of every box (block-Jacobi smoother): dinv[r*CS+c][b].
// Point Jacobi is not enough here: translations and rotations of one box are strongly coupled.
// Rank-deficient blocks (boxes with fewer than three non-collinear free nodes, or none) get the
// same relative shift as the coarsest matrix.
template <int CS>
// (diag: the CS x CS diagonal blocks [CS*CS][nb] -- slot 13 of the stencil, or, on a partitioned mesh, its sum over the domains)
__global__ void __launch_bounds__(256) k_level_dinv(int nb, const double* __restrict__ diag, double* __restrict__ dinv)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    double D[CS][2 * CS];
#pragma unroll
    for (int r = 0; r < CS; r++)
#pragma unroll
        for (int c = 0; c < CS; c++) {
            D[r][c] = diag[((size_t)r * CS + c) * nb + b];
            D[r][CS + c] = (r == c) ? 1.0 : 0.0;
        }
    double tr = 0.0;
#pragma unroll
    for (int r = 0; r < CS; r++) tr = fmax(tr, fabs(D[r][r]));
#pragma unroll
    for (int r = 0; r < CS; r++) D[r][r] = (D[r][r] == 0.0) ? ((tr > 0.0) ? tr : 1.0) : D[r][r] * (1.0 + 1e-10);
#pragma unroll
    for (int k = 0; k < CS; k++) {
        const double piv = 1.0 / D[k][k];
#pragma unroll
        for (int c = 0; c < 2 * CS; c++) D[k][c] *= piv;
#pragma unroll
        for (int r = 0; r < CS; r++) {
            if (r == k) continue;
            const double f = D[r][k];
#pragma unroll
            for (int c = 0; c < 2 * CS; c++) D[r][c] -= f * D[k][c];
        }
    }
#pragma unroll
    for (int r = 0; r < CS; r++)
#pragma unroll
        for (int c = 0; c < CS; c++) dinv[((size_t)r * CS + c) * nb + b] = D[r][CS + c];
}

// x_b = omega * Dinv_b r_b      and      x_b += omega * Dinv_b t_b   (t already holds r - A x)
// omega is read from device memory: it comes from the eigenvalue estimate of the set-up
// (boxes b0 .. b1-1 only: a partitioned run splits the level-1 work among the domains)
template <int CS, bool ADD>
__global__ void __launch_bounds__(256) k_block_smooth(int nb, int b0, int b1, const double* __restrict__ omega,
                                                      const double* __restrict__ dinv, const double* __restrict__ r,
                                                      double* __restrict__ x)
{
    const int b = b0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= b1) return;
    const double om = omega ? *omega : 1.0;
    double v[CS];
#pragma unroll
    for (int q = 0; q < CS; q++) v[q] = r[(size_t)b * CS + q];
#pragma unroll
    for (int q = 0; q < CS; q++) {
        double a = 0.0;
#pragma unroll
        for (int p = 0; p < CS; p++) a += dinv[((size_t)q * CS + p) * nb + b] * v[p];
        if (ADD) x[(size_t)b * CS + q] += om * a; else x[(size_t)b * CS + q] = om * a;
    }
}

// deterministic start vector of the power iteration
__global__ void __launch_bounds__(256) k_hash_fill(size_t n, double* __restrict__ x)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        unsigned int h = (unsigned int)i * 2654435761u + 12345u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = 0.5 + (double)(h & 0xffffu) / 65536.0;
    }
}
// the same on a box level of a partitioned mesh: the value of a box entry depends on the box's GLOBAL index only, so that every
// domain holding the box starts the power iteration from the same number
template <int CS>
__global__ void __launch_bounds__(256) k_hash_fill_box(Grid3 g, Shift3 off, Grid3 gg, double* __restrict__ x)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.nb()) return;
    const int ix = b % g.nx + off.x, iy = (b / g.nx) % g.ny + off.y, iz = b / (g.nx * g.ny) + off.z;
    const size_t gb = (size_t)ix + (size_t)gg.nx * ((size_t)iy + (size_t)gg.ny * iz);
    for (int q = 0; q < CS; q++) {
        unsigned int h = (unsigned int)(gb * CS + q) * 2654435761u + 12345u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[(size_t)b * CS + q] = 0.5 + (double)(h & 0xffffu) / 65536.0;
    }
}
// out = sum x_i^2 (single workgroup; coarse levels only, set-up time); own (may be null): [n / cs] 1 where the entry's box counts
// (partitioned mesh: a shared box is counted by its owner, the sums are then all-reduced)
__global__ void __launch_bounds__(1024) k_norm2_small(size_t n, const double* __restrict__ x, double* __restrict__ out,
                                                      const unsigned char* __restrict__ own = nullptr, int cs = 1)
{
    __shared__ double sh[16];
    double a = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) a += (!own || own[i / cs]) ? x[i] * x[i] : 0.0;
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += sh[k];
        *out = t;
    }
}
// damping from the estimate lambda ~ sqrt(n2[1] / n2[0]) of the largest eigenvalue of Dinv A:
// omega = 4 / (3 * 1.1 * lambda)   (the power iteration approaches lambda from below: 10 % margin)
__global__ void k_set_omega(const double* __restrict__ n2, double* __restrict__ omega)
{
    const double lam = (n2[0] > 0.0 && n2[1] > 0.0) ? sqrt(n2[1] / n2[0]) : 2.0;
    *omega = fmin(1.0, 4.0 / (3.0 * 1.1 * fmax(lam, 1.0)));
}

// ---- cycle kernels ------------------------------------------------------------------------------------
// The cycle's copy of a large level's stencil in single precision (MgLevel::stf): k_box_residual is a stream of the stencil (400 MB in
// doubles on level 1 of config 3, twice per application of the preconditioner).  Two entries of a block row per 8-byte word, box index
// fastest -- the same bytes per load instruction as the doubles (4-byte loads, one entry each, left the kernel with half the bytes in
// flight: 2x slower than the doubles it was meant to beat).  Arithmetic in doubles.
template <int CS>
__global__ void __launch_bounds__(256) k_stencil_to_float2(int nb, const double* __restrict__ st, float2* __restrict__ stf)
{
    constexpr int CC = CS * CS, HP = CC / 2;
    const size_t n = (size_t)27 * HP * nb;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i % nb, sk = i / nb, s = sk / HP, k = sk % HP;
        stf[i] = make_float2((float)st[(s * CC + 2 * k) * nb + b], (float)st[(s * CC + 2 * k + 1) * nb + b]);
    }
}
template <int CS, bool OWN>
__global__ void __launch_bounds__(256) k_box_residual_f2(Grid3 g, int b0, int b1, const float2* __restrict__ stf,
                                                         const double* __restrict__ x, const double* __restrict__ r,
                                                         double* __restrict__ t, const unsigned char* __restrict__ own)
{
    static_assert(CS % 2 == 0, "two entries of a block row per word");
    constexpr int HP = CS * CS / 2, HR = CS / 2;
    const int nb = g.nb();
    const int b = b0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= b1) return;
    double acc[CS];
#pragma unroll
    for (int q = 0; q < CS; q++) acc[q] = r[(size_t)b * CS + q];
    const double keep = OWN ? (own[b] ? 1.0 : 0.0) : 1.0;
    if constexpr (OWN) {
#pragma unroll
        for (int q = 0; q < CS; q++) acc[q] *= keep;
    }
    // (slot by slot, every load of the slot first.  Three slots' loads at a time, unconditional, measured 56 against 43 us on level 1 of
    // config 3; a thread per block row and box, as on the middle levels, 61: the 6-fold re-read of x at a stride of 48 bytes)
    for (int s = 0; s < 27; s++) {
        int c;
        if (!neighbour_box(g, b, s, c)) continue;
        float2 v[HP];
        double xc[CS];
#pragma unroll
        for (int k = 0; k < HP; k++) v[k] = stf[((size_t)s * HP + k) * nb + b];
#pragma unroll
        for (int p = 0; p < CS; p++) xc[p] = x[(size_t)c * CS + p];
#pragma unroll
        for (int q = 0; q < CS; q++)
#pragma unroll
            for (int h = 0; h < HR; h++) acc[q] -= (double)v[q * HR + h].x * xc[2 * h] + (double)v[q * HR + h].y * xc[2 * h + 1];
    }
#pragma unroll
    for (int q = 0; q < CS; q++) t[(size_t)b * CS + q] = acc[q];
}
// t = r - A x on a box level (27-point block stencil, thread per box)
// (OWN: partitioned mesh -- st is this domain's sub-assembled part of the operator and r counts on the owner of a shared box only,
// so that the halo sum of t over the domains is r - A x.  A template parameter, not a nullable pointer: with the run-time test in
// front of the slot loop the compiler kept 34 instead of 72 registers' worth of loads in flight and the kernel took 365 instead
// of 78 us on the level-1 grid of config 3)
template <int CS, bool OWN>
__global__ void __launch_bounds__(256) k_box_residual(Grid3 g, int b0, int b1, const double* __restrict__ st,
                                                      const double* __restrict__ x, const double* __restrict__ r,
                                                      double* __restrict__ t, const unsigned char* __restrict__ own)
{
    constexpr int CC = CS * CS;
    const int nb = g.nb();
    const int b = b0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= b1) return;
    double acc[CS];
#pragma unroll
    for (int q = 0; q < CS; q++) acc[q] = r[(size_t)b * CS + q];
    const double keep = OWN ? (own[b] ? 1.0 : 0.0) : 1.0;   // (a factor, not a branch: see above)
    if constexpr (OWN) {
#pragma unroll
        for (int q = 0; q < CS; q++) acc[q] *= keep;
    }
    for (int s = 0; s < 27; s++) {
        int c;
        if (!neighbour_box(g, b, s, c)) continue;
#pragma unroll
        for (int q = 0; q < CS; q++)
#pragma unroll
            for (int p = 0; p < CS; p++) acc[q] -= st[((size_t)s * CC + q * CS + p) * nb + b] * x[(size_t)c * CS + p];
    }
#pragma unroll
    for (int q = 0; q < CS; q++) t[(size_t)b * CS + q] = acc[q];
}

// Middle levels (a few thousand boxes: too few for a thread per box, and the 32-lanes-per-box kernel below reads every 8-byte entry
// from a line of its own -- lanes = slots, 36 nb entries apart): a thread per (row of the block, box), boxes along the lanes, so
// every load instruction reads 512 contiguous bytes and a box's CS rows are CS wavefronts; slots outside the grid are read at the
// box itself and weighted 0 (no branch around a load: nine slots' loads in flight)
#ifndef FEM_EXP_ROWS_GROUP
#define FEM_EXP_ROWS_GROUP 9   // slots whose loads are in flight together (27 = 3 x 9)
#endif
template <int CS, bool OWN>
__global__ void __launch_bounds__(256) k_box_residual_rows(Grid3 g, int nbp, const double* __restrict__ st, const double* __restrict__ x,
                                                           const double* __restrict__ r, double* __restrict__ t,
                                                           const unsigned char* __restrict__ own)
{
    constexpr int CC = CS * CS;
    const int nb = g.nb();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = tid / nbp, b = tid % nbp;
    if (q >= CS || b >= nb) return;
    double acc = (!OWN || own[b]) ? r[(size_t)b * CS + q] : 0.0;
    const int ix = b % g.nx, iy = (b / g.nx) % g.ny, iz = b / (g.nx * g.ny);
#pragma unroll 1
    for (int s0 = 0; s0 < 27; s0 += FEM_EXP_ROWS_GROUP)
#pragma unroll
    for (int s = s0; s < s0 + FEM_EXP_ROWS_GROUP; s++) {
        const int jx = ix + s % 3 - 1, jy = iy + (s / 3) % 3 - 1, jz = iz + s / 9 - 1;
        const bool in = jx >= 0 && jy >= 0 && jz >= 0 && jx < g.nx && jy < g.ny && jz < g.nz;
        const int c = in ? jx + g.nx * (jy + g.ny * jz) : b;
        const double wgt = in ? 1.0 : 0.0;
        double a = 0.0;
#pragma unroll
        for (int p = 0; p < CS; p++) a += st[((size_t)s * CC + q * CS + p) * nb + b] * x[(size_t)c * CS + p];
        acc -= wgt * a;
    }
    t[(size_t)b * CS + q] = acc;
}

// same for the small levels, which are latency-bound with one thread per box: 32 lanes per
// box (one stencil slot each, 27 active), fixed shuffle tree over the slots
template <int CS, bool OWN>
__global__ void __launch_bounds__(256) k_box_residual_slots(Grid3 g, int b0, int b1, const double* __restrict__ st,
                                                            const double* __restrict__ x, const double* __restrict__ r,
                                                            double* __restrict__ t, const unsigned char* __restrict__ own)
{
    constexpr int CC = CS * CS;
    const int nb = b1;   // boxes b0 .. b1-1 (the stencil arrays keep the stride of the whole level)
    const int nball = g.nb();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = b0 + (gid >> 5), s = gid & 31;
    double acc[CS];
#pragma unroll
    for (int q = 0; q < CS; q++) acc[q] = 0.0;
    int c;
    if (b < nb && s < 27 && neighbour_box(g, b, s, c)) {
#pragma unroll
        for (int q = 0; q < CS; q++)
#pragma unroll
            for (int p = 0; p < CS; p++) acc[q] -= st[((size_t)s * CC + q * CS + p) * nball + b] * x[(size_t)c * CS + p];
    }
#pragma unroll
    for (int q = 0; q < CS; q++)
        for (int o = 16; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, 32);
    if (b < nb && s == 0) {
#pragma unroll
        for (int q = 0; q < CS; q++) t[(size_t)b * CS + q] = ((!OWN || own[b]) ? r[(size_t)b * CS + q] : 0.0) + acc[q];
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
// z += omega * dinv * (r - Kz)   (fine level; kz = K z)
__global__ void __launch_bounds__(256) k_fine_post(size_t n, double omega, const double* __restrict__ dinv,
                                                   const double* __restrict__ r, const double* __restrict__ kz,
                                                   double* __restrict__ z)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) z[i] += omega * dinv[i] * (r[i] - kz[i]);
}

// offsets of the nodes from the centres of their boxes, single precision: by node (prolongation) and in the order of the box lists
// (restriction) -- geometry alone, kept with the aggregates on the mesh's pattern
template <int PD>
__global__ void __launch_bounds__(256) k_node_offsets(int nn, Grid3 g, BoxGeom geo, const double* __restrict__ coords,
                                                      const int* __restrict__ agg, float* __restrict__ drel_v)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    double c[3], d[3];
    box_centre(g, geo, agg[v], c);
    node_offset<PD>(coords, v, c, d);
    for (int q = 0; q < 3; q++) drel_v[(size_t)v * 3 + q] = (float)d[q];
}
__global__ void __launch_bounds__(256) k_node_offsets_by_list(int nn, const int* __restrict__ box_nodes, const float* __restrict__ drel_v,
                                                              float* __restrict__ drel_p)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nn) return;
    const int v = box_nodes[p];
    for (int q = 0; q < 3; q++) drel_p[(size_t)p * 3 + q] = drel_v[(size_t)v * 3 + q];
}

// r_1[B] = sum_{i in B} P_i^T w_i (r_i - (K z)_i)      one wavefront per box, lanes over the box's
// nodes (ascending), fixed shuffle tree -> bitwise reproducible
// FD: the nodes' offsets from their box centres come from MgBoxes::drel_p (single precision, in the order of the box lists: one
// contiguous 12-byte read per lane) instead of the coordinates (24 bytes behind the node id) -- the transfer operators of the cycle
// are a fixed pair P, P^T either way; set-up (the Galerkin products) keeps the coordinates
template <int BS, int CS, int PD, bool FD = false>
__global__ void __launch_bounds__(256) k_restrict_fine(Grid3 g, BoxGeom geo, const double* __restrict__ coords,
                                                       const int* __restrict__ box_ptr, const int* __restrict__ box_nodes,
                                                       const unsigned char* __restrict__ w, const double* __restrict__ r,
                                                       const double* __restrict__ kz, double* __restrict__ rc,
                                                       const float* __restrict__ drel = nullptr)
{
    const int nb = g.nb();
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (b >= nb) return;
    double cB[3], acc[CS];
    box_centre(g, geo, b, cB);
#pragma unroll
    for (int q = 0; q < CS; q++) acc[q] = 0.0;
    for (int p = box_ptr[b] + lane; p < box_ptr[b + 1]; p += 64) {
        const int v = box_nodes[p];
        const size_t i = (size_t)v * BS;
        double f[BS], d[3], m[CS];
#pragma unroll
        for (int q = 0; q < BS; q++) f[q] = w[i + q] ? (kz ? r[i + q] - kz[i + q] : r[i + q]) : 0.0;
        if constexpr (FD) { d[0] = drel[(size_t)p * 3]; d[1] = drel[(size_t)p * 3 + 1]; d[2] = drel[(size_t)p * 3 + 2]; }
        else node_offset<PD>(coords, v, cB, d);
        rbm_apply_t<BS, CS>(d, f, m);
#pragma unroll
        for (int q = 0; q < CS; q++) acc[q] += m[q];
    }
#pragma unroll
    for (int q = 0; q < CS; q++)
        for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o);
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < CS; q++) rc[(size_t)b * CS + q] = acc[q];
    }
}

// z_i += alpha * w_i * (P_i x_1[agg(i)])
template <int BS, int CS, int PD, bool FD = false>
__global__ void __launch_bounds__(256) k_prolong_fine(int nn, Grid3 g, BoxGeom geo, const double* __restrict__ coords,
                                                      double alpha, const int* __restrict__ agg,
                                                      const unsigned char* __restrict__ w, const double* __restrict__ xc,
                                                      double* __restrict__ z, const float* __restrict__ drel = nullptr)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    const int b = agg[v];
    double c[3], d[3], u[BS];
    if constexpr (FD) { d[0] = drel[(size_t)v * 3]; d[1] = drel[(size_t)v * 3 + 1]; d[2] = drel[(size_t)v * 3 + 2]; }
    else { box_centre(g, geo, b, c); node_offset<PD>(coords, v, c, d); }
    rbm_apply<BS, CS>(d, xc + (size_t)b * CS, u);
    const size_t i = (size_t)v * BS;
#pragma unroll
    for (int q = 0; q < BS; q++)
        if (w[i + q]) z[i + q] += alpha * u[q];
}

// additive fine level, one pass: z_i = dinv_i r_i + alpha w_i (P_i x_1[agg(i)]), plus the partials of
// z.z and z.r that the CG needs next (same fixed-order two-stage reduction as the Jacobi path)
template <int BS, int CS, int PD, bool FD = false>
__global__ void __launch_bounds__(256) k_additive_finish(int nn, Grid3 g, BoxGeom geo, const double* __restrict__ coords,
                                                         double alpha, const int* __restrict__ agg,
                                                         const unsigned char* __restrict__ w, const double* __restrict__ dinv,
                                                         const double* __restrict__ r, const double* __restrict__ xc,
                                                         double* __restrict__ z, double* __restrict__ pzz,
                                                         double* __restrict__ pzr, const int* __restrict__ done,
                                                         const float* __restrict__ drel = nullptr)
{
    if (done && *done) return;
    __shared__ double sh[2][4];
    double zz = 0.0, zr = 0.0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nn; v += gridDim.x * blockDim.x) {
        const int b = agg[v];
        double c[3], d[3], u[BS];
        if constexpr (FD) { d[0] = drel[(size_t)v * 3]; d[1] = drel[(size_t)v * 3 + 1]; d[2] = drel[(size_t)v * 3 + 2]; }
        else { box_centre(g, geo, b, c); node_offset<PD>(coords, v, c, d); }
        rbm_apply<BS, CS>(d, xc + (size_t)b * CS, u);
        const size_t i = (size_t)v * BS;
#pragma unroll
        for (int q = 0; q < BS; q++) {
            const double ri = r[i + q];
            const double zi = w[i + q] ? dinv[i + q] * ri + alpha * u[q] : dinv[i + q] * ri;
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

// r_{l+1}[A] = sum over the children a of A of T_a^T t_l[a]
template <int CS>
__global__ void __launch_bounds__(256) k_restrict_box(Grid3 gf, BoxGeom geof, Grid3 gc, BoxGeom geoc, Shift3 sh,
                                                      const double* __restrict__ tf, double* __restrict__ rc)
{
    const int A = blockIdx.x * blockDim.x + threadIdx.x;
    if (A >= gc.nb()) return;
    const int ax = A % gc.nx, ay = (A / gc.nx) % gc.ny, az = A / (gc.nx * gc.ny);
    double cA[3], acc[CS];
    box_centre(gc, geoc, A, cA);
#pragma unroll
    for (int q = 0; q < CS; q++) acc[q] = 0.0;
    for (int cz = 0; cz < 2; cz++) for (int cy = 0; cy < 2; cy++) for (int cx = 0; cx < 2; cx++) {
        const int fx = 2 * ax + cx - sh.x, fy = 2 * ay + cy - sh.y, fz = 2 * az + cz - sh.z;
        if (fx < 0 || fy < 0 || fz < 0 || fx >= gf.nx || fy >= gf.ny || fz >= gf.nz) continue;
        const int a = fx + gf.nx * (fy + gf.ny * fz);
        double ca[3], d[3], F[CS];
        box_centre(gf, geof, a, ca);
        for (int k = 0; k < 3; k++) d[k] = ca[k] - cA[k];
        box_apply_t<CS>(d, tf + (size_t)a * CS, F);
#pragma unroll
        for (int q = 0; q < CS; q++) acc[q] += F[q];
    }
#pragma unroll
    for (int q = 0; q < CS; q++) rc[(size_t)A * CS + q] = acc[q];
}

// x_l[a] += alpha * T_a x_{l+1}[parent(a)]
template <int CS>
__global__ void __launch_bounds__(256) k_prolong_box(Grid3 gf, BoxGeom geof, Grid3 gc, BoxGeom geoc, Shift3 sh, double alpha,
                                                     const double* __restrict__ xc, double* __restrict__ xf)
{
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= gf.nb()) return;
    const int px = (a % gf.nx + sh.x) >> 1, py = ((a / gf.nx) % gf.ny + sh.y) >> 1, pz = (a / (gf.nx * gf.ny) + sh.z) >> 1;
    const int A = px + gc.nx * (py + gc.ny * pz);
    double ca[3], cA[3], d[3], x[CS];
    box_centre(gf, geof, a, ca);
    box_centre(gc, geoc, A, cA);
    for (int k = 0; k < 3; k++) d[k] = ca[k] - cA[k];
    box_apply<CS>(d, xc + (size_t)A * CS, x);
#pragma unroll
    for (int q = 0; q < CS; q++) xf[(size_t)a * CS + q] += alpha * x[q];
}

// ---- coarsest grid: dense inverse by Gauss-Jordan -----------------------------------------------------
template <int CS>
__global__ void __launch_bounds__(256) k_dense_from_stencil(Grid3 g, const double* __restrict__ st, double* __restrict__ M)
{
    // one workgroup per box: its CS rows of [A | I] (2n wide) are zeroed and filled by itself (a single workgroup for the whole
    // 9 MB matrix of config 3's coarsest level took 0.27 ms of every set-up)
    constexpr int CC = CS * CS;
    const int nb = g.nb(), n = nb * CS, b = blockIdx.x;
    if (b >= nb) return;
    double* rows = M + (size_t)b * CS * 2 * n;
    for (int idx = threadIdx.x; idx < CS * 2 * n; idx += blockDim.x) rows[idx] = 0.0;
    __syncthreads();
    for (int e = threadIdx.x; e < 27 * CC; e += blockDim.x) {   // (a box's 27 neighbours are distinct: one writer per entry)
        const int s = e / CC, q = (e % CC) / CS, p = e % CS;
        int c;
        if (!neighbour_box(g, b, s, c)) continue;
        rows[(size_t)q * 2 * n + c * CS + p] += st[((size_t)s * CC + q * CS + p) * nb + b];
    }
    __syncthreads();
    if (threadIdx.x < CS) {
        const int q = threadIdx.x, i = b * CS + q;
        double& dii = rows[(size_t)q * 2 * n + i];
        // a box with fewer than three non-collinear free nodes cannot carry all its rigid-body
        // modes: its rows are rank deficient (the right-hand sides P^T r stay consistent).  A
        // relative diagonal shift keeps the pivots of the unpivoted elimination away from zero;
        // the null-space components it lets through prolong to nothing.
        dii = (dii == 0.0) ? 1.0 : dii * (1.0 + 1e-10);  // == 0: empty / fully constrained box
        rows[(size_t)q * 2 * n + n + i] = 1.0;
    }
}

// Gauss-Jordan on [A | I] (n x 2n, row-major) by CS x CS pivot blocks (one box at a time), two
// launches per block so that the rank-CS update runs on the whole chip.  A is symmetric positive
// definite after the shift above: no pivot search.
//   rowk[q][t] = (D^-1 M[block rows])[q][t]      colk[p][i] = M[i][block col p]
template <int CS>
__global__ void __launch_bounds__(256) k_gj_pivot(int n, int kb, const double* __restrict__ M, double* __restrict__ rowk,
                                                  double* __restrict__ colk)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k0 = kb * CS;
    double D[CS][2 * CS];   // [D | I] -> [I | D^-1], every thread for itself (CS <= 6)
#pragma unroll
    for (int r = 0; r < CS; r++)
#pragma unroll
        for (int c = 0; c < CS; c++) { D[r][c] = M[(size_t)(k0 + r) * 2 * n + k0 + c]; D[r][CS + c] = (r == c) ? 1.0 : 0.0; }
#pragma unroll
    for (int k = 0; k < CS; k++) {
        const double piv = 1.0 / D[k][k];
#pragma unroll
        for (int c = 0; c < 2 * CS; c++) D[k][c] *= piv;
#pragma unroll
        for (int r = 0; r < CS; r++) {
            if (r == k) continue;
            const double f = D[r][k];
#pragma unroll
            for (int c = 0; c < 2 * CS; c++) D[r][c] -= f * D[k][c];
        }
    }
    if (t < 2 * n) {
        double m[CS];
#pragma unroll
        for (int p = 0; p < CS; p++) m[p] = M[(size_t)(k0 + p) * 2 * n + t];
#pragma unroll
        for (int q = 0; q < CS; q++) {
            double v = 0.0;
#pragma unroll
            for (int p = 0; p < CS; p++) v += D[q][CS + p] * m[p];
            rowk[(size_t)q * 2 * n + t] = v;
        }
    }
    if (t < n) {
#pragma unroll
        for (int p = 0; p < CS; p++) colk[(size_t)p * n + t] = M[(size_t)t * 2 * n + k0 + p];
    }
}
template <int CS>
__global__ void __launch_bounds__(256) k_gj_update(int n, int kb, double* __restrict__ M, const double* __restrict__ rowk,
                                                   const double* __restrict__ colk)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * 2 * n) return;
    const int i = idx / (2 * n), j = idx - i * 2 * n;
    const int k0 = kb * CS;
    if (i >= k0 && i < k0 + CS) { M[idx] = rowk[(size_t)(i - k0) * 2 * n + j]; return; }
    double v = M[idx];
#pragma unroll
    for (int p = 0; p < CS; p++) v -= colk[(size_t)p * n + i] * rowk[(size_t)p * 2 * n + j];
    M[idx] = v;
}

// The same elimination as ONE resident launch: workgroup b keeps block row b of [A | I] (CS x 2n doubles) in LDS from the first step
// to the last.  Step k: the owner of block row k turns it into D^-1 [row] and publishes it (global memory, two buffers by parity);
// one grid barrier (grid_barrier.h); every other workgroup subtracts its own column-k entries times the published row from its rows.
// 125 steps x (barrier + 72 KB read from another XCD's data) instead of 250 launches: the set-up of -pc_type mg lost 1.3 ms at
// config 3.  Same operations per entry as k_gj_pivot / k_gj_update, in the same order: the inverse is bit for bit theirs.
template <int CS>
__global__ void __launch_bounds__(256) k_gj_resident(int n, double* __restrict__ M, double* __restrict__ rowbuf, GridBarrier bar)
{
    extern __shared__ double R[];   // [CS][2n]
    __shared__ unsigned shu[4];
    const int b = blockIdx.x, nbk = n / CS, n2 = 2 * n;
    for (int q = 0; q < CS; q++)
        for (int t = threadIdx.x; t < n2; t += 256) R[q * n2 + t] = M[(size_t)(b * CS + q) * n2 + t];
    GridBarrierLane lane = grid_barrier_census(bar, (unsigned)nbk, shu);
    for (int kb = 0; kb < nbk; kb++) {
        const int k0 = kb * CS;
        // Columns that matter at step k: the pivot row is zero right of column k0 + CS of the identity half, and what it holds left of
        // k0 (the residue of the blocks already eliminated, ~1e-16) only ever reaches columns that are not read again -- the
        // inverse comes out of the right half bit for bit as before, from n + CS columns per step instead of 2n
        const int t_lo = k0, t_hi = n + k0 + CS;
        double* rowk = rowbuf + (size_t)(kb & 1) * CS * n2;
        if (b == kb) {
            double D[CS][2 * CS];   // [D | I] -> [I | D^-1], every thread for itself (as k_gj_pivot)
#pragma unroll
            for (int r = 0; r < CS; r++)
#pragma unroll
                for (int c = 0; c < CS; c++) { D[r][c] = R[r * n2 + k0 + c]; D[r][CS + c] = (r == c) ? 1.0 : 0.0; }
#pragma unroll
            for (int k = 0; k < CS; k++) {
                const double piv = 1.0 / D[k][k];
#pragma unroll
                for (int c = 0; c < 2 * CS; c++) D[k][c] *= piv;
#pragma unroll
                for (int r = 0; r < CS; r++) {
                    if (r == k) continue;
                    const double f = D[r][k];
#pragma unroll
                    for (int c = 0; c < 2 * CS; c++) D[r][c] -= f * D[k][c];
                }
            }
            __syncthreads();   // (every thread has read the pivot block before the rows change)
            for (int t = t_lo + threadIdx.x; t < t_hi; t += 256) {
                double m[CS];
#pragma unroll
                for (int p = 0; p < CS; p++) m[p] = R[p * n2 + t];
#pragma unroll
                for (int q = 0; q < CS; q++) {
                    double v = 0.0;
#pragma unroll
                    for (int p = 0; p < CS; p++) v += D[q][CS + p] * m[p];
                    R[q * n2 + t] = v;
                    rowk[(size_t)q * n2 + t] = v;
                }
            }
        }
        if (!grid_barrier(bar, lane, shu)) return;
        if (b != kb) {
            double c[CS][CS];   // this block row's entries of block column k
#pragma unroll
            for (int q = 0; q < CS; q++)
#pragma unroll
                for (int p = 0; p < CS; p++) c[q][p] = R[q * n2 + k0 + p];
            __syncthreads();
            for (int t = t_lo + threadIdx.x; t < t_hi; t += 256) {
                double rk[CS];
#pragma unroll
                for (int p = 0; p < CS; p++) rk[p] = rowk[(size_t)p * n2 + t];
#pragma unroll
                for (int q = 0; q < CS; q++) {
                    double v = R[q * n2 + t];
#pragma unroll
                    for (int p = 0; p < CS; p++) v -= c[q][p] * rk[p];
                    R[q * n2 + t] = v;
                }
            }
            __syncthreads();
        }
    }
    for (int q = 0; q < CS; q++)
        for (int t = threadIdx.x; t < n2; t += 256) M[(size_t)(b * CS + q) * n2 + t] = R[q * n2 + t];
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
static Grid3 grid_of(const MgLevel& L) { return Grid3{L.nx, L.ny, L.nz}; }
static BoxGeom geom_of(const MgLevel& L) { return BoxGeom{{L.lo[0], L.lo[1], L.lo[2]}, L.H, {L.off[0], L.off[1], L.off[2]}}; }
static Shift3 shift_of(const MgLevel& L) { return Shift3{L.sh[0], L.sh[1], L.sh[2]}; }

// returns false when the box grid is too fine for the mesh (an element spans more than two boxes)
template <int CS> static void level_residual(MgLevel& L, const Grid3& g, hipStream_t s);   // (below, with the cycle)
template <int BS, int PD>
static bool setup_t(Multigrid& mg)
{
    constexpr int CS = CoarseDofs<BS>::value, CC = CS * CS;
    fem_mat A = mg.A;
    fem_mesh mesh = A->mesh;
    fem_ctx ctx = A->ctx;
    hipStream_t s = ctx->stream;
    const int nn = mesh->nnds;
    const size_t ndof = (size_t)nn * BS;
    mg.cs = CS;
    // Partitioned mesh: the boxes are cells of ONE grid over the global bounding box.  Every domain keeps, per level, the
    // sub-grid its own nodes reach and its own part P_r^T K_r P_r of the operator (K_r sub-assembled, the same P on every
    // copy of a shared node: the parts add up to the global Galerkin operator); a box that holds nodes of several domains is
    // shared like an interface node of the fine level, and the box levels are applied like the fine one: local product with
    // the own part + sum over the sharers (one neighbour exchange).  Only the coarsest level (<= 125 boxes) is replicated.
    Halo* halo = mesh->halo.get();
    Comm* comm = (halo && halo->comm->size > 1) ? halo->comm : nullptr;
    // bounding box, aggregates: from the mesh's pattern when an earlier set-up on this mesh left them there (same box size, same partition)
    std::shared_ptr<MgBoxes> bx = A->pat->mg_boxes;
    const bool have_boxes = bx && bx->box == mg.box && bx->halo_seen == (const void*)halo;
    if (!have_boxes) { bx = std::make_shared<MgBoxes>(); bx->box = mg.box; bx->halo_seen = (const void*)halo; }
    mg.bx = bx;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    double nn_global = nn;
    const int nr = comm ? comm->size : 1, me = comm ? comm->rank : 0;
    std::vector<double> tab((size_t)nr * 7, 0.0);   // every domain's bounding box (and its count of owned nodes)
    if (have_boxes) {
        for (int d = 0; d < 3; d++) { lo[d] = bx->lo[d]; hi[d] = bx->hi[d]; }
        nn_global = bx->nn_global; tab = bx->tab;
    } else {
    const int pg = 256;
    DBuf<double> dbb((size_t)pg * 6);
    hipLaunchKernelGGL(k_bbox_partial, dim3(pg), dim3(256), 0, s, (size_t)nn, PD, mesh->coords.p, dbb.p);
    std::vector<double> hbb((size_t)pg * 6);
    dbb.download(hbb.data(), hbb.size(), s);
    FEM_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < pg; b++)
        for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], hbb[b * 6 + d]); hi[d] = std::max(hi[d], hbb[b * 6 + 3 + d]); }
    for (int d = 0; d < 3; d++) { tab[(size_t)me * 7 + d] = lo[d]; tab[(size_t)me * 7 + 3 + d] = hi[d]; }
    if (comm) {
        // min / max / node count over the domains through the one collective there is: every domain
        // fills its own slot of a zeroed table, the sum of the tables is the table of all domains
        DBuf<double> dtab(tab.size());
        dtab.upload(tab.data(), tab.size(), s);
        hipLaunchKernelGGL(k_count_owned, dim3(64), dim3(256), 0, s, nn, halo->owned.p, dtab.p + (size_t)comm->rank * 7 + 6);
        comm->allreduce_sum(dtab.p, (int)tab.size(), s);
        dtab.download(tab.data(), tab.size(), s);
        FEM_HIP(hipStreamSynchronize(s));
        nn_global = 0.0;
        for (int r = 0; r < nr; r++) {
            for (int d = 0; d < PD; d++) { lo[d] = std::min(lo[d], tab[(size_t)r * 7 + d]); hi[d] = std::max(hi[d], tab[(size_t)r * 7 + 3 + d]); }
            nn_global += tab[(size_t)r * 7 + 6];
        }
    }
    for (int d = 0; d < 3; d++) { bx->lo[d] = lo[d]; bx->hi[d] = hi[d]; }
    bx->nn_global = nn_global; bx->tab = tab;
    }
    double vol = 1.0;
    for (int d = 0; d < PD; d++) vol *= std::max(hi[d] - lo[d], 1e-300);
    const double hnode = std::pow(vol / std::max(nn_global, 1.0), 1.0 / PD);  // mean node spacing
    const double H = mg.box * hnode;                                 // ~4 nodes per box edge by default
    Grid3 g{1, 1, 1};
    int* gd[3] = {&g.nx, &g.ny, &g.nz};
    for (int d = 0; d < PD; d++) *gd[d] = std::max(1, (int)std::floor((hi[d] - lo[d]) / H + 1e-9) + 1);
    for (int d = PD; d < 3; d++) lo[d] = -0.5 * H;   // flat direction: the single box layer is centred on z = 0
    FEM_CHECK((long long)g.nx * g.ny * g.nz < (1LL << 26), 9, "pc mg: box grid too large");

    // the levels of the global grid: 2x2x2 boxes merge until the grid is at most 5 boxes per direction
    // (<= 750 dofs: inverted densely; every extra tiny level would cost seven latency-bound launches)
    std::vector<Grid3> G{g};
    while (std::max(G.back().nx, std::max(G.back().ny, G.back().nz)) > 5)
        G.push_back(Grid3{(G.back().nx + 1) / 2, (G.back().ny + 1) / 2, (G.back().nz + 1) / 2});
    const int nlev = (int)G.size();
    // index range of every domain's boxes on level 1 (the same expression as k_assign_boxes, monotone in the coordinate);
    // level l: both ends >> l; the coarsest level is the whole global grid on every domain
    const double invH = 1.0 / H;
    const int gdim[3] = {g.nx, g.ny, g.nz};
    std::vector<int> ext((size_t)nr * 6, 0);
    for (int r = 0; r < nr; r++)
        for (int d = 0; d < PD; d++) {
            ext[(size_t)r * 6 + d] = std::min(std::max((int)std::floor((tab[(size_t)r * 7 + d] - lo[d]) * invH), 0), gdim[d] - 1);
            ext[(size_t)r * 6 + 3 + d] = std::min(std::max((int)std::floor((tab[(size_t)r * 7 + 3 + d] - lo[d]) * invH), 0), gdim[d] - 1);
        }
    auto range_of = [&](int r, int l, int d, int& a, int& b) {   // global index range of domain r on level l
        const int gl[3] = {G[(size_t)l].nx, G[(size_t)l].ny, G[(size_t)l].nz};
        if (l == nlev - 1 || !comm) { a = 0; b = gl[d] - 1; return; }
        a = ext[(size_t)r * 6 + d] >> l; b = ext[(size_t)r * 6 + 3 + d] >> l;
    };

    DBuf<int> err(1);
    err.zero(s);
    mg.lev.clear();
    double Hl = H;
    for (int l = 0; l < nlev; l++, Hl *= 2.0) {
        mg.lev.emplace_back();
        MgLevel& L = mg.lev.back();
        int dims[3];
        for (int d = 0; d < 3; d++) { int a, b; range_of(me, l, d, a, b); L.off[d] = a; dims[d] = b - a + 1; }
        L.nx = dims[0]; L.ny = dims[1]; L.nz = dims[2]; L.nb = L.nx * L.ny * L.nz;
        for (int d = 0; d < 3; d++) L.lo[d] = lo[d];
        if (PD == 2) L.lo[2] = -0.5 * Hl;   // keep the single layer centred on z = 0
        L.H = Hl;
        L.st.alloc((size_t)27 * CC * L.nb);
        L.dinv.alloc((size_t)L.nb * CC); L.x.alloc((size_t)L.nb * CS); L.r.alloc((size_t)L.nb * CS);
        L.t.alloc((size_t)L.nb * CS);
    }
    for (int l = 0; l + 1 < nlev; l++)
        for (int d = 0; d < 3; d++) mg.lev[(size_t)l].sh[d] = mg.lev[(size_t)l].off[d] - 2 * mg.lev[(size_t)l + 1].off[d];

    // level-1 aggregates and their node lists
    MgLevel& L1 = mg.lev[0];
    const Grid3 g1l = grid_of(L1);
    const int nb = L1.nb;
    DBuf<int>& cnt = bx->cnt;
    if (!have_boxes) {
        bx->agg.alloc(nn);
        DBuf<int> cursor((size_t)nb), tmp(exclusive_scan_tmp_ints((size_t)nb + 1));
        cnt.alloc((size_t)nb);
        cnt.zero(s); cursor.zero(s);
        hipLaunchKernelGGL(k_assign_boxes, dim3(g1(nn)), dim3(256), 0, s, nn, PD, mesh->coords.p, lo[0], lo[1], lo[2], invH,
                           g, g1l, Shift3{L1.off[0], L1.off[1], L1.off[2]}, bx->agg.p, cnt.p);
        bx->box_ptr.alloc((size_t)nb + 1);
        exclusive_scan_i32(cnt.p, bx->box_ptr.p, nb, tmp.p, s);
        bx->box_nodes.alloc((size_t)nn);
        hipLaunchKernelGGL(k_fill_boxes, dim3(g1(nn)), dim3(256), 0, s, nn, bx->agg.p, bx->box_ptr.p, cursor.p, bx->box_nodes.p);
        hipLaunchKernelGGL(k_sort_boxes, dim3(g1(nb)), dim3(256), 0, s, nb, bx->box_ptr.p, bx->box_nodes.p);
        bx->drel_v.alloc((size_t)nn * 3); bx->drel_p.alloc((size_t)nn * 3);
        hipLaunchKernelGGL(k_node_offsets<PD>, dim3(g1(nn)), dim3(256), 0, s, nn, g1l, geom_of(L1), mesh->coords.p, bx->agg.p, bx->drel_v.p);
        hipLaunchKernelGGL(k_node_offsets_by_list, dim3(g1(nn)), dim3(256), 0, s, nn, bx->box_nodes.p, bx->drel_v.p, bx->drel_p.p);
        // the groups of four boxes whose rows the level-1 product can read in whole lines (k_galerkin_fine_quad); once per mesh
        static const bool quad_env = !(std::getenv("FEM_AMD_MG_GALERKIN_QUAD") && std::atoi(std::getenv("FEM_AMD_MG_GALERKIN_QUAD")) == 0);
        const int gpr = L1.nx / MG_GROUP, ncand = gpr * L1.ny * L1.nz;
        std::vector<int> gb0, gslot, rest;
        if (quad_env && ncand > 0) {
            bx->gnodes.alloc((size_t)ncand * 256);
            DBuf<unsigned char> ok((size_t)ncand);
            hipLaunchKernelGGL(k_galerkin_groups, dim3(ncand), dim3(256), 0, s, ncand, gpr, L1.nx, bx->box_ptr.p, bx->box_nodes.p, bx->gnodes.p, ok.p);
            std::vector<unsigned char> hok((size_t)ncand);
            ok.download(hok.data(), hok.size(), s);
            FEM_HIP(hipStreamSynchronize(s));
            std::vector<unsigned char> taken((size_t)nb, 0);
            for (int gi = 0; gi < ncand; gi++) {
                if (!hok[(size_t)gi]) continue;
                const int a0 = (gi / gpr) * L1.nx + MG_GROUP * (gi % gpr);
                gb0.push_back(a0); gslot.push_back(gi);
                for (int q = 0; q < MG_GROUP; q++) taken[(size_t)a0 + q] = 1;
            }
            for (int b = 0; b < nb; b++) if (!taken[(size_t)b]) rest.push_back(b);
        } else {
            for (int b = 0; b < nb; b++) rest.push_back(b);
        }
        bx->ngroups = (int)gb0.size(); bx->nrest = (int)rest.size();
        if (bx->ngroups) { bx->group_box0.upload(gb0, s); bx->group_slot.upload(gslot, s); }
        if (bx->nrest) bx->rest_boxes.upload(rest, s);
        FEM_HIP(hipStreamSynchronize(s));   // (host vectors, cursor and tmp go out of scope; once per mesh)
        A->pat->mg_boxes = bx;
    }

    // free-dof weights and fine Jacobi diagonal
    mg.w.alloc(ndof); mg.dinv0.alloc(ndof); mg.t0.alloc(ndof);
    mat_get_diagonal_raw(A, mg.t0.p);
    halo_sum(mesh, mg.t0.p, BS, BS, 1);   // assembled diagonal: the penalty sits on the owner's copy only
    hipLaunchKernelGGL(k_free_weights, dim3(gs(ndof)), dim3(256), 0, s, ndof, mg.t0.p, mg.w.p, mg.dinv0.p);
    if (comm) {
        mg.wr.alloc(ndof);
        hipLaunchKernelGGL(k_owner_weights, dim3(g1(nn)), dim3(256), 0, s, nn, BS, mg.w.p, halo->owned.p, mg.wr.p);
    }

    // this domain's part of the level-1 operator
    if (bx->ngroups)
        hipLaunchKernelGGL((k_galerkin_fine_quad<BS, CS, PD>), dim3(bx->ngroups), dim3(256), 0, s, g1l, geom_of(L1), mesh->coords.p, bx->ngroups,
                           bx->group_box0.p, bx->group_slot.p, bx->gnodes.p, bx->agg.p, A->rowlen.p, A->boff.p, A->bcol.p, A->values(),
                           mg.w.p, L1.st.p, err.p);
    if (bx->nrest)
        hipLaunchKernelGGL((k_galerkin_fine_wave<BS, CS, PD>), dim3(g1((size_t)64 * bx->nrest)), dim3(256), 0, s, g1l, geom_of(L1),
                           mesh->coords.p, bx->box_ptr.p, bx->box_nodes.p, bx->agg.p, A->rowlen.p, A->boff.p, A->bcol.p, A->values(),
                           mg.w.p, L1.st.p, err.p, bx->rest_boxes.p, bx->nrest);
    hipLaunchKernelGGL((k_galerkin_mirror<CS>), dim3(g1((size_t)13 * CC * nb)), dim3(256), 0, s, g1l, L1.st.p);
    if (comm) {
        // the flag travels so that every domain raises the same error
        DBuf<double> derr(1);
        int herr0 = 0;
        err.download(&herr0, 1, s);
        FEM_HIP(hipStreamSynchronize(s));
        const double e0 = herr0 ? 1.0 : 0.0;
        derr.upload(&e0, 1, s);
        comm->allreduce_sum(derr.p, 1, s);
        double e1 = 0.0;
        derr.download(&e1, 1, s);
        FEM_HIP(hipStreamSynchronize(s));
        const int any = e1 > 0.0 ? 1 : 0;
        err.upload(&any, 1, s);
        FEM_HIP(hipStreamSynchronize(s));
    }
    {
        int herr = 0;
        err.download(&herr, 1, s);
        FEM_HIP(hipStreamSynchronize(s));
        if (herr) return false;
    }
    // coarser levels: T^T A_l T of the domain's own part, level by level (Galerkin products are linear in the operator)
    for (int l = 1; l < nlev; l++) {
        MgLevel& Lf = mg.lev[(size_t)l - 1];
        MgLevel& Lc = mg.lev[(size_t)l];
        hipLaunchKernelGGL(k_galerkin_box<CS>, dim3(g1((size_t)27 * Lc.nb)), dim3(256), 0, s, grid_of(Lf), geom_of(Lf), grid_of(Lc),
                           geom_of(Lc), shift_of(Lf), Lf.st.p, Lc.st.p);
    }
    if (comm) {
        // the replicated coarsest operator: the sum of the parts (at most 27 * 36 * 125 numbers, whatever the mesh)
        MgLevel& Lc = mg.lev.back();
        comm->allreduce_sum(Lc.st.p, (int)((size_t)27 * CC * Lc.nb), s);
    }
    // shared boxes, level by level: a box is shared with every other domain that also holds a node in it (level 1) / a child of
    // it (coarser levels).  The domains whose index ranges overlap exchange their occupancy of the overlap (one neighbour
    // exchange per level, a few thousand numbers), the boxes occupied on both sides are the shared ones -- in the same
    // (global lexicographic) order on both sides.
    if (comm) {
        std::vector<unsigned char> occ;
        {
            std::vector<int> hcnt((size_t)nb);
            cnt.download(hcnt.data(), hcnt.size(), s);
            FEM_HIP(hipStreamSynchronize(s));
            occ.resize((size_t)nb);
            for (int b = 0; b < nb; b++) occ[(size_t)b] = hcnt[(size_t)b] > 0;
        }
        for (int l = 0; l + 1 < nlev; l++) {
            MgLevel& L = mg.lev[(size_t)l];
            int a0[3], a1[3];
            for (int d = 0; d < 3; d++) range_of(me, l, d, a0[d], a1[d]);
            struct Cand { int q; int r0[3], r1[3]; size_t n; };
            std::vector<Cand> cand;
            std::vector<size_t> off{0};
            for (int q = 0; q < nr; q++) {
                if (q == me) continue;
                Cand c; c.q = q; c.n = 1;
                bool ov = true;
                for (int d = 0; d < 3; d++) {
                    int b0, b1;
                    range_of(q, l, d, b0, b1);
                    c.r0[d] = std::max(a0[d], b0); c.r1[d] = std::min(a1[d], b1);
                    if (c.r0[d] > c.r1[d]) ov = false;
                    else c.n *= (size_t)(c.r1[d] - c.r0[d] + 1);
                }
                if (!ov) continue;
                cand.push_back(c);
                off.push_back(off.back() + c.n);
            }
            std::vector<double> sb(std::max<size_t>(off.back(), 1), 0.0), rb(std::max<size_t>(off.back(), 1), 0.0);
            for (size_t k = 0; k < cand.size(); k++) {
                const Cand& c = cand[k];
                size_t p = off[k];
                for (int iz = c.r0[2]; iz <= c.r1[2]; iz++)
                    for (int iy = c.r0[1]; iy <= c.r1[1]; iy++)
                        for (int ix = c.r0[0]; ix <= c.r1[0]; ix++)
                            sb[p++] = occ[(size_t)(ix - a0[0]) + (size_t)L.nx * ((size_t)(iy - a0[1]) + (size_t)L.ny * (size_t)(iz - a0[2]))] ? 1.0 : 0.0;
            }
            DBuf<double> dsb(sb.size()), drb(rb.size());
            dsb.upload(sb.data(), sb.size(), s);
            std::vector<int> cr;
            for (auto& c : cand) cr.push_back(c.q);
            comm->exchange((int)cand.size(), cr.data(), off.data(), dsb.p, drb.p, s);
            drb.download(rb.data(), rb.size(), s);
            FEM_HIP(hipStreamSynchronize(s));
            std::vector<int> nrk, nptr{0}, nbox;
            for (size_t k = 0; k < cand.size(); k++) {
                const Cand& c = cand[k];
                size_t p = off[k];
                const size_t before = nbox.size();
                for (int iz = c.r0[2]; iz <= c.r1[2]; iz++)
                    for (int iy = c.r0[1]; iy <= c.r1[1]; iy++)
                        for (int ix = c.r0[0]; ix <= c.r1[0]; ix++, p++)
                            if (sb[p] != 0.0 && rb[p] != 0.0)
                                nbox.push_back((ix - a0[0]) + L.nx * ((iy - a0[1]) + L.ny * (iz - a0[2])));
                if (nbox.size() > before) { nrk.push_back(c.q); nptr.push_back((int)nbox.size()); }
            }
            L.halo = halo_build(comm, L.nb, (int)nrk.size(), nrk.data(), nptr.data(), nbox.data(), s, "pc mg (shared boxes)");
            // occupancy of the next level: a box is occupied where one of its children is
            if (l + 2 < nlev) {
                MgLevel& C = mg.lev[(size_t)l + 1];
                std::vector<unsigned char> oc((size_t)C.nb, 0);
                for (int b = 0; b < L.nb; b++) {
                    if (!occ[(size_t)b]) continue;
                    const int px = (b % L.nx + L.sh[0]) >> 1, py = ((b / L.nx) % L.ny + L.sh[1]) >> 1, pz = (b / (L.nx * L.ny) + L.sh[2]) >> 1;
                    oc[(size_t)px + (size_t)C.nx * ((size_t)py + (size_t)C.ny * (size_t)pz)] = 1;
                }
                occ.swap(oc);
            }
        }
    }
    // block-Jacobi smoother: inverse of the diagonal blocks -- of the whole operator, i.e. summed over the sharers of a box
    for (int l = 0; l < nlev; l++) {
        MgLevel& L = mg.lev[(size_t)l];
        const double* diag = L.st.p + (size_t)13 * CC * L.nb;
        if (L.halo) {
            L.diag.alloc((size_t)CC * L.nb);
            FEM_HIP(hipMemcpyAsync(L.diag.p, diag, sizeof(double) * CC * L.nb, hipMemcpyDeviceToDevice, s));
            for (int e0 = 0; e0 < CC; e0 += 6)
                halo_sum_h(L.halo.get(), L.diag.p + (size_t)e0 * L.nb, std::min(6, CC - e0), 1, (size_t)L.nb, s, true);
            diag = L.diag.p;
        }
        hipLaunchKernelGGL(k_level_dinv<CS>, dim3(g1(L.nb)), dim3(256), 0, s, L.nb, diag, L.dinv.p);
    }
    // The cycle reads the stencil of a large level twice per application and nothing else of that size: it gets a single-precision
    // copy (the preconditioner is a fixed linear operator either way; the diagonal blocks and the next level's Galerkin product have
    // used the doubles, the eigenvalue estimate below runs on what the cycle will run on).  -pc_mg_stencil_f32 0 / FEM_AMD_MG_ST_F32=0: doubles.
    for (size_t l = 0; l + 1 < mg.lev.size(); l++) {
        MgLevel& L = mg.lev[l];
        if (mg.stencil_f32 && L.nb > 16384 && CS % 2 == 0) {
            if (L.stf.n != L.st.n) L.stf.alloc(L.st.n);
            if constexpr (CS % 2 == 0)
                hipLaunchKernelGGL(k_stencil_to_float2<CS>, dim3((unsigned)std::min<size_t>(g1(L.st.n / 2), 8192)), dim3(256), 0, s, L.nb, L.st.p,
                                   reinterpret_cast<float2*>(L.stf.p));
        } else L.stf.release();
    }
    // damping of the block-Jacobi smoother on every smoothed level: power iteration on Dinv A
    mg.omega_l.alloc(mg.lev.size() + 2);
    for (size_t l = 0; l + 1 < mg.lev.size(); l++) {
        MgLevel& L = mg.lev[l];
        const Grid3 gl = grid_of(L);
        const size_t n = (size_t)L.nb * CS;
        const unsigned char* own = L.halo ? L.halo->owned.p : nullptr;
        double* n2 = mg.omega_l.p + mg.lev.size();
        L.r.zero(s);
        hipLaunchKernelGGL(k_hash_fill_box<CS>, dim3(g1(L.nb)), dim3(256), 0, s, gl, Shift3{L.off[0], L.off[1], L.off[2]}, G[l], L.x.p);
        const int npow = 12;
        for (int it = 0; it < npow; it++) {
            if (it == npow - 1) hipLaunchKernelGGL(k_norm2_small, dim3(1), dim3(1024), 0, s, n, L.x.p, n2, own, CS);
            level_residual<CS>(L, gl, s);
            halo_sum_h(L.halo.get(), L.t.p, CS, CS, 1, s, true);
            hipLaunchKernelGGL((k_block_smooth<CS, false>), dim3(g1(L.nb)), dim3(256), 0, s, L.nb, 0, L.nb, (const double*)nullptr,
                               L.dinv.p, L.t.p, L.x.p);
        }
        hipLaunchKernelGGL(k_norm2_small, dim3(1), dim3(1024), 0, s, n, L.x.p, n2 + 1, own, CS);
        if (comm) comm->allreduce_sum(n2, 2, s);
        hipLaunchKernelGGL(k_set_omega, dim3(1), dim3(1), 0, s, n2, mg.omega_l.p + l);
    }
    // fine level of the V-cycle form: point Jacobi must stay below 2 / lambda_max(D^-1 K) or the cycle
    // is indefinite (materials with nu -> 0.5, distorted elements and stiffness jumps push lambda_max
    // beyond 3); same power iteration, the norms are all-reduced so that every domain takes the
    // same damping.  The additive form needs no damping (a sum of positive definite terms).
    mg.omega_fine = mg.omega;
    if (!mg.fine_additive) {
        DBuf<double> xv(ndof), n2(2);
        hipLaunchKernelGGL(k_hash_fill, dim3(gs(ndof)), dim3(256), 0, s, ndof, xv.p);
        const int npow = 10;
        for (int it = 0; it < npow; it++) {
            if (it == npow - 1) hipLaunchKernelGGL(k_norm2_small, dim3(1), dim3(1024), 0, s, ndof, xv.p, n2.p, (const unsigned char*)nullptr, 1);
            mat_mult_raw(A, xv.p, mg.t0.p, nullptr, nullptr);
            halo_sum(mesh, mg.t0.p, BS, BS, 1);
            hipLaunchKernelGGL(k_smooth_first, dim3(gs(ndof)), dim3(256), 0, s, ndof, 1.0, mg.dinv0.p, mg.t0.p, xv.p);
        }
        hipLaunchKernelGGL(k_norm2_small, dim3(1), dim3(1024), 0, s, ndof, xv.p, n2.p + 1, (const unsigned char*)nullptr, 1);
        if (comm) comm->allreduce_sum(n2.p, 2, s);
        double h2[2] = {0.0, 0.0};
        n2.download(h2, 2, s);
        FEM_HIP(hipStreamSynchronize(s));
        const double lam = (h2[0] > 0.0 && h2[1] > 0.0) ? std::sqrt(h2[1] / h2[0]) : 2.5;
        mg.lambda_fine = lam;
        mg.omega_fine = std::min(mg.omega, 1.6 / std::max(lam, 1.0));   // 20 % below the stability bound
    }
    // coarsest: dense inverse
    const Grid3 gcoarse = grid_of(mg.lev.back());
    mg.ncoarse = gcoarse.nb() * CS;
    mg.cinv.alloc((size_t)mg.ncoarse * 2 * mg.ncoarse);
    hipLaunchKernelGGL(k_dense_from_stencil<CS>, dim3(std::max(gcoarse.nb(), 1)), dim3(256), 0, s, gcoarse, mg.lev.back().st.p, mg.cinv.p);
    // one resident launch where the block rows fit the LDS and the chip (FEM_AMD_MG_GJ_RESIDENT=0: the two launches per block)
    static const bool gj_env = !(std::getenv("FEM_AMD_MG_GJ_RESIDENT") && std::atoi(std::getenv("FEM_AMD_MG_GJ_RESIDENT")) == 0);
    bool gj_done = false;
    {
        const int n = mg.ncoarse;
        const size_t shm = (size_t)CS * 2 * n * sizeof(double);
        if (gj_env && shm <= 120 * 1024 && n / CS <= ctx->num_cu && n / CS >= 2) {
            FEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gj_resident<CS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            DBuf<double> rowbuf((size_t)2 * CS * 2 * n);
            DBuf<unsigned> words(GridBarrier::WORDS + 32);
            words.zero(s);
            GridBarrier bar{words.p, reinterpret_cast<int*>(words.p + GridBarrier::WORDS)};
            hipLaunchKernelGGL(k_gj_resident<CS>, dim3(n / CS), dim3(256), shm, s, n, mg.cinv.p, rowbuf.p, bar);
            int herr = 0;
            FEM_HIP(hipMemcpyAsync(&herr, bar.err, sizeof(int), hipMemcpyDeviceToHost, s));
            FEM_HIP(hipStreamSynchronize(s));
            gj_done = herr == 0;
            if (!gj_done)   // (the barrier gave up: the matrix is half eliminated -- build it again for the launches below)
                hipLaunchKernelGGL(k_dense_from_stencil<CS>, dim3(std::max(gcoarse.nb(), 1)), dim3(256), 0, s, gcoarse, mg.lev.back().st.p, mg.cinv.p);
        }
    }
    if (!gj_done) {
        const int n = mg.ncoarse;
        DBuf<double> rowk((size_t)CS * 2 * n), colk((size_t)CS * n);
        for (int kb = 0; kb < n / CS; kb++) {
            hipLaunchKernelGGL(k_gj_pivot<CS>, dim3(g1((size_t)2 * n)), dim3(256), 0, s, n, kb, mg.cinv.p, rowk.p, colk.p);
            hipLaunchKernelGGL(k_gj_update<CS>, dim3(g1((size_t)n * 2 * n)), dim3(256), 0, s, n, kb, mg.cinv.p, rowk.p,
                               colk.p);
        }
        FEM_HIP(hipStreamSynchronize(s));   // rowk/colk go out of scope
    }
    FEM_HIP(hipGetLastError());
    FEM_HIP(hipStreamSynchronize(s));
    if (getenv("FEM_AMD_MG_VERBOSE")) {
        std::vector<double> om(mg.lev.size() + 2);
        mg.omega_l.download(om.data(), om.size(), s);
        FEM_HIP(hipStreamSynchronize(s));
        if (!mg.fine_additive)
            fprintf(stderr, "[fem_amd mg] fine level: lambda_max(D^-1 K) ~ %g, omega = %g\n", mg.lambda_fine, mg.omega_fine);
        for (size_t l = 0; l < mg.lev.size(); l++) {
            const MgLevel& L = mg.lev[l];
            const int shared = L.halo ? L.halo->ninodes : 0;
            fprintf(stderr, "[fem_amd mg] level %zu: %d x %d x %d boxes of the global %d x %d x %d (origin %d %d %d, %d shared with %d domains), "
                            "%d dofs/box, H = %g%s\n", l + 1, L.nx, L.ny, L.nz, G[l].nx, G[l].ny, G[l].nz, L.off[0], L.off[1], L.off[2], shared,
                    L.halo ? L.halo->nneigh : 0, CS, L.H,
                    (l + 1 < mg.lev.size()) ? (", omega = " + std::to_string(om[l])).c_str() : ", dense inverse (replicated)");
        }
    }
    return true;
}

// t = r - A x on a level, by the kernel its size asks for: small levels 32 lanes per box (latency), middle ones a thread per block row
// and box, large ones one thread per box from the single-precision copy (or from the doubles where there is none)
template <int CS>
static void level_residual(MgLevel& L, const Grid3& g, hipStream_t s)
{
    static const int rows_min = std::getenv("FEM_AMD_MG_ROWS_MIN") ? std::atoi(std::getenv("FEM_AMD_MG_ROWS_MIN")) : 2048;   // (a level with more boxes than this: the row kernels)
    const unsigned char* own = L.halo ? L.halo->owned.p : nullptr;
    const dim3 gsl(g1((size_t)std::max(L.nb, 1) * 32)), gbx(g1(L.nb));
    const int nbp = (L.nb + 63) & ~63;
    const dim3 grw(g1((size_t)nbp * CS));
    if (L.nb > 16384 && CS % 2 == 0 && L.stf.n == L.st.n) {
        if constexpr (CS % 2 == 0) {
            const float2* f2 = reinterpret_cast<const float2*>(L.stf.p);
            if (own) hipLaunchKernelGGL((k_box_residual_f2<CS, true>), gbx, dim3(256), 0, s, g, 0, L.nb, f2, L.x.p, L.r.p, L.t.p, own);
            else hipLaunchKernelGGL((k_box_residual_f2<CS, false>), gbx, dim3(256), 0, s, g, 0, L.nb, f2, L.x.p, L.r.p, L.t.p, own);
        }
    } else if (L.nb > 16384) {
        if (own) hipLaunchKernelGGL((k_box_residual<CS, true>), gbx, dim3(256), 0, s, g, 0, L.nb, L.st.p, L.x.p, L.r.p, L.t.p, own);
        else hipLaunchKernelGGL((k_box_residual<CS, false>), gbx, dim3(256), 0, s, g, 0, L.nb, L.st.p, L.x.p, L.r.p, L.t.p, own);
    } else if (L.nb > rows_min) {
        if (own) hipLaunchKernelGGL((k_box_residual_rows<CS, true>), grw, dim3(256), 0, s, g, nbp, L.st.p, L.x.p, L.r.p, L.t.p, own);
        else hipLaunchKernelGGL((k_box_residual_rows<CS, false>), grw, dim3(256), 0, s, g, nbp, L.st.p, L.x.p, L.r.p, L.t.p, own);
    } else {
        if (own) hipLaunchKernelGGL((k_box_residual_slots<CS, true>), gsl, dim3(256), 0, s, g, 0, L.nb, L.st.p, L.x.p, L.r.p, L.t.p, own);
        else hipLaunchKernelGGL((k_box_residual_slots<CS, false>), gsl, dim3(256), 0, s, g, 0, L.nb, L.st.p, L.x.p, L.r.p, L.t.p, own);
    }
}

template <int CS>
static void cycle_t(Multigrid& mg, size_t l)
{
    hipStream_t s = mg.A->ctx->stream;
    MgLevel& L = mg.lev[l];
    const Grid3 g = grid_of(L);
    const size_t n = (size_t)L.nb * CS;
    if (l + 1 == mg.lev.size()) {
        hipLaunchKernelGGL(k_dense_apply, dim3(g1(n * 64)), dim3(256), 0, s, (int)n, mg.cinv.p, L.r.p, L.x.p);
        return;
    }
    // Partitioned mesh (L.halo): r and x are held for this domain's boxes, the same on every domain that shares a box; st is the
    // domain's part of the operator.  t = [own] r - A_r x is a partial residual whose sum over the sharers is r - A x: before
    // the restriction it need not be summed (the restriction is linear: the coarse right-hand side is summed instead, one
    // exchange on a level with an eighth of the boxes), before the second smoothing step it is (one neighbour exchange).
    Halo* halo = mg.A->mesh->halo.get();
    Comm* comm = (halo && halo->comm->size > 1) ? halo->comm : nullptr;
    auto residual = [&] { level_residual<CS>(L, g, s); };
    MgLevel& C = mg.lev[l + 1];
    const Grid3 gc = grid_of(C);
    hipLaunchKernelGGL((k_block_smooth<CS, false>), dim3(g1(L.nb)), dim3(256), 0, s, L.nb, 0, L.nb, mg.omega_l.p + l, L.dinv.p,
                       L.r.p, L.x.p);
    residual();
    hipLaunchKernelGGL(k_restrict_box<CS>, dim3(g1(C.nb)), dim3(256), 0, s, g, geom_of(L), gc, geom_of(C), shift_of(L), L.t.p, C.r.p);
    if (comm) {
        if (l + 2 == mg.lev.size()) comm->allreduce_sum(C.r.p, C.nb * CS, s);   // the replicated coarsest level: <= 750 numbers
        else halo_sum_h(C.halo.get(), C.r.p, CS, CS, 1, s, true);
    }
    cycle_t<CS>(mg, l + 1);
    hipLaunchKernelGGL(k_prolong_box<CS>, dim3(g1(L.nb)), dim3(256), 0, s, g, geom_of(L), gc, geom_of(C), shift_of(L), mg.alpha, C.x.p,
                       L.x.p);
    residual();
    if (comm) halo_sum_h(L.halo.get(), L.t.p, CS, CS, 1, s, true);
    hipLaunchKernelGGL((k_block_smooth<CS, true>), dim3(g1(L.nb)), dim3(256), 0, s, L.nb, 0, L.nb, mg.omega_l.p + l,
                       L.dinv.p, L.t.p, L.x.p);
}

template <int BS, int PD>
static void apply_t(Multigrid& mg, const double* r, double* z, const int* done, double* pzz, double* pzr, int npart)
{
    constexpr int CS = CoarseDofs<BS>::value;
    fem_mat A = mg.A;
    hipStream_t s = A->ctx->stream;
    const int nn = mg.nn;
    const size_t ndof = (size_t)nn * BS;
    MgLevel& L1 = mg.lev[0];
    const Grid3 g = grid_of(L1);
    const BoxGeom geo = geom_of(L1);
    const double* coords = A->mesh->coords.p;
    Halo* halo = A->mesh->halo.get();
    Comm* comm = (halo && halo->comm->size > 1) ? halo->comm : nullptr;
    // partitioned mesh: r and z are consistent on every copy of a shared node; the restriction counts
    // each node once (owner weights), the sum over the sharers of a box (one neighbour exchange) makes the
    // level-1 residual whole on every domain that holds the box, and the prolongation is local again
    const unsigned char* wr = comm ? mg.wr.p : mg.w.p;
    const bool fd = mg.float_offsets && mg.bx->drel_v.n == (size_t)nn * 3;
    auto restrict_to_level1 = [&](const double* kz) {
        if (fd) hipLaunchKernelGGL((k_restrict_fine<BS, CS, PD, true>), dim3(g1((size_t)L1.nb * 64)), dim3(256), 0, s, g, geo, coords,
                                   mg.bx->box_ptr.p, mg.bx->box_nodes.p, wr, r, kz, L1.r.p, mg.bx->drel_p.p);
        else hipLaunchKernelGGL((k_restrict_fine<BS, CS, PD>), dim3(g1((size_t)L1.nb * 64)), dim3(256), 0, s, g, geo, coords,
                                mg.bx->box_ptr.p, mg.bx->box_nodes.p, wr, r, kz, L1.r.p, (const float*)nullptr);
        if (comm) {
            if (mg.lev.size() == 1) comm->allreduce_sum(L1.r.p, L1.nb * CS, s);   // (level 1 is the replicated coarsest one)
            else halo_sum_h(L1.halo.get(), L1.r.p, CS, CS, 1, s, true);
        }
    };
    auto prolong_from_level1 = [&] {
        if (fd) hipLaunchKernelGGL((k_prolong_fine<BS, CS, PD, true>), dim3(g1(nn)), dim3(256), 0, s, nn, g, geo, coords, mg.alpha, mg.bx->agg.p,
                                   mg.w.p, L1.x.p, z, mg.bx->drel_v.p);
        else hipLaunchKernelGGL((k_prolong_fine<BS, CS, PD>), dim3(g1(nn)), dim3(256), 0, s, nn, g, geo, coords, mg.alpha, mg.bx->agg.p,
                                mg.w.p, L1.x.p, z, (const float*)nullptr);
    };
    if (mg.fine_additive) {
        // fine level additive (Jacobi + coarse-grid correction of the un-smoothed residual): the
        // preconditioner costs no fine-level product; the coarse hierarchy is still a V-cycle
        restrict_to_level1(nullptr);
        cycle_t<CS>(mg, 0);
        if (pzz) {   // the caller wants z.z and z.r as well: one fused pass (single domain)
            if (fd) hipLaunchKernelGGL((k_additive_finish<BS, CS, PD, true>), dim3(npart), dim3(256), 0, s, nn, g, geo, coords, mg.alpha,
                                       mg.bx->agg.p, mg.w.p, mg.dinv0.p, r, L1.x.p, z, pzz, pzr, done, mg.bx->drel_v.p);
            else hipLaunchKernelGGL((k_additive_finish<BS, CS, PD>), dim3(npart), dim3(256), 0, s, nn, g, geo, coords, mg.alpha,
                                    mg.bx->agg.p, mg.w.p, mg.dinv0.p, r, L1.x.p, z, pzz, pzr, done, (const float*)nullptr);
        } else {
            hipLaunchKernelGGL(k_smooth_first, dim3(gs(ndof)), dim3(256), 0, s, ndof, 1.0, mg.dinv0.p, r, z);
            prolong_from_level1();
        }
        return;
    }
    hipLaunchKernelGGL(k_smooth_first, dim3(gs(ndof)), dim3(256), 0, s, ndof, mg.omega_fine, mg.dinv0.p, r, z);
    mat_mult_raw(A, z, mg.t0.p, nullptr, done);
    halo_sum(A->mesh, mg.t0.p, BS, BS, 1);
    restrict_to_level1(mg.t0.p);
    cycle_t<CS>(mg, 0);
    prolong_from_level1();
    mat_mult_raw(A, z, mg.t0.p, nullptr, done);
    halo_sum(A->mesh, mg.t0.p, BS, BS, 1);
    hipLaunchKernelGGL(k_fine_post, dim3(gs(ndof)), dim3(256), 0, s, ndof, mg.omega_fine, mg.dinv0.p, r, mg.t0.p, z);
}

void Multigrid::setup(fem_mat A_)
{
    A = A_; bs = A->bs; nn = A->nbrows;
    A->ctx->bind();
    // graded meshes: where the largest element spans more than two boxes the grid is coarsened
    // (box edge x1.5) until every element fits; all domains of a partitioned mesh take the same decision
    for (int attempt = 0;; attempt++) {
        const bool ok = (bs == 3) ? setup_t<3, 3>(*this) : setup_t<2, 2>(*this);
        if (ok) break;
        FEM_CHECK(attempt < 6 && box < 64.0, 9,
                  "pc mg: an element spans more than two boxes of the aggregation grid even with boxes of " +
                      std::to_string(box) + " mean node spacings (mesh too graded)");
        box *= 1.5;
    }
}

void Multigrid::apply(const double* r, double* z, const int* done, double* pzz, double* pzr, int npart)
{
    if (bs == 3) apply_t<3, 3>(*this, r, z, done, pzz, pzr, npart); else apply_t<2, 2>(*this, r, z, done, pzz, pzr, npart);
}

bool Multigrid::fuses_dots() const { return fine_additive; }

size_t Multigrid::device_bytes() const
{
    size_t b = omega_l.bytes() + wr.bytes() + (bx ? bx->agg.bytes() + bx->box_ptr.bytes() + bx->box_nodes.bytes() : 0) + w.bytes() + dinv0.bytes() + t0.bytes() + cinv.bytes();
    for (auto& L : lev) b += L.st.bytes() + L.stf.bytes() + L.dinv.bytes() + L.x.bytes() + L.r.bytes() + L.t.bytes() + L.diag.bytes();
    return b;
}

}  // namespace fem
