// fem_amd/csrc/ebe_plan.hip -- the tile plan of the one-pass element-by-element product (ebe_tile.hip), built on the device.
//
// The plan is set-up work of the path behind `-mat_mult_type ebe`: the reference assembles once per run
// (src/main.F90:151-171), so whatever a run needs before its first MatMult is on its critical path.  The host builder of
// round 2 (ebe_tile.hip: ebe_tile_build_host) copied connectivity, coordinates and node lists to the host and took 150-340 ms
// at config 3 -- longer than the whole solve.  Here the same plan comes out of integer kernels over data that never
// leaves HBM (a few milliseconds at config 3), the way the symbolic phase of the assembly (mat.hip) is built:
//
//   1. reductions: bounding box of the nodes; per element the mean edge vectors (the lattice a structured mesh sits
//      on), the mean spacings and the class (distorted / parallelepiped / axis-aligned brick); fixed-order sums
//   2. node -> box of NB nodes per direction (in the coordinates of the mesh's own lattice where it has one), counting sort
//   3. one workgroup per box ("set" of owned nodes): elements that touch an owned node (LDS hash over the node -> element
//      lists of the symbolic phase), ordered along the grid (bitonic sort in LDS), the tile's nodes (owned first, each part
//      ascending), tile-local connectivity, chains of x-neighbours (<= 16, block scans), greedy colouring of the chains in
//      that order (one wavefront, the lanes cover a chain's node references), 16-lane rows, steps of <= 256 lanes
//      -- a first pass counts (and halves the sets that exceed the LDS image of the product kernels: k_plan_split, until
//      every set fits), a scan turns the counts into offsets, a second pass writes
//   4. stored geometry of the parallelepipeds (k_tile_geom, as before)
//
// The arithmetic that decides anything (box of a node, grid key of an element, cut of a halved set) is written so that
// host and device take the same decisions (no contracted multiply-adds); on every mesh of tools/soak_tile_products.py and of
// tests/test_gpu_ebe.py the two builders produce the same plan word for word (tests/test_gpu_tile_plan.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>

#include "ebe_tile.h"
#include "scan.h"

using namespace fem;

namespace {

constexpr int PL_T = 256;                  // threads of the per-set kernels
constexpr int PL_MAXN = 3072;              // nodes of a tile the LDS tables hold (>= TILE_MAX_NODES)
constexpr int PL_SORT = 4096;              // scratch of the node sorts / element hash (power of two >= PL_MAXN + PL_T)
constexpr int PL_MAXE = PLAN_MAX_ELEMS;    // elements of a tile (power of two)
constexpr int PL_NH = 8192;                // hash of the nodes that are not owned
constexpr int PL_SEG = PL_MAXE / PL_T;     // consecutive elements per thread in the chain scans
static_assert(PL_MAXN >= TILE_MAX_NODES && PL_SORT >= PL_MAXN + PL_T && PL_SEG * PL_T == PL_MAXE, "plan limits");
constexpr int NRED = 16;                   // reduced doubles of the element pass

struct PlanGeom {
    double lo[3], h[3], eps[3];            // grid origin, spacing, a millionth of the spacing (cut of a halved set)
    int NB[3], nbox[3];
    int max_nodes, max_own;
};
// a set of owned nodes = a range of `border` (nodes sorted by box)
struct SetRec {
    int start, cnt;
    int status;                            // 0: to be examined, 1: a tile, 2: to be halved
    int nnode, nslots, nsteps, nel, info;  // of a tile; info = class (0 distorted, 1 parallelepipeds, 2 bricks) | shares a node with another domain << 2
};
struct PlanCounters {
    int n_split, n_fail, fail_code, fail_a, fail_b;
    int max_nnode, max_nsteps, max_nown;
    unsigned long long visits, slots, nodes;
};
enum { FAIL_TOO_MANY_NODES = 1, FAIL_COINCIDENT = 2, FAIL_COLOURS = 3 };

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// ---- 1. reductions ----------------------------------------------------------------------------------------------------
// per workgroup: min / max of every coordinate column -> part[block][6]
__global__ void __launch_bounds__(256) k_plan_bbox(int nn, int pd, const double* __restrict__ x, double* __restrict__ part)
{
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < (size_t)nn; v += (size_t)gridDim.x * 256)
        for (int d = 0; d < pd; d++) { const double c = x[v * pd + d]; mn[d] = fmin(mn[d], c); mx[d] = fmax(mx[d], c); }
    __shared__ double sh[4][6];
    for (int d = 0; d < 3; d++) {
        for (int o = 32; o > 0; o >>= 1) { mn[d] = fmin(mn[d], __shfl_down(mn[d], o)); mx[d] = fmax(mx[d], __shfl_down(mx[d], o)); }
        if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][d] = mn[d]; sh[threadIdx.x >> 6][3 + d] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int j = threadIdx.x;
        double r = sh[0][j];
        for (int w = 1; w < 4; w++) r = j < 3 ? fmin(r, sh[w][j]) : fmax(r, sh[w][j]);
        part[(size_t)blockIdx.x * 6 + j] = r;
    }
}
__global__ void __launch_bounds__(64) k_plan_bbox_final(int nblocks, const double* __restrict__ part, double* __restrict__ out)
{
    const int j = threadIdx.x;
    if (j >= 6) return;
    double r = part[j];
    for (int b = 1; b < nblocks; b++) r = j < 3 ? fmin(r, part[(size_t)b * 6 + j]) : fmax(r, part[(size_t)b * 6 + j]);
    out[j] = r;
}

// the parallelepiped test of ebe_tile.hip's host_hex_affine (same operations in the same order)
__device__ bool plan_hex_affine(const double (&X)[8][3])
{
    for (int j = 0; j < 3; j++) {
        double v[8];
        for (int a = 0; a < 8; a++) v[a] = X[a][j];
        const double s00 = v[1] + v[0], d00 = v[1] - v[0], s10 = v[2] + v[3], d10 = v[2] - v[3];
        const double s01 = v[5] + v[4], d01 = v[5] - v[4], s11 = v[6] + v[7], d11 = v[6] - v[7];
        const double ss0 = s10 + s00, sd0 = s10 - s00, ds0 = d10 + d00, dd0 = d10 - d00;
        const double ss1 = s11 + s01, sd1 = s11 - s01, ds1 = d11 + d01, dd1 = d11 - d01;
        const double c0 = 0.125 * (ss1 + ss0), cx = 0.125 * (ds1 + ds0), cy = 0.125 * (sd1 + sd0), cz = 0.125 * (ss1 - ss0);
        const double cxy = 0.125 * (dd1 + dd0), cxz = 0.125 * (ds1 - ds0), cyz = 0.125 * (sd1 - sd0), cxyz = 0.125 * (dd1 - dd0);
        const double tol = 0x1p-50 * (fabs(c0) + fabs(cx) + fabs(cy) + fabs(cz));
        if (!(fabs(cxy) <= tol && fabs(cxz) <= tol && fabs(cyz) <= tol && fabs(cxyz) <= tol)) return false;
    }
    return true;
}

// per element: class; per workgroup the sums of [0..2] largest mean-edge extent per direction, [3..11] signed mean edge
// vectors along xi, eta, zeta, [12..14] their lengths, [15] number of parallelepipeds -> part[block][NRED]
template <bool HEX>
__global__ void __launch_bounds__(256) k_plan_elems(int nel, const int* __restrict__ conn, const double* __restrict__ coords,
                                                    unsigned char* __restrict__ aff, double* __restrict__ part)
{
    constexpr int NPE = HEX ? 8 : 4, PD = HEX ? 3 : 2;
    double acc[NRED];
    for (int j = 0; j < NRED; j++) acc[j] = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)nel; e += (size_t)gridDim.x * 256) {
        double X[8][3];
        for (int a = 0; a < 8; a++)
            for (int d = 0; d < 3; d++) X[a][d] = 0.0;
        for (int a = 0; a < NPE; a++) {
            const size_t v = (size_t)conn[e * NPE + a];
            for (int d = 0; d < PD; d++) X[a][d] = coords[v * PD + d];
        }
        double ev[3][3] = {};
        for (int d = 0; d < PD; d++) {
            if (HEX) {
                ev[0][d] = 0.25 * ((X[1][d] + X[2][d] + X[5][d] + X[6][d]) - (X[0][d] + X[3][d] + X[4][d] + X[7][d]));
                ev[1][d] = 0.25 * ((X[3][d] + X[2][d] + X[7][d] + X[6][d]) - (X[0][d] + X[1][d] + X[4][d] + X[5][d]));
                ev[2][d] = 0.25 * ((X[4][d] + X[5][d] + X[6][d] + X[7][d]) - (X[0][d] + X[1][d] + X[2][d] + X[3][d]));
                acc[d] += fmax(fabs(ev[0][d]), fmax(fabs(ev[1][d]), fabs(ev[2][d])));
            } else {
                ev[0][d] = 0.5 * ((X[1][d] + X[2][d]) - (X[0][d] + X[3][d]));
                ev[1][d] = 0.5 * ((X[3][d] + X[2][d]) - (X[0][d] + X[1][d]));
                acc[d] += fmax(fabs(ev[0][d]), fabs(ev[1][d]));
            }
        }
        for (int k = 0; k < PD; k++) {
            double n2 = 0.0;
            for (int d = 0; d < PD; d++) { acc[3 + 3 * k + d] += ev[k][d]; n2 = __dadd_rn(n2, __dmul_rn(ev[k][d], ev[k][d])); }
            acc[12 + k] += sqrt(n2);
        }
        unsigned char cls = 0;
        if (HEX && plan_hex_affine(X)) {
            cls = 1;   // an axis-aligned brick: the three edges at node 0 run along x, y, z exactly
            if (X[1][1] == X[0][1] && X[1][2] == X[0][2] && X[3][0] == X[0][0] && X[3][2] == X[0][2] && X[4][0] == X[0][0] &&
                X[4][1] == X[0][1])
                cls = 2;
            acc[15] += 1.0;
        }
        aff[e] = cls;
    }
    __shared__ double sh[4][NRED];
    for (int j = 0; j < NRED; j++) {
        const double s = wave_sum(acc[j]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < NRED) part[(size_t)blockIdx.x * NRED + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
__global__ void __launch_bounds__(64) k_plan_elems_final(int nblocks, const double* __restrict__ part, double* __restrict__ out)
{
    const int j = threadIdx.x;
    if (j >= NRED) return;
    double r = 0.0;
    for (int b = 0; b < nblocks; b++) r += part[(size_t)b * NRED + j];   // fixed order
    out[j] = r;
}

// node coordinates in the coordinates of the mesh's lattice: q = Li x, products and sums rounded one by one as on the host
__global__ void __launch_bounds__(256) k_plan_lattice(int nn, int pd, const double* __restrict__ x, const double* __restrict__ Li,
                                                      double* __restrict__ q)
{
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= (size_t)nn) return;
    double X[3] = {0.0, 0.0, 0.0};
    for (int d = 0; d < pd; d++) X[d] = x[v * pd + d];
    for (int d = 0; d < pd; d++)
        q[v * pd + d] = __dadd_rn(__dadd_rn(__dmul_rn(Li[3 * d], X[0]), __dmul_rn(Li[3 * d + 1], X[1])), __dmul_rn(Li[3 * d + 2], X[2]));
}

// ---- 2. nodes by box --------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long plan_grid(double c, double lo, double h) { return FEM_PLAN_GRID(c, lo, h); }
template <int PD>
__device__ __forceinline__ int plan_box_of(const double* __restrict__ q, size_t v, const PlanGeom& G)
{
    int b[3] = {0, 0, 0};
    for (int d = 0; d < PD; d++) {
        const long long i = plan_grid(q[v * PD + d], G.lo[d], G.h[d]);
        b[d] = (int)min(max(i / G.NB[d], 0LL), (long long)G.nbox[d] - 1);
    }
    return b[0] + G.nbox[0] * (b[1] + G.nbox[1] * b[2]);
}
template <int PD>
__global__ void __launch_bounds__(256) k_plan_node_box(int nn, const double* __restrict__ q, PlanGeom G, int* __restrict__ nbx,
                                                       int* __restrict__ bcount)
{
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= (size_t)nn) return;
    const int b = plan_box_of<PD>(q, v, G);
    nbx[v] = b;
    atomicAdd(&bcount[b], 1);
}
// (the order inside a box is whatever the atomics give: every consumer either sorts its nodes or does not depend on the order)
__global__ void __launch_bounds__(256) k_plan_node_scatter(int nn, const int* __restrict__ nbx, const int* __restrict__ bstart,
                                                           int* __restrict__ bfill, int* __restrict__ border)
{
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= (size_t)nn) return;
    const int b = nbx[v];
    border[bstart[b] + atomicAdd(&bfill[b], 1)] = (int)v;
}
__global__ void __launch_bounds__(256) k_plan_box_flags(int nboxes, const int* __restrict__ bstart, int* __restrict__ flag)
{
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (b < (size_t)nboxes) flag[b] = bstart[b + 1] > bstart[b];
}
__global__ void __launch_bounds__(256) k_plan_sets_init(int nboxes, const int* __restrict__ bstart, const int* __restrict__ idx,
                                                        SetRec* __restrict__ sets)
{
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= (size_t)nboxes) return;
    const int cnt = bstart[b + 1] - bstart[b];
    if (cnt > 0) sets[idx[b]] = SetRec{bstart[b], cnt, 0, 0, 0, 0, 0, 0};
}
__global__ void __launch_bounds__(256) k_plan_mark(int n, const int* __restrict__ nodes, unsigned char* __restrict__ flag)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)n) flag[nodes[i]] = 1;
}

// ---- 3. one workgroup per set ------------------------------------------------------------------------------------------
// bitonic sorts in LDS, n a power of two, every thread of the workgroup calls
__device__ void plan_sort_int(int* a, int n)
{
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int p = threadIdx.x; p < n / 2; p += PL_T) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                const int x = a[i], y = a[l];
                if ((x > y) == ((i & k) == 0)) { a[i] = y; a[l] = x; }
            }
        }
    __syncthreads();
}
__device__ void plan_sort_pairs(unsigned long long* key, int* id, int n)   // by (key, id)
{
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int p = threadIdx.x; p < n / 2; p += PL_T) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), l = i | j;
                const unsigned long long kx = key[i], ky = key[l];
                const int ix = id[i], iy = id[l];
                const bool gt = kx > ky || (kx == ky && ix > iy);
                if (gt == ((i & k) == 0)) { key[i] = ky; key[l] = kx; id[i] = iy; id[l] = ix; }
            }
        }
    __syncthreads();
}
__device__ __forceinline__ int plan_find(const int* a, int n, int v)   // position of v in the ascending list a[0 .. n), or -1
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return (lo < n && a[lo] == v) ? lo : -1;
}
__device__ __forceinline__ int plan_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
// inclusive scans over the threads of the workgroup (Hillis-Steele in LDS; 256 values)
__device__ __forceinline__ int plan_scan_max(int v, int* sh)
{
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < PL_T; o <<= 1) {
        const int t = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : INT_MIN;
        __syncthreads();
        sh[threadIdx.x] = max(sh[threadIdx.x], t);
        __syncthreads();
    }
    return sh[threadIdx.x];
}
__device__ __forceinline__ int plan_scan_add(int v, int* sh)
{
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < PL_T; o <<= 1) {
        const int t = (threadIdx.x >= (unsigned)o) ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    return sh[threadIdx.x];
}

struct PlanOut {   // where the second pass writes (the plan's arrays)
    const int *node_off, *elem_off, *step_off;   // per set (= per tile), exclusive scans of the first pass's counts
    int* hdr;
    int2* steps;
    int* nodes;
    int* elems;
    uint4* conn16;
    unsigned char *flags, *taff;
    int* tinfo;
};

constexpr size_t plan_tile_lds()
{
    return sizeof(int) * ((size_t)PL_MAXN + PL_SORT + PL_NH + PL_MAXN) + sizeof(unsigned long long) * PL_MAXE + sizeof(int) * PL_MAXE +
           sizeof(unsigned short) * ((size_t)PL_MAXE * 8 + PL_MAXE + PL_MAXE + 2 + PL_MAXE) + PL_MAXE;
}

template <bool HEX, bool WRITE>
__global__ void __launch_bounds__(PL_T) k_plan_tile(int nsets, SetRec* __restrict__ sets, const int* __restrict__ border, PlanGeom G,
                                                    const int* __restrict__ conn, const double* __restrict__ q,
                                                    const int* __restrict__ n2e_ptr, const unsigned* __restrict__ n2e,
                                                    const unsigned char* __restrict__ aff, const unsigned char* __restrict__ shared_node,
                                                    PlanCounters* __restrict__ ctr, PlanOut out)
{
    constexpr int NPE = HEX ? 8 : 4, PD = HEX ? 3 : 2, NFP = NPE / 2;
    constexpr int c_plus[4] = {1, 2, 5, 6}, c_minus[4] = {0, 3, 4, 7};
    const int si = blockIdx.x, tid = threadIdx.x;
    if (si >= nsets) return;
    const SetRec rec = sets[si];
    if (rec.status != (WRITE ? 1 : 0)) return;
    const int nown = rec.cnt;
    extern __shared__ unsigned long long plan_lds[];
    unsigned long long* s_ekey = plan_lds;                              // [PL_MAXE] grid key of an element's first node
    int* s_eid = reinterpret_cast<int*>(s_ekey + PL_MAXE);              // [PL_MAXE] the tile's elements, in the order of the grid
    int* s_nodes = s_eid + PL_MAXE;                                     // [PL_MAXN] owned nodes, then the others, each part ascending
    int* s_scr = s_nodes + PL_MAXN;                                     // [PL_SORT] sort scratch / element hash
    int* s_nhash = s_scr + PL_SORT;                                     // [PL_NH]   nodes that are not owned
    int* s_cmask = s_nhash + PL_NH;                                     // [PL_MAXN] colours in use at a node
    unsigned short* s_lconn = reinterpret_cast<unsigned short*>(s_cmask + PL_MAXN);   // [PL_MAXE * 8] tile-local node numbers
    unsigned short* s_chain = s_lconn + PL_MAXE * 8;                    // [PL_MAXE] chain of an element
    unsigned short* s_cstart = s_chain + PL_MAXE;                       // [PL_MAXE + 2] first element of a chain
    unsigned short* s_coff = s_cstart + PL_MAXE + 2;                    // [PL_MAXE] first lane slot of a chain inside its colour
    unsigned char* s_ccol = reinterpret_cast<unsigned char*>(s_coff + PL_MAXE);   // [PL_MAXE] colour of a chain
    __shared__ int s_scan[PL_T];
    __shared__ int s_n[4];             // [0] elements, [1] nodes not owned, [2] chains, [3] colours (or -1: more than TILE_MAXC)
    __shared__ int s_first[TILE_MAXC + 1], s_cnt[TILE_MAXC];

    auto verdict = [&](int status, int code, int a, int b) {   // first pass only: the set cannot be a tile as it is
        if (tid == 0) {
            sets[si].status = status;
            if (status == 2) atomicAdd(&ctr->n_split, 1);
            else if (atomicAdd(&ctr->n_fail, 1) == 0) { ctr->fail_code = code; ctr->fail_a = a; ctr->fail_b = b; }
        }
    };
    auto too_big = [&](int nn_) {   // halve the set, or give up on a single node whose elements alone exceed a tile
        if (nown >= 2) verdict(2, 0, 0, 0);
        else verdict(-1, FAIL_TOO_MANY_NODES, nn_, nown);
    };
    if (!WRITE && (nown > G.max_own || nown > G.max_nodes)) { too_big(nown); return; }

    // owned nodes, ascending
    for (int i = tid; i < PL_SORT; i += PL_T) s_scr[i] = (i < nown) ? border[rec.start + i] : INT_MAX;
    plan_sort_int(s_scr, plan_pow2(nown));
    for (int i = tid; i < nown; i += PL_T) s_nodes[i] = s_scr[i];
    __syncthreads();
    // the elements that touch an owned node: set in LDS (open addressing), list in s_eid
    for (int i = tid; i < PL_SORT; i += PL_T) s_scr[i] = -1;
    if (tid < 4) s_n[tid] = 0;
    __syncthreads();
    for (int k = tid; k < nown; k += PL_T) {
        const int v = s_nodes[k];
        const int q0 = n2e_ptr[v], q1 = n2e_ptr[v + 1];
        for (int qq = q0; qq < q1; qq++) {
            if (*(volatile int*)&s_n[0] > PL_MAXE) break;   // too many already: the set will be halved
            const int e = (int)((n2e[qq] & 0x3fffffffu) >> 3);
            unsigned h = ((unsigned)e * 2654435761u) & (PL_SORT - 1);
            while (true) {
                const int prev = atomicCAS(&s_scr[h], -1, e);
                if (prev == -1) { const int idx = atomicAdd(&s_n[0], 1); if (idx < PL_MAXE) s_eid[idx] = e; break; }
                if (prev == e) break;
                h = (h + 1) & (PL_SORT - 1);
            }
        }
    }
    __syncthreads();
    const int nel = s_n[0];
    if (nel > PL_MAXE) { if (!WRITE) too_big(nel); return; }
    // ... in the order of the grid: z, y, x of the first node in box coordinates, then the element id
    const int nel2 = plan_pow2(max(nel, 1));
    for (int i = tid; i < nel2; i += PL_T) {
        if (i < nel) {
            const size_t v = (size_t)conn[(size_t)s_eid[i] * NPE];
            unsigned long long k = 0;
            for (int d = PD - 1; d >= 0; d--) {
                const long long g = plan_grid(q[v * PD + d], G.lo[d], G.h[d]) + 4;
                k = (k << 21) | (unsigned long long)min(max(g, 0LL), (1LL << 21) - 1);
            }
            s_ekey[i] = k;
        } else { s_ekey[i] = ~0ull; s_eid[i] = INT_MAX; }
    }
    plan_sort_pairs(s_ekey, s_eid, nel2);
    // the other nodes of those elements: set in LDS, list in the scratch, then sorted behind the owned ones
    for (int i = tid; i < PL_NH; i += PL_T) s_nhash[i] = -1;
    __syncthreads();
    for (int idx = tid; idx < nel * NPE; idx += PL_T) {
        const int v = conn[(size_t)s_eid[idx / NPE] * NPE + (idx % NPE)];
        if (plan_find(s_nodes, nown, v) >= 0) continue;
        if (*(volatile int*)&s_n[1] > PL_MAXN) continue;   // too many already
        unsigned h = ((unsigned)v * 2654435761u) & (PL_NH - 1);
        while (true) {
            const int prev = atomicCAS(&s_nhash[h], -1, v);
            if (prev == -1) { const int id = atomicAdd(&s_n[1], 1); if (id < PL_SORT) s_scr[id] = v; break; }
            if (prev == v) break;
            h = (h + 1) & (PL_NH - 1);
        }
    }
    __syncthreads();
    const int nno = s_n[1], nnode = nown + nno;
    if (nnode > G.max_nodes) { if (!WRITE) too_big(nnode); return; }
    const int nno2 = plan_pow2(max(nno, 1));
    for (int i = nno + tid; i < nno2; i += PL_T) s_scr[i] = INT_MAX;
    plan_sort_int(s_scr, nno2);
    for (int i = tid; i < nno; i += PL_T) s_nodes[nown + i] = s_scr[i];
    for (int i = tid; i < nnode; i += PL_T) s_cmask[i] = 0;
    __syncthreads();
    // tile-local connectivity
    for (int idx = tid; idx < nel * NPE; idx += PL_T) {
        const int v = conn[(size_t)s_eid[idx / NPE] * NPE + (idx % NPE)];
        int p = plan_find(s_nodes, nown, v);
        if (p < 0) p = nown + plan_find(s_nodes + nown, nno, v);
        s_lconn[idx] = (unsigned short)p;
    }
    __syncthreads();
    // chains: runs of elements (in that order) whose +x face is the next one's -x face, at most 16 long.  Element i starts a
    // chain when it does not continue its predecessor or sits a multiple of 16 behind the start of its run.
    int nchain;
    {
        const int i0 = tid * PL_SEG;
        bool cont[PL_SEG];   // element i continues element i - 1
        int lastbreak = INT_MIN;
#pragma unroll
        for (int k = 0; k < PL_SEG; k++) {
            const int i = i0 + k;
            bool c = i > 0 && i < nel;
            if (c)
                for (int f = 0; f < NFP; f++) c = c && s_lconn[(i - 1) * NPE + c_plus[f]] == s_lconn[i * NPE + c_minus[f]];
            cont[k] = c;
            if (i < nel && !c) lastbreak = i;
        }
        const int incl = plan_scan_max(lastbreak, s_scan);
        int run = (tid > 0) ? s_scan[tid - 1] : INT_MIN;   // start of the run the segment's first element continues
        (void)incl;
        bool starts[PL_SEG];
        int ns = 0;
#pragma unroll
        for (int k = 0; k < PL_SEG; k++) {
            const int i = i0 + k;
            if (i < nel && !cont[k]) run = i;
            starts[k] = i < nel && ((i - run) & 15) == 0;
            ns += starts[k];
        }
        const int upto = plan_scan_add(ns, s_scan);
        int ch = upto - ns - 1;   // chain of the element before the segment
#pragma unroll
        for (int k = 0; k < PL_SEG; k++) {
            const int i = i0 + k;
            if (i >= nel) break;
            if (starts[k]) { ch++; s_cstart[ch] = (unsigned short)i; }
            s_chain[i] = (unsigned short)ch;
        }
        nchain = s_scan[PL_T - 1];
        if (tid == 0) s_cstart[nchain] = (unsigned short)nel;
    }
    // class of the tile: every element a parallelepiped / an axis-aligned brick; does it own a node shared with another domain
    int cls_ok = 3, ifc = 0;
    for (int i = tid; i < nel; i += PL_T) { const int c = aff[s_eid[i]]; cls_ok &= (c != 0 ? 1 : 0) | (c == 2 ? 2 : 0); }
    if (shared_node)
        for (int k = tid; k < nown; k += PL_T) ifc |= shared_node[s_nodes[k]];
    const int all_aff = __syncthreads_and(cls_ok & 1), all_cart = __syncthreads_and(cls_ok & 2);
    const int iface = __syncthreads_or(ifc);
    const int info = (all_cart ? 2 : all_aff ? 1 : 0) | (iface ? 4 : 0);
    // greedy colouring of the chains in their order (two chains of one colour share no node): one wavefront, its lanes cover
    // the node references of a chain (<= 16 elements x NPE nodes)
    if (tid < 64) {
        int ncol = 0;
        for (int ch = 0; ch < nchain && ncol >= 0; ch++) {
            const int c0 = s_cstart[ch], refs = (s_cstart[ch + 1] - c0) * NPE;
            int l0 = -1, l1 = -1, u0 = 0, u1 = 0;
            if (tid < refs) { l0 = s_lconn[c0 * NPE + tid]; u0 = s_cmask[l0]; }
            if (tid + 64 < refs) { l1 = s_lconn[c0 * NPE + 64 + tid]; u1 = s_cmask[l1]; }
            const int used = u0 | u1;
            int c = 0;
            while (c < TILE_MAXC && __ballot((used >> c) & 1) != 0ull) c++;
            if (c >= TILE_MAXC) { ncol = -1; break; }
            // (lanes that hold the same node write the same word: every lane has read it before any lane writes)
            if (l0 >= 0) s_cmask[l0] = u0 | (1 << c);
            if (l1 >= 0) s_cmask[l1] = u1 | (1 << c);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (tid == 0) s_ccol[ch] = (unsigned char)c;
            ncol = max(ncol, c + 1);
        }
        if (tid == 0) s_n[3] = ncol;
    }
    __syncthreads();
    const int ncol = s_n[3];
    if (ncol < 0) { if (!WRITE) verdict(-1, FAIL_COLOURS, nel, nchain); return; }
    // lane slots: colour by colour, a chain does not straddle 16-lane rows, every colour ends on a row boundary
    if (tid < TILE_MAXC) {
        int pos = 0;
        if (tid < ncol)
            for (int ch = 0; ch < nchain; ch++) {
                if (s_ccol[ch] != tid) continue;
                const int len = s_cstart[ch + 1] - s_cstart[ch], used_in_row = pos & 15;
                if (used_in_row + len > 16) pos += 16 - used_in_row;
                s_coff[ch] = (unsigned short)pos;
                pos += len;
            }
        s_cnt[tid] = (pos + 15) & ~15;
    }
    __syncthreads();
    if (tid == 0) {
        int f = 0;
        for (int c = 0; c < TILE_MAXC; c++) { s_first[c] = f; f += s_cnt[c]; }
        s_first[TILE_MAXC] = f;
    }
    __syncthreads();
    const int nslots = s_first[TILE_MAXC];
    int nsteps = 0;
    for (int c = 0; c < ncol; c++) nsteps += (s_cnt[c] + 255) / 256;
    if constexpr (!WRITE) {
        if (tid == 0) {
            SetRec r = rec;
            r.status = 1; r.nnode = nnode; r.nslots = nslots; r.nsteps = nsteps; r.nel = nel; r.info = info;
            sets[si] = r;
            atomicMax(&ctr->max_nnode, nnode); atomicMax(&ctr->max_nsteps, nsteps); atomicMax(&ctr->max_nown, nown);
            atomicAdd(&ctr->visits, (unsigned long long)nel); atomicAdd(&ctr->slots, (unsigned long long)nslots);
            atomicAdd(&ctr->nodes, (unsigned long long)nnode);
        }
    } else {
        const int t = si;   // every set is a tile by now
        const int node_off = out.node_off[si], elem_off = out.elem_off[si], step_off = out.step_off[si];
        for (int i = tid; i < nnode; i += PL_T) out.nodes[(size_t)node_off + i] = s_nodes[i];
        const unsigned char cls = (unsigned char)(info & 3);
        for (int p = tid; p < nslots; p += PL_T) {   // the padding of the 16-lane rows; the elements overwrite their slots below
            const size_t k = (size_t)elem_off + p;
            out.elems[k] = -1; out.conn16[k] = make_uint4(0, 0, 0, 0); out.flags[k] = 0; out.taff[k] = cls;
        }
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < nel; i += PL_T) {
            const int ch = s_chain[i];
            const size_t k = (size_t)elem_off + s_first[s_ccol[ch]] + s_coff[ch] + (i - s_cstart[ch]);
            unsigned l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int a = 0; a < NPE; a++) l[a] = s_lconn[i * NPE + a];
            out.elems[k] = s_eid[i];
            out.conn16[k] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
            out.flags[k] = (unsigned char)(2 | (i + 1 < s_cstart[ch + 1] ? 1 : 0));   // bit 1: an element, bit 0: linked to the next lane
        }
        if (tid == 0) {
            int k = 0;
            for (int c = 0; c < ncol; c++)
                for (int r = 0; r < s_cnt[c]; r += 256) {
                    const int n = min(256, s_cnt[c] - r);
                    out.steps[(size_t)step_off + k++] = make_int2(s_first[c] + r, (r + 256 >= s_cnt[c]) ? (int)((unsigned)n | 0x80000000u) : n);
                }
            int* hd = out.hdr + (size_t)6 * t;
            hd[0] = node_off; hd[1] = nown; hd[2] = nnode; hd[3] = elem_off; hd[4] = step_off; hd[5] = nsteps;
            out.tinfo[t] = info;
        }
    }
}

// halves a set that cannot be a tile (same rule as the host builder's split_own): across the longest extent in grid
// spacings, at the middle of that extent less a millionth of a spacing; the range of `border` is partitioned in place
template <int PD>
__global__ void __launch_bounds__(PL_T) k_plan_split(int nsets, SetRec* __restrict__ sets, int* __restrict__ border, int* __restrict__ tmp,
                                                     PlanGeom G, const double* __restrict__ q, PlanCounters* __restrict__ ctr)
{
    const int si = blockIdx.x, tid = threadIdx.x;
    if (si >= nsets) return;
    const SetRec rec = sets[si];
    if (rec.status != 2) return;
    __shared__ double sh[4][6];
    __shared__ int s_lo, s_hi;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int i = tid; i < rec.cnt; i += PL_T) {
        const size_t v = (size_t)border[rec.start + i];
        for (int d = 0; d < PD; d++) { const double c = q[v * PD + d]; mn[d] = fmin(mn[d], c); mx[d] = fmax(mx[d], c); }
    }
    for (int d = 0; d < 3; d++) {
        for (int o = 32; o > 0; o >>= 1) { mn[d] = fmin(mn[d], __shfl_down(mn[d], o)); mx[d] = fmax(mx[d], __shfl_down(mx[d], o)); }
        if ((tid & 63) == 0) { sh[tid >> 6][d] = mn[d]; sh[tid >> 6][3 + d] = mx[d]; }
    }
    if (tid == 0) { s_lo = 0; s_hi = 0; }
    __syncthreads();
    for (int d = 0; d < 3; d++) {
        mn[d] = fmin(fmin(sh[0][d], sh[1][d]), fmin(sh[2][d], sh[3][d]));
        mx[d] = fmax(fmax(sh[0][3 + d], sh[1][3 + d]), fmax(sh[2][3 + d], sh[3][3 + d]));
    }
    int dl = 0;
    for (int d = 1; d < PD; d++)
        if ((mx[d] - mn[d]) / G.h[d] > (mx[dl] - mn[dl]) / G.h[dl]) dl = d;
    const double eps = G.eps[dl];
    if (!(mx[dl] - mn[dl] > 2.0 * eps)) {
        if (tid == 0) {
            sets[si].status = -2;
            if (atomicAdd(&ctr->n_fail, 1) == 0) { ctr->fail_code = FAIL_COINCIDENT; ctr->fail_a = rec.cnt; ctr->fail_b = 0; }
        }
        return;
    }
    const double cut = 0.5 * (mn[dl] + mx[dl]) - eps;   // (0.5 * sum is exact: contracted or not, one rounding)
    int mine = 0;
    for (int i = tid; i < rec.cnt; i += PL_T) mine += q[(size_t)border[rec.start + i] * PD + dl] < cut;
    __shared__ int s_cntlo[PL_T / 64];
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
    if ((tid & 63) == 0) s_cntlo[tid >> 6] = mine;
    __syncthreads();
    const int nlo = s_cntlo[0] + s_cntlo[1] + s_cntlo[2] + s_cntlo[3];
    for (int i = tid; i < rec.cnt; i += PL_T) {
        const int v = border[rec.start + i];
        const bool lo = q[(size_t)v * PD + dl] < cut;
        tmp[rec.start + (lo ? atomicAdd(&s_lo, 1) : nlo + atomicAdd(&s_hi, 1))] = v;
    }
    __syncthreads();
    for (int i = tid; i < rec.cnt; i += PL_T) border[rec.start + i] = tmp[rec.start + i];
    if (tid == 0) sets[si].nnode = nlo;   // (read by k_plan_sets_expand)
}
// the set list with every halved set replaced by its two halves (the lower half first: tiles stay in box order)
__global__ void __launch_bounds__(256) k_plan_split_flags(int nsets, const SetRec* __restrict__ sets, int* __restrict__ flag)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)nsets) flag[i] = sets[i].status == 2;
}
__global__ void __launch_bounds__(256) k_plan_sets_expand(int nsets, const SetRec* __restrict__ sets, const int* __restrict__ before,
                                                          SetRec* __restrict__ nsets_out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nsets) return;
    const SetRec r = sets[i];
    const size_t j = i + (size_t)before[i];
    if (r.status != 2) { nsets_out[j] = r; return; }
    nsets_out[j] = SetRec{r.start, r.nnode, 0, 0, 0, 0, 0, 0};
    nsets_out[j + 1] = SetRec{r.start + r.nnode, r.cnt - r.nnode, 0, 0, 0, 0, 0, 0};
}
__global__ void __launch_bounds__(256) k_plan_counts(int nsets, const SetRec* __restrict__ sets, int* __restrict__ nnode,
                                                     int* __restrict__ nslots, int* __restrict__ nsteps)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)nsets) return;
    nnode[i] = sets[i].nnode; nslots[i] = sets[i].nslots; nsteps[i] = sets[i].nsteps;
}

}  // namespace

// ---- the builder -------------------------------------------------------------------------------------------------------
std::shared_ptr<EbeTilePlan> ebe_tile_build_device(fem_mesh m, hipStream_t s)
{
    if (m->groups.size() != 1 || !m->pattern) return nullptr;
    const bool hex = m->groups[0].type == FEM_HEX && m->pdim == 3;
    if (!hex && !(m->groups[0].type == FEM_QUA && m->pdim == 2)) return nullptr;
    const int npe = hex ? 8 : 4, pd = hex ? 3 : 2;
    static const bool verbose = std::getenv("FEM_AMD_EBE_VERBOSE") != nullptr;
    auto refuse = [&](const char* why, long long a = 0, long long b = 0) -> std::shared_ptr<EbeTilePlan> {
        if (verbose) std::fprintf(stderr, "[fem_amd] ebe tiles: not used (%s %lld %lld)\n", why, a, b);
        return nullptr;
    };
    auto& grp = m->groups[0];
    const int nel = grp.nel, nn = m->nnds;
    if (nel < 1) return nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto tl = t0;
    auto lap = [&](const char* what) {
        if (!verbose) return;
        FEM_HIP(hipStreamSynchronize(s));
        const auto tn = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[fem_amd] ebe tiles set-up (device): %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(tn - tl).count());
        tl = tn;
    };
    const MatPattern& pat = *m->pattern;
    // 1. reductions: bounding box, mean spacings, the mesh's lattice, element classes
    constexpr int RB = 512;   // workgroups of the reductions (fixed: the order of the sums does not depend on the machine)
    DBuf<double> part((size_t)RB * NRED), red(6 + NRED);
    DBuf<unsigned char> aff((size_t)nel);
    hipLaunchKernelGGL(k_plan_bbox, dim3(RB), dim3(256), 0, s, nn, pd, m->coords.p, part.p);
    hipLaunchKernelGGL(k_plan_bbox_final, dim3(1), dim3(64), 0, s, RB, part.p, red.p);
    if (hex) hipLaunchKernelGGL(k_plan_elems<true>, dim3(RB), dim3(256), 0, s, nel, grp.conn.p, m->coords.p, aff.p, part.p);
    else hipLaunchKernelGGL(k_plan_elems<false>, dim3(RB), dim3(256), 0, s, nel, grp.conn.p, m->coords.p, aff.p, part.p);
    hipLaunchKernelGGL(k_plan_elems_final, dim3(1), dim3(64), 0, s, RB, part.p, red.p + 6);
    double hr[6 + NRED];
    int n2e_total = 0;
    FEM_HIP(hipMemcpyAsync(hr, red.p, sizeof(hr), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipMemcpyAsync(&n2e_total, pat.n2e_ptr.p + nn, sizeof(int), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipStreamSynchronize(s));
    if (n2e_total != nel * npe) return refuse("node lists do not match the group");
    lap("reductions");
    double lo[3] = {hr[0], hr[1], hr[2]}, hi[3] = {hr[3], hr[4], hr[5]}, hsum[3] = {hr[6], hr[7], hr[8]};
    const double* eg = hr + 6 + 3;   // [3 k + d] signed mean edge vectors, [9 + k] their lengths
    const long long npar = (long long)std::llround(hr[6 + 15]);
    // box size: as the host builder (FEM_AMD_TILE_NB="nx,ny,nz" for experiments; 15,15,7 won the sweep at config 3)
    int NB[3] = {hex ? 15 : 31, hex ? 15 : 31, hex ? 7 : 1};
    if (hex && npar * 10 >= (long long)nel * 9) { NB[1] = 7; NB[2] = 7; }   // a mesh of parallelepipeds: small tiles, x of a tile in LDS
    if (const char* ps = std::getenv("FEM_AMD_TILE_NB")) {
        int qv[3] = {0, 0, 0};
        if (std::sscanf(ps, "%d,%d,%d", &qv[0], &qv[1], &qv[2]) == 3 && qv[0] > 0 && qv[1] > 0 && qv[2] > 0)
            for (int d = 0; d < pd; d++) NB[d] = qv[d];
    }
    const bool small_tiles = hex && NB[1] <= 7 && NB[2] <= 7;
    PlanGeom G{};
    G.max_nodes = small_tiles ? TILE_X_LDS_NODES : TILE_MAX_NODES;
    G.max_own = small_tiles ? TILE_X_MAX_OWN : 0x7fffffff;
    // the lattice of the mesh (see the host builder): L = mean edge vectors, nodes binned in xi = L^-1 x where the signed
    // edges keep 0.9 of the mean edge lengths
    bool lattice = false;
    DBuf<double> qbuf;
    const double* q = m->coords.p;
    {
        double Lm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, len[3] = {1, 1, 1};
        bool ok = true;
        for (int k = 0; k < pd; k++) {
            double n2 = 0.0;
            for (int d = 0; d < pd; d++) { Lm[d][k] = eg[3 * k + d] / nel; n2 += Lm[d][k] * Lm[d][k]; }
            len[k] = std::sqrt(n2);
            ok = ok && eg[9 + k] > 0.0 && len[k] >= 0.9 * eg[9 + k] / nel;
        }
        const double det = Lm[0][0] * (Lm[1][1] * Lm[2][2] - Lm[1][2] * Lm[2][1]) - Lm[0][1] * (Lm[1][0] * Lm[2][2] - Lm[1][2] * Lm[2][0]) +
                           Lm[0][2] * (Lm[1][0] * Lm[2][1] - Lm[1][1] * Lm[2][0]);
        ok = ok && std::fabs(det) >= 0.1 * len[0] * len[1] * len[2];
        static const bool lattice_env = !(std::getenv("FEM_AMD_TILE_LATTICE") && std::atoi(std::getenv("FEM_AMD_TILE_LATTICE")) == 0);
        if (ok && lattice_env) {
            double Li[9];
            Li[0] = (Lm[1][1] * Lm[2][2] - Lm[1][2] * Lm[2][1]) / det; Li[1] = (Lm[0][2] * Lm[2][1] - Lm[0][1] * Lm[2][2]) / det;
            Li[2] = (Lm[0][1] * Lm[1][2] - Lm[0][2] * Lm[1][1]) / det; Li[3] = (Lm[1][2] * Lm[2][0] - Lm[1][0] * Lm[2][2]) / det;
            Li[4] = (Lm[0][0] * Lm[2][2] - Lm[0][2] * Lm[2][0]) / det; Li[5] = (Lm[0][2] * Lm[1][0] - Lm[0][0] * Lm[1][2]) / det;
            Li[6] = (Lm[1][0] * Lm[2][1] - Lm[1][1] * Lm[2][0]) / det; Li[7] = (Lm[0][1] * Lm[2][0] - Lm[0][0] * Lm[2][1]) / det;
            Li[8] = (Lm[0][0] * Lm[1][1] - Lm[0][1] * Lm[1][0]) / det;
            DBuf<double> dLi(9);
            FEM_HIP(hipMemcpyAsync(dLi.p, Li, sizeof(Li), hipMemcpyHostToDevice, s));
            qbuf.alloc((size_t)nn * pd);
            hipLaunchKernelGGL(k_plan_lattice, dim3(cdiv((size_t)nn, 256)), dim3(256), 0, s, nn, pd, m->coords.p, dLi.p, qbuf.p);
            hipLaunchKernelGGL(k_plan_bbox, dim3(RB), dim3(256), 0, s, nn, pd, qbuf.p, part.p);
            hipLaunchKernelGGL(k_plan_bbox_final, dim3(1), dim3(64), 0, s, RB, part.p, red.p);
            FEM_HIP(hipMemcpyAsync(hr, red.p, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
            FEM_HIP(hipStreamSynchronize(s));   // (dLi goes out of scope)
            for (int d = 0; d < 3; d++) { lo[d] = hr[d]; hi[d] = hr[3 + d]; }
            for (int d = 0; d < pd; d++) hsum[d] = (double)nel;   // unit spacing
            q = qbuf.p;
            lattice = true;
        }
    }
    long long nboxes = 1;
    for (int d = 0; d < 3; d++) { G.lo[d] = 0.0; G.h[d] = 1.0; G.eps[d] = 1e-6; G.NB[d] = 1; G.nbox[d] = 1; }
    for (int d = 0; d < pd; d++) {
        const double hd = hsum[d] / nel;
        if (!(hd > 0.0)) return refuse("flat mesh");
        const long long npts = (long long)std::llround((hi[d] - lo[d]) / hd) + 1;
        const long long nb = (npts + NB[d] - 1) / NB[d];
        if (nb > 0x3fffffffLL) return refuse("grid does not describe the mesh");
        G.lo[d] = lo[d]; G.h[d] = hd; G.eps[d] = 1e-6 * hd; G.NB[d] = NB[d]; G.nbox[d] = (int)nb;
        nboxes *= nb;
        if ((double)nboxes > 4.0 * nn + 1024.0) return refuse("grid does not describe the mesh");
    }
    // 2. nodes by box
    DBuf<int> nbx((size_t)nn), bcount((size_t)nboxes + 1), bstart((size_t)nboxes + 1), bfill((size_t)nboxes + 1), border((size_t)nn);
    DBuf<int> scan_tmp(exclusive_scan_tmp_ints((size_t)std::max<long long>(nboxes, nn) + 1)), bflag((size_t)nboxes + 1), bidx((size_t)nboxes + 1);
    bcount.zero(s); bfill.zero(s);
    if (hex) hipLaunchKernelGGL(k_plan_node_box<3>, dim3(cdiv((size_t)nn, 256)), dim3(256), 0, s, nn, q, G, nbx.p, bcount.p);
    else hipLaunchKernelGGL(k_plan_node_box<2>, dim3(cdiv((size_t)nn, 256)), dim3(256), 0, s, nn, q, G, nbx.p, bcount.p);
    exclusive_scan_i32(bcount.p, bstart.p, (size_t)nboxes, scan_tmp.p, s);
    hipLaunchKernelGGL(k_plan_node_scatter, dim3(cdiv((size_t)nn, 256)), dim3(256), 0, s, nn, nbx.p, bstart.p, bfill.p, border.p);
    hipLaunchKernelGGL(k_plan_box_flags, dim3(cdiv((size_t)nboxes, 256)), dim3(256), 0, s, (int)nboxes, bstart.p, bflag.p);
    exclusive_scan_i32(bflag.p, bidx.p, (size_t)nboxes, scan_tmp.p, s);
    int nsets = 0;
    FEM_HIP(hipMemcpyAsync(&nsets, bidx.p + nboxes, sizeof(int), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipStreamSynchronize(s));
    lap("node boxes");
    // the sets may double a few times (a graded mesh): room for the list and its successor
    size_t set_cap = (size_t)nsets + 1024;
    DBuf<SetRec> sets_a(set_cap), sets_b(set_cap);
    SetRec* sets = sets_a.p;
    SetRec* sets_next = sets_b.p;
    hipLaunchKernelGGL(k_plan_sets_init, dim3(cdiv((size_t)nboxes, 256)), dim3(256), 0, s, (int)nboxes, bstart.p, bidx.p, sets);
    // nodes shared with other domains (a partitioned mesh): tiles that own one are computed first (ksp.hip)
    DBuf<unsigned char> shared_node;
    if (Halo* hl = m->halo.get(); hl && hl->ninodes > 0) {
        shared_node.alloc((size_t)nn);
        shared_node.zero(s);
        hipLaunchKernelGGL(k_plan_mark, dim3(cdiv((size_t)hl->ninodes, 256)), dim3(256), 0, s, hl->ninodes, hl->inodes.p, shared_node.p);
    }
    // 3. first pass: counts; sets that exceed a tile are halved until none is left
    static OncePerDevice once;
    if (once.first(m->ctx->device))
        for (const void* f : {reinterpret_cast<const void*>(k_plan_tile<true, false>), reinterpret_cast<const void*>(k_plan_tile<true, true>),
                              reinterpret_cast<const void*>(k_plan_tile<false, false>), reinterpret_cast<const void*>(k_plan_tile<false, true>)})
            FEM_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan_tile_lds()));
    DBuf<PlanCounters> ctr(1);
    ctr.zero(s);
    DBuf<int> border_tmp;
    PlanCounters hc{};
    PlanOut none{};
    for (int level = 0;; level++) {
        if (level > 60) return refuse("sets still too large after 60 halvings");
        if (hex) hipLaunchKernelGGL((k_plan_tile<true, false>), dim3(nsets), dim3(PL_T), plan_tile_lds(), s, nsets, sets, border.p, G, grp.conn.p, q,
                                    pat.n2e_ptr.p, pat.n2e.p, aff.p, shared_node.p, ctr.p, none);
        else hipLaunchKernelGGL((k_plan_tile<false, false>), dim3(nsets), dim3(PL_T), plan_tile_lds(), s, nsets, sets, border.p, G, grp.conn.p, q,
                                pat.n2e_ptr.p, pat.n2e.p, aff.p, shared_node.p, ctr.p, none);
        FEM_HIP(hipGetLastError());
        FEM_HIP(hipMemcpyAsync(&hc, ctr.p, sizeof(hc), hipMemcpyDeviceToHost, s));
        FEM_HIP(hipStreamSynchronize(s));
        if (hc.n_fail) break;
        if (!hc.n_split) break;
        if (!border_tmp.p) border_tmp.alloc((size_t)nn);
        if (hex) hipLaunchKernelGGL(k_plan_split<3>, dim3(nsets), dim3(PL_T), 0, s, nsets, sets, border.p, border_tmp.p, G, q, ctr.p);
        else hipLaunchKernelGGL(k_plan_split<2>, dim3(nsets), dim3(PL_T), 0, s, nsets, sets, border.p, border_tmp.p, G, q, ctr.p);
        const int nnew = nsets + hc.n_split;
        if ((size_t)nnew > set_cap) {   // (rare: more room, the current list copied over)
            const size_t ncap = (size_t)nnew * 2 + 1024;
            DBuf<SetRec> na(ncap), nb(ncap);
            FEM_HIP(hipMemcpyAsync(na.p, sets, (size_t)nsets * sizeof(SetRec), hipMemcpyDeviceToDevice, s));
            FEM_HIP(hipStreamSynchronize(s));
            sets_a = std::move(na); sets_b = std::move(nb);
            sets = sets_a.p; sets_next = sets_b.p; set_cap = ncap;
        }
        DBuf<int> sflag((size_t)nsets + 1), sbefore((size_t)nsets + 1), stmp(exclusive_scan_tmp_ints((size_t)nsets + 1));
        hipLaunchKernelGGL(k_plan_split_flags, dim3(cdiv((size_t)nsets, 256)), dim3(256), 0, s, nsets, sets, sflag.p);
        exclusive_scan_i32(sflag.p, sbefore.p, (size_t)nsets, stmp.p, s);
        hipLaunchKernelGGL(k_plan_sets_expand, dim3(cdiv((size_t)nsets, 256)), dim3(256), 0, s, nsets, sets, sbefore.p, sets_next);
        FEM_HIP(hipMemsetAsync(&ctr.p->n_split, 0, sizeof(int), s));
        FEM_HIP(hipStreamSynchronize(s));   // (the scan buffers go out of scope)
        std::swap(sets, sets_next);
        nsets = nnew;
    }
    if (hc.n_fail) {
        // (a failure inside the halving kernel is seen one round trip later: read the counters once more)
        FEM_HIP(hipMemcpy(&hc, ctr.p, sizeof(hc), hipMemcpyDeviceToHost));
        const char* why = hc.fail_code == FAIL_TOO_MANY_NODES ? "tile with too many nodes" : hc.fail_code == FAIL_COINCIDENT ? "coincident nodes" : "too many colours";
        return refuse(why, hc.fail_a, hc.fail_b);
    }
    lap("tiles, first pass");
    if (hc.slots > 0x7fffffffULL || hc.nodes > 0x7fffffffULL) return refuse("too many lane slots");
    // offsets
    const int ntiles = nsets;
    DBuf<int> c_nnode((size_t)ntiles + 1), c_nslots((size_t)ntiles + 1), c_nsteps((size_t)ntiles + 1);
    DBuf<int> node_off((size_t)ntiles + 1), elem_off((size_t)ntiles + 1), step_off((size_t)ntiles + 1);
    DBuf<int> stmp(exclusive_scan_tmp_ints((size_t)ntiles + 1));
    hipLaunchKernelGGL(k_plan_counts, dim3(cdiv((size_t)ntiles, 256)), dim3(256), 0, s, ntiles, sets, c_nnode.p, c_nslots.p, c_nsteps.p);
    exclusive_scan_i32(c_nnode.p, node_off.p, (size_t)ntiles, stmp.p, s);
    exclusive_scan_i32(c_nslots.p, elem_off.p, (size_t)ntiles, stmp.p, s);
    exclusive_scan_i32(c_nsteps.p, step_off.p, (size_t)ntiles, stmp.p, s);
    int nsteps_total = 0;
    FEM_HIP(hipMemcpyAsync(&nsteps_total, step_off.p + ntiles, sizeof(int), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipStreamSynchronize(s));
    auto plan = std::make_shared<EbeTilePlan>();
    plan->ntiles = ntiles; plan->max_nnode = hc.max_nnode; plan->max_nsteps = hc.max_nsteps; plan->max_nown = hc.max_nown;
    plan->tile_elements = (long long)hc.visits; plan->slots = (long long)hc.slots; plan->tile_nodes = (long long)hc.nodes;
    plan->hdr.alloc((size_t)ntiles * 6);
    plan->steps.alloc(std::max<size_t>((size_t)nsteps_total, 1)); plan->nodes.alloc(std::max<size_t>((size_t)hc.nodes, 1));
    plan->elems.alloc(std::max<size_t>((size_t)hc.slots, 1)); plan->conn16.alloc(std::max<size_t>((size_t)hc.slots, 1));
    plan->flags.alloc(std::max<size_t>((size_t)hc.slots, 1));
    DBuf<unsigned char> d_taff(std::max<size_t>((size_t)hc.slots, 1));
    DBuf<int> tinfo((size_t)ntiles);
    PlanOut po{node_off.p, elem_off.p, step_off.p, plan->hdr.p, plan->steps.p, plan->nodes.p, plan->elems.p, plan->conn16.p, plan->flags.p, d_taff.p, tinfo.p};
    // second pass: the lists
    if (hex) hipLaunchKernelGGL((k_plan_tile<true, true>), dim3(ntiles), dim3(PL_T), plan_tile_lds(), s, ntiles, sets, border.p, G, grp.conn.p, q,
                                pat.n2e_ptr.p, pat.n2e.p, aff.p, shared_node.p, ctr.p, po);
    else hipLaunchKernelGGL((k_plan_tile<false, true>), dim3(ntiles), dim3(PL_T), plan_tile_lds(), s, ntiles, sets, border.p, G, grp.conn.p, q,
                            pat.n2e_ptr.p, pat.n2e.p, aff.p, shared_node.p, ctr.p, po);
    FEM_HIP(hipGetLastError());
    // tiles by kernel: axis-aligned bricks / other parallelepipeds / the others; in every list the tiles that own a node
    // shared with another domain come first (the per-tile records are small: sorted on the host)
    std::vector<int> h_info((size_t)ntiles);
    FEM_HIP(hipMemcpyAsync(h_info.data(), tinfo.p, (size_t)ntiles * sizeof(int), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipStreamSynchronize(s));
    lap("tiles, second pass");
    std::vector<int> laff, lgen, lcart;
    for (int t = 0; t < ntiles; t++) ((h_info[t] & 3) == 2 ? lcart : (h_info[t] & 3) == 1 ? laff : lgen).push_back(t);
    auto iface_first = [&](std::vector<int>& l) {
        return (int)(std::stable_partition(l.begin(), l.end(), [&](int t) { return (h_info[(size_t)t] & 4) != 0; }) - l.begin());
    };
    plan->n_aff_if = iface_first(laff); plan->n_gen_if = iface_first(lgen); plan->n_cart_if = iface_first(lcart);
    plan->halo_seen = m->halo.get();
    plan->n_aff = (int)laff.size(); plan->n_gen = (int)lgen.size(); plan->n_cart = (int)lcart.size();
    auto up = [&](auto& dbuf, const auto& vec) {
        dbuf.alloc(std::max<size_t>(vec.size(), 1));
        if (!vec.empty()) FEM_HIP(hipMemcpyAsync(dbuf.p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice, s));
    };
    up(plan->list_aff, laff); up(plan->list_gen, lgen); up(plan->list_cart, lcart);
    ebe_tile_list_headers(*plan, !hex, s);
    if (!laff.empty() || !lcart.empty()) ebe_tile_store_geometry(m, *plan, d_taff.p, s);
    if (!hex) ebe_tile_quad_factors(m, *plan, s);
    FEM_HIP(hipStreamSynchronize(s));   // (the scratch buffers go out of scope; the lists were uploaded from host vectors)
    plan->t_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    lap("lists, stored geometry");
    if (verbose)
        std::fprintf(stderr, "[fem_amd] ebe tiles: %d (%d affine, %d Cartesian), <= %d nodes, %lld element visits in %lld lane slots for %d elements, %d steps, "
                             "set-up %.2f ms on the device (%s, %d x %d x %d nodes per box)\n",
                     ntiles, plan->n_aff, plan->n_cart, plan->max_nnode, plan->tile_elements, plan->slots, nel, nsteps_total, plan->t_setup_ms,
                     lattice ? "in the coordinates of the mesh's lattice" : "axis-aligned in space", NB[0], NB[1], NB[2]);
    return plan;
}
