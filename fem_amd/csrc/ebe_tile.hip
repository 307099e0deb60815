// fem_amd/csrc/ebe_tile.hip -- MatMult by element-by-element recomputation, one pass: y = K x for a mesh of hex8 (or,
// same plan in two dimensions, quad4) elements without the element-force buffer of the two-kernel product (elem_kernels.hip: k_ebe_apply + k_ebe_gather,
// 1.3 GB of traffic per product at config 3, half of it that buffer written and read back).
//
// Tiles.  The nodes are binned into boxes (15 x 15 x 7 nodes of a regular grid; any mesh: boxes of the mean element
// size); a tile OWNS the nodes of its box and computes every element that touches one of them -- the elements of a
// one-element rim are computed by two (or more) tiles, 1.3x the arithmetic for 16 x 16 x 8 element tiles, and in
// exchange every owned node receives all of its contributions inside one workgroup: nodal sums live in LDS, y is
// written once, nothing is exchanged between tiles and nothing is atomic.
// Order.  The elements of a tile are coloured at set-up (no two elements of a colour share a node; 8 colours on a
// regular grid) and processed colour by colour with a barrier in between: inside a colour every LDS sum has one writer,
// so the plain read-add-write is race-free and the order of the additions into a node is fixed -- bitwise reproducible.
// p.Kp comes from the owned nodes: sum x_i y_i (penalty rows included), one partial per tile.
// Traffic at config 3: tile lists 0.1 GB + x / coordinates through L2 + E, nu + y = about 0.5 GB.
// Kernels.  k_ebe_tile<KIND>: one tile per 256-thread workgroup, two workgroups per CU, corner values gathered from global
// memory (distorted elements, parallelepipeds; with OP 1, 2 the nodal stress sums over the same tiles).  k_ebe_tile_x<2>:
// meshes of axis-aligned bricks -- smaller tiles, x of the tile in LDS, three workgroups per CU.  k_ebe_tile_q4: quad4.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sched.h>

#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "comm.h"
#include "cg_device.h"
#include "ebe_tile.h"
#include "grid_barrier.h"
#include "hex8_ebe.h"
#include "quad4_ebe.h"
#include "objects.h"

using namespace fem;

namespace fem {

// What a lane needs of its coming elements is fetched ahead of the arithmetic (two waves per SIMD is all the LDS image of a
// tile allows, so the latencies have to be covered inside the lane): the index chain -- local numbers, global ids from
// LDS, E and nu -- two steps ahead, and, registers permitting (the affine kernel), the gathers of x and the stored geometry
// one step ahead.
// Chains.  The lanes of a 16-lane row hold runs of elements whose +x face is the next lane's -x face (set-up): such a lane
// loads only its -x face nodes and takes the others from the next lane (DPP row shift), and hands its +x face forces to
// the next lane instead of adding them to LDS itself -- half the gathers, half the LDS traffic, and consecutive lanes read
// consecutive nodes.
struct TileIdx {
    uint4 c16;               // eight tile-local node numbers
    int v[8];                // their global ids (+x face: only where the chain ends)
    double E, nu;            // (general kernel)
    unsigned fl;             // bit 0: linked to the next lane
    bool valid;
};
struct TileData {
    double u[8][3];
    double geo[11];          // a parallelepiped's stored geometry: adj(J) row by row, then the two Lame factors (hex_affine_geom)
};
__device__ constexpr int c_tplus[4] = {1, 2, 5, 6}, c_tminus[4] = {0, 3, 4, 7};
__device__ __forceinline__ int tile_loc(const uint4& c, int a)
{
    const unsigned w = a < 2 ? c.x : a < 4 ? c.y : a < 6 ? c.z : c.w;
    return (a & 1) ? (int)(w >> 16) : (int)(w & 0xffffu);
}
// value of the next / the previous lane of the 16-lane row (DPP row_shl:1 / row_shr:1 with bound_ctrl: a lane without a source
// reads 0 and no register has to be zeroed for it beforehand -- those lanes never use the value)
__device__ __forceinline__ double row_next(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x101, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x101, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_prev(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// The tile lists (slot records, flags, stored geometry, node lists) are read exactly once per product, 57 B per lane slot = 260 MB at
// config 3, beside 2 x 81 MB of x and y that the CG's vector kernels want to find in the L2 / Infinity Cache again: on a plan whose
// lists and vectors do not fit those caches (EbeTilePlan::stream) the product kernels of hex8 meshes load the lists with the streaming
// hint (global_load ... nt).  Config 3: product 0.1384 -> 0.1296 ms, CG iteration 0.2782 -> 0.2710 ms (0.2572 with the hints of the
// CG's own kernels, ksp.hip).  Where everything fits, the next product finds the lists in the cache and the hint costs: 100^3
// elements 0.0443 -> 0.0488 ms per product, config 2 42.6 -> 44.5 us per iteration -- hence plain loads there.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool STREAM, class T>
__device__ __forceinline__ T ld_list(const T* p) { if constexpr (STREAM) return __builtin_nontemporal_load(p); else return *p; }
template <bool STREAM>
__device__ __forceinline__ uint4 ld_list(const uint4* p)
{
    if constexpr (!STREAM) return *p;
    else {
        const u32x4 c = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(c.x, c.y, c.z, c.w);
    }
}
// stage 1 (three steps ahead): the lane's slot from the tile lists -- plain global loads, nothing depends on them yet
struct TileRaw {
    uint4 c16;
    double E, nu;
    unsigned fl;
    bool valid;
};
template <bool AFFINE, bool STREAM>
__device__ __forceinline__ void tile_fetch_raw(TileRaw& rw, bool inside, size_t k, const int* __restrict__ telems,
                                               const uint4* __restrict__ tconn, const unsigned char* __restrict__ tflags,
                                               const double* __restrict__ Ev, const double* __restrict__ nuv)
{
    rw.valid = false; rw.fl = 0;
    if (!inside) return;
    const int e = ld_list<STREAM>(telems + k);
    rw.c16 = ld_list<STREAM>(tconn + k);
    rw.fl = ld_list<STREAM>(tflags + k);
    if (e < 0) return;   // padding of a 16-lane row
    rw.valid = true;
    if constexpr (!AFFINE) { rw.E = Ev[e]; rw.nu = nuv[e]; }
}
// stage 2 (two steps ahead): global node ids from the tile's list in LDS
template <bool CHAINS = true>   // CHAINS false (the general kernel): every node id, the links are not used
__device__ __forceinline__ void tile_resolve(TileIdx& in, const TileRaw& rw, const int* gl)
{
    in.valid = rw.valid; in.fl = rw.valid ? rw.fl : 0u; in.c16 = rw.c16; in.E = rw.E; in.nu = rw.nu;
    if (!rw.valid) return;
#pragma unroll
    for (int i = 0; i < 4; i++) in.v[c_tminus[i]] = gl[tile_loc(in.c16, c_tminus[i])];
    if (!CHAINS || !(in.fl & 1u)) {
#pragma unroll
        for (int i = 0; i < 4; i++) in.v[c_tplus[i]] = gl[tile_loc(in.c16, c_tplus[i])];
    }
}
// the three values of a node in an interleaved vector: 8 B aligned, one 16 B and one 8 B load (three 8 B loads walk the same
// cache lines three times through the address unit)
struct alignas(8) Node3 { double a, b, c; };
__device__ __forceinline__ void load_node3(double (&o)[3], const double* __restrict__ v, int node)
{
    const Node3 t = *reinterpret_cast<const Node3*>(v + (size_t)node * 3);
    o[0] = t.a; o[1] = t.b; o[2] = t.c;
}
// loads of one element: -x face nodes always, +x face nodes where the chain ends (the others come from the next lane, see
// tile_complete)
template <int NG, bool STREAM>
__device__ __forceinline__ void tile_fetch_data(TileData& dt, const TileIdx& in, size_t k, const double* __restrict__ geo,
                                                size_t geo_stride, const double* __restrict__ x)
{
    if (!in.valid) return;
#pragma unroll
    for (int j = 0; j < NG; j++) dt.geo[j] = ld_list<STREAM>(geo + (size_t)j * geo_stride + k);
#pragma unroll
    for (int i = 0; i < 4; i++) load_node3(dt.u[c_tminus[i]], x, in.v[c_tminus[i]]);
    if (!(in.fl & 1u)) {
#pragma unroll
        for (int i = 0; i < 4; i++) load_node3(dt.u[c_tplus[i]], x, in.v[c_tplus[i]]);
    }
}
// every lane of the wavefront: +x face values of the linked lanes from the next lane's -x face
template <int N>
__device__ __forceinline__ void tile_complete(double (&u)[8][N], unsigned fl)
{
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int d = 0; d < N; d++) {
            const double nx = row_next(u[c_tminus[i]][d]);
            if (fl & 1u) u[c_tplus[i]][d] = nx;
        }
}

// End of a tile's workgroup: its p.Kp partial, and -- where the caller wants the sum (the CG) -- the workgroup that finishes
// last adds all the tiles' partials in a fixed order (thread i: partials i, i + 256, ...; then the fixed tree of the waves), so
// that no separate reduction kernel follows the product and the result does not depend on which workgroup that was.
__device__ __forceinline__ void tile_finish(double dot, int t, double* sh, double* __restrict__ partial, unsigned* __restrict__ rcount,
                                            int rtotal, double* __restrict__ red)
{
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (!rcount) {
        if (threadIdx.x == 0) partial[t] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        return;
    }
    // Experimental (FEM_AMD_SELF_REDUCE=1, off by default).  Release / acquire at agent scope as the HIP memory model asks: the
    // counter's acq_rel read-modify-write publishes this workgroup's partial and, in the workgroup whose count comes last, makes
    // every other partial visible.  The release writes back this XCD's L2 -- the kernel's own y -- once per workgroup: measured at
    // 3x the kernel time, which is why the separate reduction kernel (8 us) is the default.  (Round 3 had relaxed atomics with an
    // s_waitcnt and sc1 accesses here: faster, but correct only by this part's cache behaviour, not by the model.)
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partial[t], (sh[0] + sh[1]) + (sh[2] + sh[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = __hip_atomic_fetch_add(rcount, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(rtotal - 1);
    }
    __syncthreads();
    if (!s_last) return;
    double sum = 0.0;
    for (int i = threadIdx.x; i < rtotal; i += 256) sum += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { *red = (sh[0] + sh[1]) + (sh[2] + sh[3]); *rcount = 0u; }
}

// geometry records of the elements of the affine tiles (set-up): geo[j * stride + k], k = position in the tile lists
__global__ void __launch_bounds__(256) k_tile_geom(size_t ntel, const int* __restrict__ telems, const unsigned char* __restrict__ taff,
                                                   const int* __restrict__ conn, const double* __restrict__ coords,
                                                   const double* __restrict__ Ev, const double* __restrict__ nuv, double w,
                                                   double* __restrict__ geo)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntel || !taff[k] || telems[k] < 0) return;
    const size_t e = (size_t)telems[k];
    double X[4][3];
    constexpr int xa[4] = {0, 1, 3, 4};
    for (int i = 0; i < 4; i++)
        for (int d = 0; d < 3; d++) X[i][d] = coords[(size_t)conn[e * 8 + xa[i]] * 3 + d];
    double J[3][3], A[3][3], lw, mw;
    hex_affine_jac(X[0], X[1], X[2], X[3], J);
    const DCoef D = dmat_coef(Ev[e], nuv[e]);
    hex_affine_geom(J, w, D.d12, D.d44, A, lw, mw);
    if (taff[k] == 2) {   // a tile of axis-aligned bricks: the diagonal of adj(J) and the two factors
        for (int i = 0; i < 3; i++) geo[(size_t)i * ntel + k] = A[i][i];
        geo[(size_t)3 * ntel + k] = lw;
        geo[(size_t)4 * ntel + k] = mw;
        return;
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) geo[(size_t)(3 * i + j) * ntel + k] = A[i][j];
    geo[(size_t)9 * ntel + k] = lw;
    geo[(size_t)10 * ntel + k] = mw;
}

// Tiles whose elements all carry the same geometry record, bit for bit (a brick of one spacing and one material: every tile): the
// position of that record goes into the tile's list header (lhdr[8 i + 7] = slot + 1, 0 otherwise) and the product reads 40 B (88 B)
// per TILE instead of per lane slot.  One workgroup per tile of the list; NG words per record.
template <int NG>
__global__ void __launch_bounds__(256) k_tile_geo_uniform(int nlist, int* __restrict__ lhdr, const int2* __restrict__ steps,
                                                          const int* __restrict__ telems, const double* __restrict__ geo,
                                                          size_t stride, unsigned long long* __restrict__ count)
{
    const int i = blockIdx.x;
    if (i >= nlist) return;
    const int* hd = lhdr + 8 * (size_t)i;
    const int elem_off = hd[3], step_off = hd[4], nsteps = hd[5];
    __shared__ int s_ref, s_diff;
    if (threadIdx.x == 0) { s_ref = 0x7fffffff; s_diff = 0; }
    __syncthreads();
    int mine = 0x7fffffff;   // first slot of the tile that holds an element
    for (int s = 0; s < nsteps; s++) {
        const int2 st = steps[step_off + s];
        if ((int)threadIdx.x < (st.y & 0x7fffffff) && telems[(size_t)elem_off + st.x + threadIdx.x] >= 0) mine = min(mine, st.x + (int)threadIdx.x);
    }
    atomicMin(&s_ref, mine);
    __syncthreads();
    const int ref = s_ref;
    if (ref == 0x7fffffff) return;   // (no element: the header keeps 0)
    long long rv[NG];
    for (int j = 0; j < NG; j++) rv[j] = __double_as_longlong(geo[(size_t)j * stride + elem_off + ref]);
    int diff = 0;
    for (int s = 0; s < nsteps; s++) {
        const int2 st = steps[step_off + s];
        if ((int)threadIdx.x >= (st.y & 0x7fffffff)) continue;
        const size_t k = (size_t)elem_off + st.x + threadIdx.x;
        if (telems[k] < 0) continue;
        for (int j = 0; j < NG; j++) diff |= (__double_as_longlong(geo[(size_t)j * stride + k]) != rv[j]);
    }
    if (diff) s_diff = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !s_diff) { lhdr[8 * (size_t)i + 7] = elem_off + ref + 1; atomicAdd(count, 1ull); }
}

// one tile per workgroup.  LDS: acc[3 * max_nown] sums of the owned nodes, then gl[nnode] global node ids.
// steps[step_off .. step_off + nsteps): {first element (relative), count | barrier-after << 31}: rounds of 256 elements
// of one colour, a barrier behind the last round of a colour
// OP 0: y = K x.  OP 1, 2: the nodal stress sums of RecoverNodalStress + src/main.F90:258-279 over the same tiles (x = the
// displacements; y = component 0 / 3 of stress_node(nnds, 6), column-major; divided by the valence where one is given)
template <int KIND, int OP = 0, bool STREAM = false>   // KIND 0: distorted elements, 1: parallelepipeds (stored adj(J)), 2: axis-aligned bricks (stored diagonal); STREAM: see ld_list
#ifndef FEM_EXP_GEN_WAVES
#define FEM_EXP_GEN_WAVES 2
#endif
__global__ void __launch_bounds__(256, FEM_EXP_GEN_WAVES) k_ebe_tile(int nlist, const int* __restrict__ lhdr,
                                                     const int2* __restrict__ steps, const int* __restrict__ tnodes,
                                                     const int* __restrict__ telems, const uint4* __restrict__ tconn,
                                                     const unsigned char* __restrict__ tflags,
                                                     const double* __restrict__ coords, const double* __restrict__ Ev,
                                                     const double* __restrict__ nuv, const double* __restrict__ x,
                                                     const int* __restrict__ bc_node, const int* __restrict__ bc_dof,
                                                     const double* __restrict__ bc_shift, const double* __restrict__ geo,
                                                     size_t geo_stride, double g, double w, int max_nown,
                                                     double* __restrict__ y, double* __restrict__ partial,
                                                     const int* __restrict__ done, const double* __restrict__ valence,
                                                     size_t nn, unsigned* __restrict__ rcount, int rtotal, double* __restrict__ red)
{
    if (done && *done) return;
    static_assert(OP == 0 || KIND == 0, "the stress passes read the coordinates");
    constexpr bool AFFINE = KIND != 0;
    extern __shared__ double lds[];
    __shared__ double sh[4];
    // workgroups b and b + 8 share an XCD (and its L2): neighbouring tiles, which read the same nodes, go to one XCD
    const unsigned chunk = gridDim.x / 8, lb = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
    if (lb >= (unsigned)nlist) return;
    // the tile's record by its position in the list (one scalar load of 32 B; the list -> header indirection cost every tile a round trip)
    const int* hd = lhdr + 8 * (size_t)lb;
    const int node_off = hd[0], nown = hd[1], nnode = hd[2], elem_off = hd[3], step_off = hd[4], nsteps = hd[5], t = hd[6];
    double* acc = lds;
    int* gl = reinterpret_cast<int*>(lds + 3 * (size_t)max_nown);   // (sums for the owned nodes only: rim nodes are never added to -- 48 instead of 73 KB for the 15 x 15 x 7 tile)
    {   // node list -> LDS, sums zeroed (all loads of a lane in flight together: TILE_MAX_NODES <= 12 * 256)
        int nv[12];
#pragma unroll
        for (int j = 0; j < 12; j++) { const int i = threadIdx.x + 256 * j; nv[j] = (i < nnode) ? ld_list<STREAM>(tnodes + node_off + i) : 0; }
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < nnode) gl[i] = nv[j];
            if (i < nown) { acc[3 * i] = 0.0; acc[3 * i + 1] = 0.0; acc[3 * i + 2] = 0.0; }
        }
    }
    __syncthreads();
    auto pos_of = [&](int s) { return (size_t)elem_off + (s < nsteps ? steps[step_off + s].x : 0) + threadIdx.x; };   // (a tile of nodes without elements has no steps)
    auto raw_of = [&](TileRaw& rw, int s) {
        const bool have = s < nsteps;
        const int2 st = have ? steps[step_off + s] : make_int2(0, 0);
        tile_fetch_raw<AFFINE, STREAM>(rw, have && (int)threadIdx.x < (st.y & 0x7fffffff), (size_t)elem_off + st.x + threadIdx.x, telems,
                               tconn, tflags, Ev, nuv);
    };
    TileRaw r2, r3;            // steps s + 2, s + 3
    TileIdx i0, i1, i2;        // steps s, s + 1, s + 2
    TileData d0, d1;           // steps s, s + 1 (affine kernel)
    raw_of(r2, 0); tile_resolve<AFFINE>(i0, r2, gl);
    raw_of(r2, 1); tile_resolve<AFFINE>(i1, r2, gl);
    raw_of(r2, 2);
    constexpr int NG = KIND == 2 ? 5 : 11;
    if constexpr (AFFINE) tile_fetch_data<NG, STREAM>(d0, i0, pos_of(0), geo, geo_stride, x);
    // the general kernel: coordinates and values of the elements of step s + 1 are asked for at the END of step s -- behind the
    // arithmetic, in front of the hand-over, the LDS adds and the barrier -- so that their way through L2 / the Infinity Cache runs
    // under those instead of in front of the next step's arithmetic (round-4 counters: 30 % of the wave time parked at s_waitcnt with
    // the gathers at the top of the step).  No register is added at the peak: inside the Gauss pairs these 48 doubles are dead.
    // The general kernel does not use the chains of the lane rows: all eight corners are gathered and all eight forces go to LDS
    // (round 6: the 96 DPP moves and selects of the hand-overs cost this kernel, whose bound is the vector issue, more than the
    // eight extra gathers and twelve extra LDS adds -- 0.377 against 0.389 ms per product at the distorted config 3; same values
    // gathered, so the element forces are bit for bit the same; the order of the LDS adds inside a colour differs).
    double Xn[8][3], un[8][3];
    auto gather_general = [&](const TileIdx& in) {
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int d = 0; d < 3; d++) { Xn[a][d] = 0.0; un[a][d] = 0.0; }
        if (in.valid) {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                load_node3(Xn[a], coords, in.v[a]);
                load_node3(un[a], x, in.v[a]);
            }
        }
    };
    if constexpr (!AFFINE) gather_general(i0);
    auto do_step = [&](int s, TileData& dcur, TileData& dnext) {
        const bool bar = steps[step_off + s].y < 0;
        if constexpr (AFFINE) {
            raw_of(r3, s + 3);
            tile_resolve<true>(i2, r2, gl);
            tile_fetch_data<NG, STREAM>(dnext, i1, pos_of(s + 1), geo, geo_stride, x);
        }
        // (lanes without an element run the arithmetic on whatever their registers hold: nothing of it is stored or handed on)
        double fe[8][3], dot;
        if constexpr (KIND == 2) {
            tile_complete<3>(dcur.u, i0.fl);
            const double a[3] = {dcur.geo[0], dcur.geo[1], dcur.geo[2]};
            hex_forces_cart(a, dcur.geo[3], dcur.geo[4], dcur.u, g, fe, dot);
        } else if constexpr (KIND == 1) {
            tile_complete<3>(dcur.u, i0.fl);
            double A[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) A[i][j] = dcur.geo[3 * i + j];
            hex_forces_geom(A, dcur.geo[9], dcur.geo[10], dcur.u, g, fe, dot);
        } else {
            double X[8][3], u[8][3];
#pragma unroll
            for (int a = 0; a < 8; a++)
#pragma unroll
                for (int d = 0; d < 3; d++) { X[a][d] = Xn[a][d]; u[a][d] = un[a][d]; }
            if constexpr (OP == 0) {
                const DCoef D = dmat_coef(i0.valid ? i0.E : 1.0, i0.valid ? i0.nu : 0.25);
                hex_forces<true>(X, u, g, w, D.d12, D.d44, fe, dot);
            } else hex_nodal_stress<OP - 1>(X, u, g, i0.valid ? i0.E : 1.0, i0.valid ? i0.nu : 0.25, fe);
            gather_general(i1);   // step s + 1 (an invalid record beyond the last step: nothing is loaded)
        }
        // +x face forces of a linked lane go to the next lane's -x face nodes (kernels of stored geometry)
        if constexpr (AFFINE) {
            const bool recv = __builtin_amdgcn_update_dpp(0, (int)(i0.fl & 1u), 0x111, 0xf, 0xf, true) != 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const double pv = row_prev(fe[c_tplus[i]][d]);
                    if (recv) fe[c_tminus[i]][d] += pv;
                }
        }
        if (i0.valid) {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                if (AFFINE && (i0.fl & 1u) && (a == 1 || a == 2 || a == 5 || a == 6)) continue;   // handed over
                const int la = tile_loc(i0.c16, a);
                if (la >= nown) continue;     // a rim node: its sum belongs to another tile
#pragma unroll
                for (int d = 0; d < 3; d++)   // one writer per node and colour: the LDS unit's add is as good as a read-add-write
                    __hip_atomic_fetch_add(&acc[3 * la + d], fe[a][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (bar) __syncthreads();
        if constexpr (AFFINE) { i0 = i1; i1 = i2; r2 = r3; }
        else { i0 = i1; tile_resolve<false>(i1, r2, gl); raw_of(r2, s + 3); }   // (registers: one stage less than the affine kernels)
    };
    for (int s = 0; s < nsteps; s++) {   // (letting the two data buffers change roles instead of copying one costs registers: spills)
        do_step(s, d0, d1);
        if constexpr (AFFINE) d0 = d1;
    }
    double dot = 0.0;
    for (int i0n = 0; i0n < nown; i0n += 1024) {   // four owned nodes per lane and round: their loads in flight together
        int kb[4];
        double xv[4][3];
        size_t vv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0n + (int)threadIdx.x + 256 * j;
            vv[j] = (i < nown) ? (size_t)gl[i] : 0;
            kb[j] = (OP == 0 && i < nown && bc_node) ? bc_node[vv[j]] : -1;
            xv[j][0] = xv[j][1] = xv[j][2] = 0.0;
            if (OP == 0 && i < nown) load_node3(xv[j], x, (int)vv[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0n + (int)threadIdx.x + 256 * j;
            if (i >= nown) continue;
            if constexpr (OP != 0) {   // stress_node(i,:) / global_count_node (src/main.F90:277-279; no solid element: 0)
                const double rv = valence ? valence[vv[j]] : 1.0;
#pragma unroll
                for (int d = 0; d < 3; d++) y[(size_t)d * nn + vv[j]] = (valence && rv != 0.0) ? acc[3 * i + d] / rv : acc[3 * i + d];
                continue;
            }
            double sv[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                sv[d] = acc[3 * i + d];
                if (kb[j] >= 0 && bc_dof[kb[j] + d] >= 0) sv[d] += bc_shift[kb[j] + d] * xv[j][d];   // ApplyKBC INSERTs on the diagonal only
                dot += xv[j][d] * sv[d];
            }
            *reinterpret_cast<Node3*>(y + vv[j] * 3) = Node3{sv[0], sv[1], sv[2]};
        }
    }
    if constexpr (OP != 0) return;
    tile_finish(dot, t, sh, partial, rcount, rtotal, red);
}

// a tile's step table, first-step slot records and geometry record by global list position (EbeTilePlan::psteps ...); gp0: position of
// the launch's first tile
struct TileFirst { const int2* psteps; const uint4* pconn0; const unsigned char* pfl0; const double* pgeo; unsigned long long gp0; unsigned nsl; };   // nsl: EbeTilePlan::pslot_steps

// Axis-aligned bricks with x of the tile in LDS.  On a mesh of bricks the plan uses small tiles (15 x 7 x 7 owned nodes: 1,024
// elements = one 256-lane step in each of the four colours, 1,377 nodes), so that x of all the tile's nodes (24 B each) and the sums of its
// owned nodes (24 B each) are 51 KB of LDS and THREE workgroups share a CU: their phases (stage-in, steps, write-out)
// overlap, which one big tile per CU could not offer (see DESIGN.md 4d for the persistent 512-thread kernel this replaced).
// No gathers from global memory inside the steps: corner values come from LDS (SoA: consecutive lanes read consecutive
// words), the slot records and the 40 B of stored geometry stream in one step ahead, the steps of the tile sit in the lanes
// of every wavefront (v_readlane: nothing waits for a scalar load).  The LDS adds of a step are issued BEHIND the barrier
// that closes the previous colour and the next step's corner values are read ahead of it (x is read-only), so the adds run
// under the next step's arithmetic; contributions to rim nodes are not added at all (their sums belong to other tiles).
template <int KIND, bool STREAM = false, bool UNI = false>   // 2: axis-aligned bricks (three workgroups per CU), 1: other parallelepipeds (full adj(J) stored: two); UNI: every tile of the plan has one geometry record (lhdr[7])
__global__ void __launch_bounds__(256, KIND == 2 ? 3 : 2) k_ebe_tile_x(int nlist, const int* __restrict__ lhdr,
                                                       const int2* __restrict__ steps, const int* __restrict__ pnodes,
                                                       const uint4* __restrict__ tconn, const unsigned char* __restrict__ tflags,
                                                       const double* __restrict__ x, const int* __restrict__ bc_node,
                                                       const int* __restrict__ bc_dof, const double* __restrict__ bc_shift,
                                                       const double* __restrict__ geo, size_t geo_stride, double g, int max_nnode,
                                                       int max_nown, double* __restrict__ y, double* __restrict__ partial,
                                                       const int* __restrict__ done, unsigned* __restrict__ rcount, int rtotal,
                                                       double* __restrict__ red, TileFirst tf)
{
    if (done && *done) return;
    static_assert(KIND == 1 || KIND == 2, "stored geometry only");
    constexpr int NG = KIND == 2 ? 5 : 11;
    constexpr int NPT = TILE_X_MAX_NODES / 256, NPO = TILE_X_MAX_OWN / 256;   // nodes / owned nodes per thread
    extern __shared__ double lds[];
    __shared__ double sh[4];
    const unsigned chunk = gridDim.x / 8, lb = (blockIdx.x % 8) * chunk + blockIdx.x / 8;   // (XCD map as in k_ebe_tile)
    if (lb >= (unsigned)nlist) return;
    // the tile's record by its position in the list (one scalar load of 32 B; the list -> header indirection cost every tile a round trip)
    const int* hd = lhdr + 8 * (size_t)lb;
    const int nown = hd[1], nnode = hd[2], elem_off = hd[3], nsteps = hd[5], t = hd[6];
    constexpr int MN = TILE_X_LDS_NODES, MO = TILE_X_MAX_OWN;
    double* xs = lds;                        // xs[d * MN + i]: x of the tile's nodes
    double* acc = lds + 3 * MN;              // acc[d * MO + i]: sums of the owned nodes (the first nown of the list)
    // Everything of the tile that sits at a fixed stride per list position is asked for FIRST, beside the tile's record: the step table
    // (into the lanes of every wavefront: lane s holds step s; at most 64, checked by the host), the first step's slot records, the
    // tile's geometry record, then the node ids.  (Behind the stage-in -- where they stood until the tile's life was measured -- they
    // were a third round trip: issued when the gathers of x had come back, 2.2 us before the first step could start.)
    const size_t gp = (size_t)tf.gp0 + lb;
    const int2 sv = tf.psteps[gp * 64 + (threadIdx.x & 63)];
    const uint4 first_c16 = ld_list<STREAM>(tf.pconn0 + gp * tf.nsl * 256 + threadIdx.x);
    const unsigned first_fl = ld_list<STREAM>(tf.pfl0 + gp * tf.nsl * 256 + threadIdx.x);
    double ug[UNI ? NG : 1], uk[(UNI && KIND == 2) ? CART_NK : 1];
    if constexpr (UNI && KIND == 2) {   // one record for the tile, the same address in every lane: the coefficients of hex_forces_cart_k
#pragma unroll
        for (int j = 0; j < CART_NK; j++) uk[j] = tf.pgeo[gp * TILE_GEO_STRIDE + 16 + j];
    } else if constexpr (UNI) {
#pragma unroll
        for (int j = 0; j < NG; j++) ug[j] = tf.pgeo[gp * TILE_GEO_STRIDE + j];
    }
    int nv[NPT];
    int kb[NPO];   // penalty rows among the owned nodes (ApplyKBC): asked for with the gathers of x, needed at the write-out
    {   // stage in: node list, x -> LDS, sums zeroed (all loads of a lane in flight together)
        double xv[NPT][3];
#pragma unroll
        for (int j = 0; j < NPT; j++)   // (the list at a fixed stride per list position: asked for beside the record, not behind it; 0 beyond nnode)
            nv[j] = ld_list<STREAM>(pnodes + (size_t)lb * TILE_X_MAX_NODES + threadIdx.x + 256 * j);
#pragma unroll
        for (int j = 0; j < NPT; j++) {   // a node's three values as one 16 B and one 8 B load (a third fewer trips through the address unit)
            const int i = threadIdx.x + 256 * j;
            const Node3 v = (i < nnode) ? *reinterpret_cast<const Node3*>(x + (size_t)nv[j] * 3) : Node3{0.0, 0.0, 0.0};
            xv[j][0] = v.a; xv[j][1] = v.b; xv[j][2] = v.c;
            // the node's penalty record with its values, unconditionally (no list: the tile's own record stands in and the result is not
            // looked at; lanes without an owned node are skipped at the write-out).  As conditional loads behind the stage-in these came
            // out of the compiler one behind the other, each waiting for the one before, in front of the barrier: 1.5 us of a 14 us tile.
            if (j < NPO) kb[j] = (bc_node ? bc_node : hd)[bc_node ? nv[j] : 0];
        }
#pragma unroll
        for (int j = 0; j < NPT; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < nnode) {
#pragma unroll
                for (int d = 0; d < 3; d++) xs[d * MN + i] = xv[j][d];
            }
            if (i < nown) {
#pragma unroll
                for (int d = 0; d < 3; d++) acc[d * MO + i] = 0.0;
            }
        }
    }
    struct Slot { uint4 c16; unsigned fl; double geo[NG]; };   // fl bit 1: an element (0: padding or past the step), bit 0: linked to the next lane
    auto fetch = [&](Slot& q, int s) {
        q.fl = 0;
        if (s >= nsteps) return;
        const int first = __builtin_amdgcn_readlane(sv.x, s), cnt = __builtin_amdgcn_readlane(sv.y, s) & 0x7fffffff;
        if ((int)threadIdx.x >= cnt) return;
        const size_t k = (size_t)elem_off + first + threadIdx.x;
        q.c16 = ld_list<STREAM>(tconn + k);
        q.fl = ld_list<STREAM>(tflags + k);
        if constexpr (UNI) {
#pragma unroll
            for (int j = 0; j < NG; j++) q.geo[j] = ug[j];
        } else {
#pragma unroll
            for (int j = 0; j < NG; j++) q.geo[j] = ld_list<STREAM>(geo + (size_t)j * geo_stride + k);
        }
    };
    auto fetch_first = [&](Slot& q) {   // step 0 (lanes beyond its count and tiles without a step hold zeros: no element)
        q.c16 = first_c16;
        q.fl = first_fl;
        if constexpr (UNI) {
#pragma unroll
            for (int j = 0; j < NG; j++) q.geo[j] = ug[j];
        } else {
            const size_t k = (size_t)elem_off + __builtin_amdgcn_readlane(sv.x, 0) + threadIdx.x;
#pragma unroll
            for (int j = 0; j < NG; j++) q.geo[j] = (q.fl & 2u) ? ld_list<STREAM>(geo + (size_t)j * geo_stride + k) : 0.0;
        }
    };
    // corner values of a step's elements: -x face always, +x face where the chain ends (tile_complete brings the others)
    auto read_u = [&](double (&u)[8][3], const Slot& q) {
        const bool valid = (q.fl & 2u) != 0, linked = (q.fl & 1u) != 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int la = tile_loc(q.c16, c_tminus[i]);
#pragma unroll
            for (int d = 0; d < 3; d++) u[c_tminus[i]][d] = valid ? xs[d * MN + la] : 0.0;
        }
        if (valid && !linked) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int la = tile_loc(q.c16, c_tplus[i]);
#pragma unroll
                for (int d = 0; d < 3; d++) u[c_tplus[i]][d] = xs[d * MN + la];
            }
        }
    };
    Slot q0, q1;
    fetch_first(q0);
    __syncthreads();
    double u[8][3];
    read_u(u, q0);
    bool prev_bar = false;   // the previous step closed a colour: its sums must be in before this step adds
    for (int s = 0; s < nsteps; s++) {
        const bool bar = __builtin_amdgcn_readlane(sv.y, s) < 0;
        fetch(q1, s + 1);
        const bool valid = (q0.fl & 2u) != 0, linked = (q0.fl & 1u) != 0;
        // (lanes without an element run the arithmetic on whatever their registers hold: nothing of it is stored or handed on)
        tile_complete<3>(u, q0.fl);
        double fe[8][3], dot;
        if constexpr (KIND == 2 && UNI) {
            hex_forces_cart_k(uk, u, fe, dot);
        } else if constexpr (KIND == 2) {
            const double a[3] = {q0.geo[0], q0.geo[1], q0.geo[2]};
            hex_forces_cart(a, q0.geo[3], q0.geo[4], u, g, fe, dot);
        } else {
            double A[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) A[i][j] = q0.geo[3 * i + j];
            hex_forces_geom(A, q0.geo[9], q0.geo[10], u, g, fe, dot);
        }
        {   // +x face forces of a linked lane go to the next lane's -x face nodes
            const bool recv = __builtin_amdgcn_update_dpp(0, (int)(q0.fl & 1u), 0x111, 0xf, 0xf, true) != 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const double pv = row_prev(fe[c_tplus[i]][d]);
                    if (recv) fe[c_tminus[i]][d] += pv;
                }
        }
        read_u(u, q1);
        if (prev_bar) __syncthreads();
        if (valid) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (linked && (c == 1 || c == 2 || c == 5 || c == 6)) continue;   // handed over
                const int la = tile_loc(q0.c16, c);
                if (la >= nown) continue;     // a rim node: its sum belongs to another tile
#pragma unroll
                for (int d = 0; d < 3; d++)   // one writer per node and colour
                    __hip_atomic_fetch_add(&acc[d * MO + la], fe[c][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        prev_bar = bar;
        q0 = q1;
    }
    __syncthreads();   // every sum is in
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NPO; j++) {
        const int i = threadIdx.x + 256 * j;
        if (i >= nown) continue;
        double sum[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            sum[d] = acc[d * MO + i];
            const double xi = xs[d * MN + i];
            if (bc_node && kb[j] >= 0 && bc_dof[kb[j] + d] >= 0) sum[d] += bc_shift[kb[j] + d] * xi;   // ApplyKBC INSERTs on the diagonal only
            dot += xi * sum[d];
        }
        *reinterpret_cast<Node3*>(y + (size_t)nv[j] * 3) = Node3{sum[0], sum[1], sum[2]};
    }
    tile_finish(dot, t, sh, partial, rcount, rtotal, red);
}

// The same kernel for plans whose tiles all have one geometry record (UNI) and at most four steps: the slot records of ALL the steps sit
// at the fixed stride (EbeTilePlan::pslot_steps == 4) and are asked for with the tile's record -- no load is in flight at the barriers
// between the colours and no step waits for the records of the next one (measured with s_memrealtime stamps: the first step of a tile
// took 3.1 us, the others 1.4-1.8, of which the arithmetic is 0.9).
template <int KIND, bool STREAM>
__global__ void __launch_bounds__(256, KIND == 2 ? 3 : 2) k_ebe_tile_x4(int nlist, const int* __restrict__ lhdr,
                                                       const int2* __restrict__ steps, const int* __restrict__ pnodes,
                                                       const uint4* __restrict__ tconn, const unsigned char* __restrict__ tflags,
                                                       const double* __restrict__ x, const int* __restrict__ bc_node,
                                                       const int* __restrict__ bc_dof, const double* __restrict__ bc_shift,
                                                       const double* __restrict__ geo, size_t geo_stride, double g, int max_nnode,
                                                       int max_nown, double* __restrict__ y, double* __restrict__ partial,
                                                       const int* __restrict__ done, unsigned* __restrict__ rcount, int rtotal,
                                                       double* __restrict__ red, TileFirst tf)
{
    if (done && *done) return;
    static_assert(KIND == 1 || KIND == 2, "stored geometry only");
    constexpr int NG = KIND == 2 ? 5 : 11;
    constexpr bool UNI = true;
    constexpr int NPT = TILE_X_MAX_NODES / 256, NPO = TILE_X_MAX_OWN / 256;   // nodes / owned nodes per thread
    extern __shared__ double lds[];
    __shared__ double sh[4];
    const unsigned chunk = gridDim.x / 8, lb = (blockIdx.x % 8) * chunk + blockIdx.x / 8;   // (XCD map as in k_ebe_tile)
    if (lb >= (unsigned)nlist) return;
    // the tile's record by its position in the list (one scalar load of 32 B; the list -> header indirection cost every tile a round trip)
    const int* hd = lhdr + 8 * (size_t)lb;
    const int nown = hd[1], nnode = hd[2], nsteps = hd[5], t = hd[6];
    constexpr int MN = TILE_X_LDS_NODES, MO = TILE_X_MAX_OWN;
    double* xs = lds;                        // xs[d * MN + i]: x of the tile's nodes
    double* acc = lds + 3 * MN;              // acc[d * MO + i]: sums of the owned nodes (the first nown of the list)
    // Everything of the tile that sits at a fixed stride per list position is asked for FIRST, beside the tile's record: the step table
    // (into the lanes of every wavefront: lane s holds step s; at most 64, checked by the host), the first step's slot records, the
    // tile's geometry record, then the node ids.  (Behind the stage-in -- where they stood until the tile's life was measured -- they
    // were a third round trip: issued when the gathers of x had come back, 2.2 us before the first step could start.)
    const size_t gp = (size_t)tf.gp0 + lb;
    const int2 sv = tf.psteps[gp * 64 + (threadIdx.x & 63)];
    uint4 sc16[4];
    unsigned sfl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        sc16[k] = ld_list<STREAM>(tf.pconn0 + (gp * 4 + k) * 256 + threadIdx.x);
        sfl[k] = ld_list<STREAM>(tf.pfl0 + (gp * 4 + k) * 256 + threadIdx.x);
    }
    double ug[UNI ? NG : 1], uk[(UNI && KIND == 2) ? CART_NK : 1];
    if constexpr (UNI && KIND == 2) {   // one record for the tile, the same address in every lane: the coefficients of hex_forces_cart_k
#pragma unroll
        for (int j = 0; j < CART_NK; j++) uk[j] = tf.pgeo[gp * TILE_GEO_STRIDE + 16 + j];
    } else if constexpr (UNI) {
#pragma unroll
        for (int j = 0; j < NG; j++) ug[j] = tf.pgeo[gp * TILE_GEO_STRIDE + j];
    }
    int nv[NPT];
    int kb[NPO];   // penalty rows among the owned nodes (ApplyKBC): asked for with the gathers of x, needed at the write-out
    {   // stage in: node list, x -> LDS, sums zeroed (all loads of a lane in flight together)
        double xv[NPT][3];
#pragma unroll
        for (int j = 0; j < NPT; j++)   // (the list at a fixed stride per list position: asked for beside the record, not behind it; 0 beyond nnode)
            nv[j] = ld_list<STREAM>(pnodes + (size_t)lb * TILE_X_MAX_NODES + threadIdx.x + 256 * j);
#pragma unroll
        for (int j = 0; j < NPT; j++) {   // a node's three values as one 16 B and one 8 B load (a third fewer trips through the address unit)
            const int i = threadIdx.x + 256 * j;
            const Node3 v = (i < nnode) ? *reinterpret_cast<const Node3*>(x + (size_t)nv[j] * 3) : Node3{0.0, 0.0, 0.0};
            xv[j][0] = v.a; xv[j][1] = v.b; xv[j][2] = v.c;
            // the node's penalty record with its values, unconditionally (no list: the tile's own record stands in and the result is not
            // looked at; lanes without an owned node are skipped at the write-out).  As conditional loads behind the stage-in these came
            // out of the compiler one behind the other, each waiting for the one before, in front of the barrier: 1.5 us of a 14 us tile.
            if (j < NPO) kb[j] = (bc_node ? bc_node : hd)[bc_node ? nv[j] : 0];
        }
#pragma unroll
        for (int j = 0; j < NPT; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < nnode) {
#pragma unroll
                for (int d = 0; d < 3; d++) xs[d * MN + i] = xv[j][d];
            }
            if (i < nown) {
#pragma unroll
                for (int d = 0; d < 3; d++) acc[d * MO + i] = 0.0;
            }
        }
    }
    // corner values of a step's elements: -x face always, +x face where the chain ends (tile_complete brings the others)
    auto read_u = [&](double (&u)[8][3], const uint4& c16, const unsigned fl) {
        const bool valid = (fl & 2u) != 0, linked = (fl & 1u) != 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int la = tile_loc(c16, c_tminus[i]);
#pragma unroll
            for (int d = 0; d < 3; d++) u[c_tminus[i]][d] = valid ? xs[d * MN + la] : 0.0;
        }
        if (valid && !linked) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int la = tile_loc(c16, c_tplus[i]);
#pragma unroll
                for (int d = 0; d < 3; d++) u[c_tplus[i]][d] = xs[d * MN + la];
            }
        }
    };
    __syncthreads();
    double u[8][3];
    read_u(u, sc16[0], sfl[0]);
    bool prev_bar = false;   // the previous step closed a colour: its sums must be in before this step adds
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if (s >= nsteps) break;
        const bool bar = __builtin_amdgcn_readlane(sv.y, s) < 0;
        const uint4 c16 = sc16[s];
        const unsigned fl = sfl[s];
        const bool valid = (fl & 2u) != 0, linked = (fl & 1u) != 0;
        // (lanes without an element run the arithmetic on whatever their registers hold: nothing of it is stored or handed on)
        tile_complete<3>(u, fl);
        double fe[8][3], dot;
        if constexpr (KIND == 2) {
            hex_forces_cart_k(uk, u, fe, dot);
        } else {
            double A[3][3];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) A[i][j] = ug[3 * i + j];
            hex_forces_geom(A, ug[9], ug[10], u, g, fe, dot);
        }
        {   // +x face forces of a linked lane go to the next lane's -x face nodes
            const bool recv = __builtin_amdgcn_update_dpp(0, (int)(fl & 1u), 0x111, 0xf, 0xf, true) != 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const double pv = row_prev(fe[c_tplus[i]][d]);
                    if (recv) fe[c_tminus[i]][d] += pv;
                }
        }
        if (s < 3) read_u(u, sc16[s < 3 ? s + 1 : 3], sfl[s < 3 ? s + 1 : 3]);
        if (prev_bar) __syncthreads();
        if (valid) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                if (linked && (c == 1 || c == 2 || c == 5 || c == 6)) continue;   // handed over
                const int la = tile_loc(c16, c);
                if (la >= nown) continue;     // a rim node: its sum belongs to another tile
#pragma unroll
                for (int d = 0; d < 3; d++)   // one writer per node and colour
                    __hip_atomic_fetch_add(&acc[d * MO + la], fe[c][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        prev_bar = bar;
    }
    __syncthreads();   // every sum is in
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NPO; j++) {
        const int i = threadIdx.x + 256 * j;
        if (i >= nown) continue;
        double sum[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            sum[d] = acc[d * MO + i];
            const double xi = xs[d * MN + i];
            if (bc_node && kb[j] >= 0 && bc_dof[kb[j] + d] >= 0) sum[d] += bc_shift[kb[j] + d] * xi;   // ApplyKBC INSERTs on the diagonal only
            dot += xi * sum[d];
        }
        *reinterpret_cast<Node3*>(y + (size_t)nv[j] * 3) = Node3{sum[0], sum[1], sum[2]};
    }
    tile_finish(dot, t, sh, partial, rcount, rtotal, red);
}

// Nodal stress of a mesh of parallelepipeds (KIND 2: axis-aligned bricks, 1: the others) in ONE pass over the small tiles (u of the tile and six sums per owned node in
// LDS, 69 KB: two workgroups per CU).  On a parallelepiped the strains are (bi)linear along the element, i.e. inside the
// trilinear space the N2^-1 extrapolation of RecoverNodalStress (src/m_global.F90:506-509) reproduces exactly: the value the
// extrapolation gives at a node IS the strain of the trilinear displacement at that node, and along an element edge that
// displacement is linear -- so the nodal strain is made of the differences of u along the three edges that meet at the node,
// mapped to x by the stored adj(J) / det (axis-aligned bricks: divided by the edge lengths, 1 / h_i = a_i mw / (16 w mu) from the
// stored diagonal and the Lame factor): no coordinate is read.  sigma = D eps (DMat, src/m_local.F90:204-228; Voigt order xx yy zz xy yz zx of BMat :160-168), summed
// per node over the elements (src/main.F90:258-268) in LDS -- colour by colour, +x face values handed along the 16-lane rows
// as in the product, so every sum has one writer and a fixed order -- and divided by the valence on the way out (:277-279).
// UNI: every tile of the plan has one geometry record (lhdr[7], k_tile_geo_uniform: a brick of one spacing and one material) -- the
// record and the material of its element are read once per tile instead of 48 B per lane slot plus a gather of E and nu behind
// the element id (round 3: 0.63 GB of traffic for 0.43 GB of nodal data, 0.19 GB of it these records)
// ALL4 (with UNI; plans of at most four steps per tile, EbeTilePlan::pslot_steps == 4): the slot records of every step are asked for with
// the tile's record and the steps are unrolled, as in k_ebe_tile_x4
template <int KIND, bool UNI = false, bool ALL4 = false>
__global__ void __launch_bounds__(256, 2) k_tile_stress_cart(int nlist, const int* __restrict__ lhdr, const int2* __restrict__ steps,
                                                          const int* __restrict__ tnodes, const int* __restrict__ telems,
                                                          const uint4* __restrict__ tconn, const unsigned char* __restrict__ tflags,
                                                          const double* __restrict__ U, const double* __restrict__ geo, size_t geo_stride,
                                                          const double* __restrict__ Ev, const double* __restrict__ nuv, double w,
                                                          double* __restrict__ S, const double* __restrict__ valence, size_t nn,
                                                          const int* __restrict__ pnodes, TileFirst tf)
{
    constexpr int NPT = TILE_X_MAX_NODES / 256, NPO = TILE_X_MAX_OWN / 256;
    constexpr int MN = TILE_X_LDS_NODES, MO = TILE_X_MAX_OWN;
    extern __shared__ double lds[];
    const unsigned chunk = gridDim.x / 8, lb = (blockIdx.x % 8) * chunk + blockIdx.x / 8;   // (XCD map as in k_ebe_tile)
    if (lb >= (unsigned)nlist) return;
    const int* hd = lhdr + 8 * (size_t)lb;
    const int nown = hd[1], nnode = hd[2], elem_off = hd[3], nsteps = hd[5];
    // (the tile's lists at fixed strides per list position, asked for with its record: see k_ebe_tile_x; with one record per tile the
    // geometry, E and nu come from there too and no step loads anything but its slot records)
    const size_t gp = (size_t)tf.gp0 + lb;
    const int2 sv = tf.psteps[gp * 64 + (threadIdx.x & 63)];
    uint4 sc16[ALL4 ? 4 : 1]; unsigned sfl[ALL4 ? 4 : 1];
    if constexpr (ALL4) {
#pragma unroll
        for (int k = 0; k < 4; k++) { sc16[k] = tf.pconn0[(gp * 4 + k) * 256 + threadIdx.x]; sfl[k] = tf.pfl0[(gp * 4 + k) * 256 + threadIdx.x]; }
    } else {
        sc16[0] = tf.pconn0[gp * tf.nsl * 256 + threadIdx.x]; sfl[0] = tf.pfl0[gp * tf.nsl * 256 + threadIdx.x];
    }
    constexpr int NA = KIND == 2 ? 3 : 9;   // stored adj(J): the diagonal / row by row
    double ua[UNI ? NA : 1], umw = 0.0, uE = 1.0, unu = 0.25;
    if constexpr (UNI) {
#pragma unroll
        for (int j = 0; j < NA; j++) ua[j] = tf.pgeo[gp * TILE_GEO_STRIDE + j];
        umw = tf.pgeo[gp * TILE_GEO_STRIDE + (KIND == 2 ? 4 : 10)]; uE = tf.pgeo[gp * TILE_GEO_STRIDE + 14]; unu = tf.pgeo[gp * TILE_GEO_STRIDE + 15];
    }
    double* xs = lds;              // xs[d * MN + i]: u of the tile's nodes
    double* acc = lds + 3 * MN;    // acc[c * MO + i]: six stress sums of the owned nodes
    int nv[NPT];
    {
        double xv[NPT][3];
#pragma unroll
        for (int j = 0; j < NPT; j++) nv[j] = pnodes[(size_t)lb * TILE_X_MAX_NODES + threadIdx.x + 256 * j];
#pragma unroll
        for (int j = 0; j < NPT; j++) {
            const int i = threadIdx.x + 256 * j;
            const Node3 v = (i < nnode) ? *reinterpret_cast<const Node3*>(U + (size_t)nv[j] * 3) : Node3{0.0, 0.0, 0.0};
            xv[j][0] = v.a; xv[j][1] = v.b; xv[j][2] = v.c;
        }
#pragma unroll
        for (int j = 0; j < NPT; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < nnode) {
#pragma unroll
                for (int d = 0; d < 3; d++) xs[d * MN + i] = xv[j][d];
            }
            if (i < nown) {
#pragma unroll
                for (int c = 0; c < 6; c++) acc[c * MO + i] = 0.0;
            }
        }
    }
    struct Slot { uint4 c16; unsigned fl; double a[NA], mw, E, nu; };
    auto fetch = [&](Slot& q, int s) {
        q.fl = 0;
        if (s >= nsteps) return;
        const int first = __builtin_amdgcn_readlane(sv.x, s), cnt = __builtin_amdgcn_readlane(sv.y, s) & 0x7fffffff;
        if ((int)threadIdx.x >= cnt) return;
        const size_t k = (size_t)elem_off + first + threadIdx.x;
        q.c16 = tconn[k];
        q.fl = tflags[k];
        if constexpr (UNI) {
#pragma unroll
            for (int j = 0; j < NA; j++) q.a[j] = ua[j];
            q.mw = umw; q.E = uE; q.nu = unu;
        } else {
            const int e = telems[k];
#pragma unroll
            for (int j = 0; j < NA; j++) q.a[j] = geo[(size_t)j * geo_stride + k];
            q.mw = geo[(size_t)(KIND == 2 ? 4 : 10) * geo_stride + k];
            q.E = (e >= 0) ? Ev[e] : 1.0; q.nu = (e >= 0) ? nuv[e] : 0.25;
        }
    };
    auto read_u = [&](double (&u)[8][3], const Slot& q) {
        const bool valid = (q.fl & 2u) != 0, linked = (q.fl & 1u) != 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int la = tile_loc(q.c16, c_tminus[i]);
#pragma unroll
            for (int d = 0; d < 3; d++) u[c_tminus[i]][d] = valid ? xs[d * MN + la] : 0.0;
        }
        if (valid && !linked) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int la = tile_loc(q.c16, c_tplus[i]);
#pragma unroll
                for (int d = 0; d < 3; d++) u[c_tplus[i]][d] = xs[d * MN + la];
            }
        }
    };
    Slot q0, q1;
    if constexpr (UNI) {   // the first step's slot records came with the tile's record
        q0.c16 = sc16[0]; q0.fl = sfl[0];
#pragma unroll
        for (int j = 0; j < NA; j++) q0.a[j] = ua[j];
        q0.mw = umw; q0.E = uE; q0.nu = unu;
    } else {
        fetch(q0, 0);
    }
    __syncthreads();
    double u[8][3];
    read_u(u, q0);
    bool prev_bar = false;
    // (ALL4: one outer pass, the four steps unrolled, their slot records in registers -- past a step's lane count and past the last step
    // they are zero, k_tile_first; otherwise one step per outer pass)
    for (int s0 = 0; s0 < (ALL4 ? 1 : nsteps); s0++) {
#pragma unroll
    for (int k = 0; k < (ALL4 ? 4 : 1); k++) {
        const int s = s0 + k;
        if (s >= nsteps) break;
        const bool bar = __builtin_amdgcn_readlane(sv.y, s) < 0;
        if constexpr (ALL4) {
            q1.fl = 0;
            if (k < 3) { q1.c16 = sc16[k < 3 ? k + 1 : 0]; q1.fl = sfl[k < 3 ? k + 1 : 0]; }
#pragma unroll
            for (int j = 0; j < NA; j++) q1.a[j] = ua[j];
            q1.mw = umw; q1.E = uE; q1.nu = unu;
        } else {
            fetch(q1, s + 1);
        }
        const bool valid = (q0.fl & 2u) != 0, linked = (q0.fl & 1u) != 0;
        tile_complete<3>(u, q0.fl);
        // differences of u along the four xi-, eta-, zeta-edges (node pairs in the order of their lower node: bottom CCW, top CCW)
        constexpr int ex[4][2] = {{0, 1}, {3, 2}, {4, 5}, {7, 6}}, ey[4][2] = {{0, 3}, {1, 2}, {4, 7}, {5, 6}}, ez[4][2] = {{0, 4}, {1, 5}, {2, 6}, {3, 7}};
        const DCoef D = dmat_coef(q0.E, q0.nu);
        const double kinv = q0.mw / (16.0 * w * D.d44);          // 1 / (2 det J)
        double sn[8][6];
        {
            double dx[4][3], dy[4][3], dz[4][3];   // differences of u_d along the edges (= 2 du_d / d xi_j at the edge's nodes)
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    dx[k][d] = u[ex[k][1]][d] - u[ex[k][0]][d];
                    dy[k][d] = u[ey[k][1]][d] - u[ey[k][0]][d];
                    dz[k][d] = u[ez[k][1]][d] - u[ez[k][0]][d];
                }
            // node a sits on x-edge kx[a], y-edge ky[a], z-edge kz[a]
            constexpr int kx[8] = {0, 0, 1, 1, 2, 2, 3, 3}, ky[8] = {0, 1, 1, 0, 2, 3, 3, 2}, kz[8] = {0, 1, 2, 3, 0, 1, 2, 3};
            const double two_mu = 2.0 * D.d44;
#pragma unroll
            for (int a = 0; a < 8; a++) {
                double H[3][3];   // H[l][i] = du_l / dx_i = sum_j adj(J)[i][j] du_l/dxi_j / det
#pragma unroll
                for (int l = 0; l < 3; l++) {
                    const double g0 = dx[kx[a]][l], g1 = dy[ky[a]][l], g2 = dz[kz[a]][l];
                    if constexpr (KIND == 2) {
                        H[l][0] = q0.a[0] * kinv * g0; H[l][1] = q0.a[1] * kinv * g1; H[l][2] = q0.a[2] * kinv * g2;
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; i++) H[l][i] = kinv * (q0.a[3 * i] * g0 + q0.a[3 * i + 1] * g1 + q0.a[3 * i + 2] * g2);
                    }
                }
                const double exx = H[0][0], eyy = H[1][1], ezz = H[2][2];
                const double tr = D.d12 * ((exx + eyy) + ezz);
                sn[a][0] = fma(two_mu, exx, tr); sn[a][1] = fma(two_mu, eyy, tr); sn[a][2] = fma(two_mu, ezz, tr);
                sn[a][3] = D.d44 * (H[0][1] + H[1][0]);
                sn[a][4] = D.d44 * (H[1][2] + H[2][1]);
                sn[a][5] = D.d44 * (H[2][0] + H[0][2]);
            }
        }
        {   // +x face values of a linked lane go to the next lane's -x face nodes
            const bool recv = __builtin_amdgcn_update_dpp(0, (int)(q0.fl & 1u), 0x111, 0xf, 0xf, true) != 0;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const double pv = row_prev(sn[c_tplus[i]][c]);
                    if (recv) sn[c_tminus[i]][c] += pv;
                }
        }
        read_u(u, q1);
        if (prev_bar) __syncthreads();
        if (valid) {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                if (linked && (a == 1 || a == 2 || a == 5 || a == 6)) continue;   // handed over
                const int la = tile_loc(q0.c16, a);
                if (la >= nown) continue;     // a rim node: its sums belong to another tile
#pragma unroll
                for (int c = 0; c < 6; c++)
                    __hip_atomic_fetch_add(&acc[c * MO + la], sn[a][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        prev_bar = bar;
        q0 = q1;
    }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPO; j++) {
        const int i = threadIdx.x + 256 * j;
        if (i >= nown) continue;
        const double rv = valence ? valence[nv[j]] : 1.0;   // stress_node(i,:) / global_count_node (no solid element: 0)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const double sum = acc[c * MO + i];
            S[(size_t)c * nn + nv[j]] = (valence && rv != 0.0) ? sum / rv : sum;
        }
    }
}

// The tile product of a quad4 mesh (plane strain; config 2): the same plan in two dimensions -- 31 x 31 owned nodes per
// tile, chains of 16 elements along x (two per element row, so the greedy pass needs four colours), LDS image 20 B per node
// -- and the bilinear arithmetic of quad4_ebe.h.  At a million dofs the CG iteration is a handful of 10 microsecond
// kernels: one kernel for the product instead of element forces + gather + a reduction of their partials.
__global__ void __launch_bounds__(256, 3) k_ebe_tile_q4(int nlist, const int* __restrict__ lhdr,
                                                     const int2* __restrict__ steps, const int* __restrict__ tnodes,
                                                     const int* __restrict__ telems, const uint4* __restrict__ tconn,
                                                     const unsigned char* __restrict__ tflags,
                                                     const double* __restrict__ coords, const double* __restrict__ Ev,
                                                     const double* __restrict__ nuv, const double* __restrict__ x,
                                                     const int* __restrict__ bc_node, const int* __restrict__ bc_dof,
                                                     const double* __restrict__ bc_shift, double g, double w, int max_nnode,
                                                     double* __restrict__ y, double* __restrict__ partial,
                                                     const int* __restrict__ done, unsigned* __restrict__ rcount, int rtotal,
                                                     double* __restrict__ red)
{
    if (done && *done) return;
    extern __shared__ double lds[];
    __shared__ double sh[4];
    const unsigned chunk = gridDim.x / 8, lb = (blockIdx.x % 8) * chunk + blockIdx.x / 8;   // XCD-wise, as k_ebe_tile
    if (lb >= (unsigned)nlist) return;
    // the tile's record by its position in the list (one scalar load of 32 B; the list -> header indirection cost every tile a round trip)
    const int* hd = lhdr + 8 * (size_t)lb;
    const int node_off = hd[0], nown = hd[1], nnode = hd[2], elem_off = hd[3], step_off = hd[4], nsteps = hd[5], t = hd[6];
    double* acc = lds;
    int* gl = reinterpret_cast<int*>(lds + 2 * (size_t)max_nnode);
    {
        int nv[12];
#pragma unroll
        for (int j = 0; j < 12; j++) { const int i = threadIdx.x + 256 * j; nv[j] = (i < nnode) ? tnodes[node_off + i] : 0; }
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < nnode) { gl[i] = nv[j]; acc[2 * i] = 0.0; acc[2 * i + 1] = 0.0; }
        }
    }
    __syncthreads();
    // a lane's element one step ahead: local numbers, flag, E and nu, and -- through the node list in LDS -- the gathers of the
    // coordinates and of x for its -x edge (a linked lane takes the +x edge from the next lane, a chain's last lane loads it)
    struct El { uint4 c16; double E, nu, X[4][2], u[4][2]; unsigned fl; };
    auto fetch = [&](El& el, int s) {
        el.fl = 0;
        if (s >= nsteps) return;
        const int2 st = steps[step_off + s];
        if ((int)threadIdx.x >= (st.y & 0x7fffffff)) return;
        const size_t k = (size_t)elem_off + st.x + threadIdx.x;
        const int e = telems[k];
        if (e < 0) return;
        el.c16 = tconn[k];
        el.fl = tflags[k];
        el.E = Ev[e]; el.nu = nuv[e];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            if ((el.fl & 1u) && (a == 1 || a == 2)) continue;
            const size_t v = (size_t)gl[tile_loc(el.c16, a)];
#pragma unroll
            for (int d = 0; d < 2; d++) { el.X[a][d] = coords[v * 2 + d]; el.u[a][d] = x[v * 2 + d]; }
        }
    };
    // two steps ahead: every tile of a million-dof mesh is resident at once (529 workgroups; three waves per SIMD at 160
    // registers -- a third stage would not fit), so the kernel lasts as long as one tile's chain of latencies
    El e0, e1, e2;
    fetch(e0, 0); fetch(e1, 1);
    for (int s = 0; s < nsteps; s++) {
        const bool bar = steps[step_off + s].y < 0;
        fetch(e2, s + 2);
        const bool valid = (e0.fl & 2u) != 0, linked = (e0.fl & 1u) != 0;
        {   // the +x edge (nodes 1, 2) of a linked lane: the next lane's -x edge (nodes 0, 3)
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const double x1 = row_next(e0.X[0][d]), x2 = row_next(e0.X[3][d]), u1 = row_next(e0.u[0][d]), u2 = row_next(e0.u[3][d]);
                if (linked) { e0.X[1][d] = x1; e0.X[2][d] = x2; e0.u[1][d] = u1; e0.u[2][d] = u2; }
            }
        }
        double fe[4][2], dot;
        const DCoef D = dmat_coef(valid ? e0.E : 1.0, valid ? e0.nu : 0.25);
        quad4_forces(e0.X, e0.u, g, w, D.d12, D.d44, fe, dot);
        {   // +x edge forces of a linked lane go to the next lane's -x edge nodes
            const bool recv = __builtin_amdgcn_update_dpp(0, (int)(e0.fl & 1u), 0x111, 0xf, 0xf, true) != 0;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const double p1 = row_prev(fe[1][d]), p2 = row_prev(fe[2][d]);
                if (recv) { fe[0][d] += p1; fe[3][d] += p2; }
            }
        }
        if (valid) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (linked && (a == 1 || a == 2)) continue;   // handed over
                const int la = tile_loc(e0.c16, a);
                if (la >= nown) continue;     // a rim node: its sum belongs to another tile
#pragma unroll
                for (int d = 0; d < 2; d++)   // one writer per node and colour
                    __hip_atomic_fetch_add(&acc[2 * la + d], fe[a][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (bar) __syncthreads();
        e0 = e1; e1 = e2;
    }
    double dot = 0.0;
    for (int i0n = 0; i0n < nown; i0n += 1024) {
        int kb[4];
        double xv[4][2];
        size_t vv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0n + (int)threadIdx.x + 256 * j;
            vv[j] = (i < nown) ? (size_t)gl[i] : 0;
            kb[j] = (i < nown && bc_node) ? bc_node[vv[j]] : -1;
#pragma unroll
            for (int d = 0; d < 2; d++) xv[j][d] = (i < nown) ? x[vv[j] * 2 + d] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int i = i0n + (int)threadIdx.x + 256 * j;
            if (i >= nown) continue;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                double sv = acc[2 * i + d];
                if (kb[j] >= 0 && bc_dof[kb[j] + d] >= 0) sv += bc_shift[kb[j] + d] * xv[j][d];   // ApplyKBC INSERTs on the diagonal only
                y[vv[j] * 2 + d] = sv;
                dot += xv[j][d] * sv;
            }
        }
    }
    tile_finish(dot, t, sh, partial, rcount, rtotal, red);
}

// ---- the steps of a quad4 tile whose x and coordinates sit in LDS (k_ebe_tile_q4x and the resident CG below) ----------------------
constexpr int TILE_Q4X_MAX_NODES = 1280, TILE_Q4X_MAX_OWN = 1024;   // 5 / 4 slots per thread
constexpr int TILE_Q4X_LDS_NODES = 1104, TILE_Q4X_LDS_OWN = 976;    // the LDS image (33 x 33 nodes, 31 x 31 owned): 50 KB, three workgroups per CU
struct Q4Slot { uint4 c16; unsigned fl; double lam, mu; };   // fl bit 1: an element (0: padding or past the step), bit 0: linked to the next lane
// the slot record of step s (s >= nsteps, or a lane past the step's count: an empty record)
__device__ __forceinline__ void q4x_fetch(Q4Slot& q, int s, int nsteps, const int2& sv, int elem_off, const uint4* __restrict__ tconn,
                                          const unsigned char* __restrict__ tflags, const double* __restrict__ geo, size_t geo_stride)
{
    q.fl = 0; q.lam = 1.0; q.mu = 1.0;
    if (s >= nsteps) return;
    const int first = __builtin_amdgcn_readlane(sv.x, s), cnt = __builtin_amdgcn_readlane(sv.y, s) & 0x7fffffff;
    if ((int)threadIdx.x >= cnt) return;
    const size_t k = (size_t)elem_off + first + threadIdx.x;
    q.c16 = tconn[k];
    q.fl = tflags[k];
    q.lam = geo[k]; q.mu = geo[geo_stride + k];
}
__device__ __forceinline__ void q4x_read_corners(double (&X)[4][2], double (&u)[4][2], const Q4Slot& q, const double* xs, const double* Xs)
{
    constexpr int MN = TILE_Q4X_LDS_NODES;
    const bool valid = (q.fl & 2u) != 0, linked = (q.fl & 1u) != 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        if (a == 1 || a == 2) continue;
        const int la = tile_loc(q.c16, a);
#pragma unroll
        for (int d = 0; d < 2; d++) { X[a][d] = valid ? Xs[d * MN + la] : 0.0; u[a][d] = valid ? xs[d * MN + la] : 0.0; }
    }
    if (valid && !linked) {
#pragma unroll
        for (int a = 1; a < 3; a++) {
            const int la = tile_loc(q.c16, a);
#pragma unroll
            for (int d = 0; d < 2; d++) { X[a][d] = Xs[d * MN + la]; u[a][d] = xs[d * MN + la]; }
        }
    }
}
// Every step of the tile: element forces from the corner values in LDS, the hand-over along the lane rows, the adds into the sums of
// the owned nodes (acc, zeroed by the caller; a barrier between the colours and one at the end).  q0: the slot record of step 0.
// KEEP (the resident CG): the records of the first KEEP steps come from `kept` (registers of the caller) instead of the lists.
template <int KEEP = 0>
__device__ __forceinline__ void q4x_steps(Q4Slot q0, int nsteps, const int2& sv, int elem_off, int nown, const uint4* __restrict__ tconn,
                                          const unsigned char* __restrict__ tflags, const double* __restrict__ geo, size_t geo_stride,
                                          const double* xs, const double* Xs, double* acc, double g, double w, const Q4Slot* kept = nullptr)
{
    constexpr int MO = TILE_Q4X_LDS_OWN;
    Q4Slot q1;
    double X[4][2], u[4][2];
    q4x_read_corners(X, u, q0, xs, Xs);
    bool prev_bar = false;
    auto one_step = [&](int s, const Q4Slot* next) {   // next: the record of step s + 1 where the caller keeps it
        const bool bar = __builtin_amdgcn_readlane(sv.y, s) < 0;
        if (next) { q1 = *next; if (s + 1 >= nsteps) q1.fl = 0; }
        else q4x_fetch(q1, s + 1, nsteps, sv, elem_off, tconn, tflags, geo, geo_stride);
        const bool valid = (q0.fl & 2u) != 0, linked = (q0.fl & 1u) != 0;
#pragma unroll
        for (int d = 0; d < 2; d++) {   // the +x edge (nodes 1, 2) of a linked lane: the next lane's -x edge (nodes 0, 3)
            const double x1 = row_next(X[0][d]), x2 = row_next(X[3][d]), u1 = row_next(u[0][d]), u2 = row_next(u[3][d]);
            if (linked) { X[1][d] = x1; X[2][d] = x2; u[1][d] = u1; u[2][d] = u2; }
        }
        double fe[4][2], dote;
        quad4_forces(X, u, g, w, q0.lam, q0.mu, fe, dote);
        {
            const bool recv = __builtin_amdgcn_update_dpp(0, (int)(q0.fl & 1u), 0x111, 0xf, 0xf, true) != 0;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const double p1 = row_prev(fe[1][d]), p2 = row_prev(fe[2][d]);
                if (recv) { fe[0][d] += p1; fe[3][d] += p2; }
            }
        }
        q4x_read_corners(X, u, q1, xs, Xs);
        if (prev_bar) __syncthreads();
        if (valid) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (linked && (a == 1 || a == 2)) continue;   // handed over
                const int la = tile_loc(q0.c16, a);
                if (la >= nown) continue;     // a rim node: its sum belongs to another tile
#pragma unroll
                for (int d = 0; d < 2; d++)   // one writer per node and colour
                    __hip_atomic_fetch_add(&acc[d * MO + la], fe[a][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        prev_bar = bar;
        q0 = q1;
    };
    if constexpr (KEEP > 0) {
#pragma unroll
        for (int s = 0; s < KEEP; s++)
            if (s < nsteps) one_step(s, s + 1 < KEEP ? kept + s + 1 : nullptr);
        for (int s = KEEP; s < nsteps; s++) one_step(s, nullptr);
    } else {
        for (int s = 0; s < nsteps; s++) one_step(s, nullptr);
    }
    __syncthreads();
}

// The same product with x AND the coordinates of the tile's nodes staged in LDS (the quad4 counterpart of k_ebe_tile_x): at a
// million dofs every tile of the mesh is resident at once and the kernel lasts as long as one tile's chain of latencies, which
// the gathering kernel above runs through once per step (slot -> node ids in LDS -> coordinates and x from global memory).  Here
// the chain is: tile record -> node ids -> x, X (one wave of loads) and from then on LDS only; the slot records (local numbers,
// flags, the element's two D factors: stored at set-up, no E / nu gather behind the element id) stream in one step ahead.
// LDS: x and X of the tile's nodes (32 B each) + the sums of its owned nodes (16 B each): 50 KB for the 33 x 33-node tile.
template <bool FUSED>   // FUSED: the kernel opens the CG iteration itself (convergence test, p = M^-1 r + b p in the stage-in, x += a p): cg_device.h
__global__ void __launch_bounds__(256, 3) k_ebe_tile_q4x(int nlist, const int* __restrict__ lhdr, const int2* __restrict__ steps,
                                                      const int* __restrict__ pnodes, const uint4* __restrict__ tconn,
                                                      const unsigned char* __restrict__ tflags, const double* __restrict__ geo,
                                                      size_t geo_stride, const double* __restrict__ coords, const double* __restrict__ x,
                                                      const int* __restrict__ bc_node, const int* __restrict__ bc_dof,
                                                      const double* __restrict__ bc_shift, double g, double w,
                                                      double* __restrict__ y, double* __restrict__ partial, const int* __restrict__ done,
                                                      unsigned* __restrict__ rcount, int rtotal, double* __restrict__ red, CgFuseArgs cg,
                                                      TileFirst tf)
{
    constexpr int NPT = TILE_Q4X_MAX_NODES / 256, NPO = TILE_Q4X_MAX_OWN / 256, MN = TILE_Q4X_LDS_NODES, MO = TILE_Q4X_LDS_OWN;
    extern __shared__ double lds[];
    __shared__ double sh[4];
    const unsigned chunk = gridDim.x / 8, lb = (blockIdx.x % 8) * chunk + blockIdx.x / 8;   // XCD-wise, as k_ebe_tile
    if constexpr (FUSED) {
        // (as k_cg_update_p: not on st->done, which this kernel's own verdict sets while other workgroups are still starting and
        // every workgroup owes the last x += a p of its nodes; `stop` is written by the x / r update that follows)
        if (cg.open.st->stop) return;
    } else {
        if (done && *done) return;
    }
    if (lb >= (unsigned)nlist) return;   // (workgroup 0, which records the verdict of a fused opening, holds tile 0 of the list)
    const int* hd = lhdr + 8 * (size_t)lb;
    const int nown = hd[1], nnode = hd[2], elem_off = hd[3], nsteps = hd[5], t = hd[6];
    double* xs = lds;               // xs[d * MN + i]: x of the tile's nodes
    double* Xs = lds + 2 * MN;      // Xs[d * MN + i]: their coordinates
    double* acc = lds + 4 * MN;     // acc[d * MO + i]: sums of the owned nodes
    // (step table, first step's slot records and D factors at fixed strides per list position, asked for first: see k_ebe_tile_x)
    const size_t gp = (size_t)tf.gp0 + lb;
    const int2 sv = tf.psteps[gp * 64 + (threadIdx.x & 63)];
    Q4Slot q0;
    q0.c16 = tf.pconn0[gp * tf.nsl * 256 + threadIdx.x]; q0.fl = tf.pfl0[gp * tf.nsl * 256 + threadIdx.x];
    q0.lam = tf.pgeo[gp * tf.nsl * 512 + threadIdx.x]; q0.mu = tf.pgeo[gp * tf.nsl * 512 + 256 + threadIdx.x];
    int nv[NPT];
    int kb[NPO];
    {
        double xv[NPT][2], cv[NPT][2];
#pragma unroll
        for (int j = 0; j < NPT; j++) nv[j] = pnodes[(size_t)lb * TILE_Q4X_MAX_NODES + threadIdx.x + 256 * j];   // (fixed stride per list position: see EbeTilePlan::pnodes_*)
        // (the penalty records of the owned nodes with the gathers, unconditionally: see k_ebe_tile_x)
#pragma unroll
        for (int j = 0; j < NPO; j++) kb[j] = (bc_node ? bc_node : hd)[bc_node ? nv[j] : 0];
        if constexpr (FUSED) {
            // everything the stage-in reads is asked for BEFORE the scalars of the opening are summed (they do not depend on each
            // other, and with fewer tiles than workgroup slots the kernel lasts as long as one tile's chain of latencies)
            double2 rr[NPT], dd[NPT], po[NPT], xo[NPT];
#pragma unroll
            for (int j = 0; j < NPT; j++) {
                const int i = threadIdx.x + 256 * j;
                const double2 z2 = make_double2(0.0, 0.0);
                rr[j] = (i < nnode) ? reinterpret_cast<const double2*>(cg.r)[nv[j]] : z2;
                dd[j] = (i < nnode) ? reinterpret_cast<const double2*>(cg.dinv)[nv[j]] : z2;
                po[j] = (i < nnode) ? reinterpret_cast<const double2*>(cg.p_old)[nv[j]] : z2;   // (iteration 0: whatever the buffer holds; not used)
                xo[j] = (i < nown) ? reinterpret_cast<const double2*>(cg.x)[nv[j]] : z2;
                const double2 cn = (i < nnode) ? reinterpret_cast<const double2*>(coords)[nv[j]] : z2;
                cv[j][0] = cn.x; cv[j][1] = cn.y;
            }
            const CgOpened op = cg_open_block(cg.open, blockIdx.x == 0 && threadIdx.x == 0, sh);
#pragma unroll
            for (int j = 0; j < NPT; j++) {
                const int i = threadIdx.x + 256 * j;
                if (op.i > 0 && i < nown) {   // the x += a p the iteration just completed owes the owned nodes
                    xo[j].x += op.aprev * po[j].x; xo[j].y += op.aprev * po[j].y;
                    reinterpret_cast<double2*>(cg.x)[nv[j]] = xo[j];
                }
                // (iteration 0: p = z; b = 0 there, but 0 * p_old must not meet whatever the buffer holds)
                xv[j][0] = (op.i > 0) ? cg_p(cg_z(dd[j].x, rr[j].x), op.b, po[j].x) : cg_z(dd[j].x, rr[j].x);
                xv[j][1] = (op.i > 0) ? cg_p(cg_z(dd[j].y, rr[j].y), op.b, po[j].y) : cg_z(dd[j].y, rr[j].y);
                if (!op.reason && i < nown) reinterpret_cast<double2*>(cg.p_new)[nv[j]] = make_double2(xv[j][0], xv[j][1]);
            }
            if (op.reason) return;   // the solve ends here (uniform over the grid: every workgroup reached the same verdict)
        } else {
#pragma unroll
            for (int j = 0; j < NPT; j++) {
                const int i = threadIdx.x + 256 * j;
                const double2 xn = (i < nnode) ? reinterpret_cast<const double2*>(x)[nv[j]] : make_double2(0.0, 0.0);   // (16 B aligned)
                const double2 cn = (i < nnode) ? reinterpret_cast<const double2*>(coords)[nv[j]] : make_double2(0.0, 0.0);
                xv[j][0] = xn.x; xv[j][1] = xn.y; cv[j][0] = cn.x; cv[j][1] = cn.y;
            }
        }
#pragma unroll
        for (int j = 0; j < NPT; j++) {
            const int i = threadIdx.x + 256 * j;
            if (i < nnode) {
#pragma unroll
                for (int d = 0; d < 2; d++) { xs[d * MN + i] = xv[j][d]; Xs[d * MN + i] = cv[j][d]; }
            }
            if (j < NPO && i < nown) { acc[i] = 0.0; acc[MO + i] = 0.0; }
        }
    }
    __syncthreads();
    q4x_steps(q0, nsteps, sv, elem_off, nown, tconn, tflags, geo, geo_stride, xs, Xs, acc, g, w);
    double dot = 0.0;
#pragma unroll
    for (int j = 0; j < NPO; j++) {
        const int i = threadIdx.x + 256 * j;
        if (i >= nown) continue;
        double sum[2];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            sum[d] = acc[d * MO + i];
            const double xi = xs[d * MN + i];
            if (bc_node && kb[j] >= 0 && bc_dof[kb[j] + d] >= 0) sum[d] += bc_shift[kb[j] + d] * xi;   // ApplyKBC INSERTs on the diagonal only
            dot += xi * sum[d];
        }
        reinterpret_cast<double2*>(y)[nv[j]] = make_double2(sum[0], sum[1]);
    }
    tile_finish(dot, t, sh, partial, rcount, rtotal, red);
}
// ---- KSPCG / PCJACOBI as ONE resident launch (quad4 plans whose tiles are all resident at once: config 2) -----------------------
// At a million dofs the CG of the launch-per-kernel path is a chain of latencies: 529 tiles, all resident, two kernels per iteration,
// and the product kernel spends most of its 24 us on its own start (tile record -> node ids -> stage-in loads -> the opening's
// partial sums).  Here one workgroup per tile stays for the whole solve: node ids, penalty records, coordinates (LDS), M^-1, the slot
// records of the first four steps, and r / x of its owned nodes live in registers or LDS across the iterations; p of all the tile's
// nodes (owned + rim) stays in LDS and is advanced by the tile itself -- the rim recomputed as in k_ebe_tile_q4x<true> -- from the rim
// nodes' r, the only vector data that crosses tiles (16 B per rim node and iteration, through global memory).  Two grid barriers
// (grid_barrier.h, 6-7 us each) per iteration:
//   [sum z.z, z.r partials -> KSPConvergedDefault, beta, b]  [x += a p, p = M^-1 r + b p in LDS]  [w = K p, tile steps]  p.w partial
//   -- barrier --  [sum p.w partials -> a]  [r -= a w, z = M^-1 r]  z.z, z.r partials, r of the owned nodes to global  -- barrier --
// Same operations per entry as the kernels of ksp.hip (cg_z, cg_p, cg_open_logic); the p.w partials are the launch path's (one
// per tile, summed in the same order), z.z and z.r are summed tile by tile instead of in index order -- the iteration path agrees
// with the launch path to round-off, not to the bit.
// two sums at once: the shuffle trees and the fixed tree over the four waves of block_sum (cg_device.h), one pair of barriers for both
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh)
{
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[w] = a; sh[4 + w] = b; }
    __syncthreads();
    a = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    b = (sh[4] + sh[5]) + (sh[6] + sh[7]);
}
__device__ __forceinline__ double2 ld2(const double* base, unsigned byte_off)
{
    return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st2(double* base, unsigned byte_off, double2 v)
{
    *reinterpret_cast<double2*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__global__ void __launch_bounds__(256, 3) k_cg_resident_q4(int nlist, const int* __restrict__ lhdr, const int* __restrict__ pnodes,
                                                           const uint4* __restrict__ tconn, const unsigned char* __restrict__ tflags,
                                                           const double* __restrict__ geo, size_t geo_stride,
                                                           const double* __restrict__ coords, const int* __restrict__ bc_node,
                                                           const int* __restrict__ bc_dof, const double* __restrict__ bc_shift,
                                                           double g, double w, CgResidentArgs cg, TileFirst tf)
{
    constexpr int NPT = TILE_Q4X_MAX_NODES / 256, NPO = TILE_Q4X_MAX_OWN / 256, MN = TILE_Q4X_LDS_NODES, MO = TILE_Q4X_LDS_OWN;
    constexpr int KEEP = 1;   // (the first step's slot record stays in registers; the others stream in one step ahead, as in k_ebe_tile_q4x: four kept records pushed the kernel over the 168 registers of three waves per SIMD)
    extern __shared__ double lds[];
    __shared__ double sh[8];
    __shared__ unsigned shu[4];
    // penalty dofs among the owned nodes (ApplyKBC: K_jj = PENALTY, applied as w_j += shift_j p_j): position and shift, gathered once
    // -- read per iteration through node -> record -> shift, two dependent loads, they held the tiles of a clamped edge 4 us behind
    // the others at every barrier.  More than RES_BC of them in a tile: that tile keeps the loads (bc_many).
    constexpr int RES_BC = 128;
    __shared__ int bc_pos[RES_BC];
    __shared__ double bc_val[RES_BC];
    __shared__ int bc_cnt;
    const unsigned chunk = gridDim.x / 8, lb = (blockIdx.x % 8) * chunk + blockIdx.x / 8;   // XCD-wise, as k_ebe_tile
    if (lb >= (unsigned)nlist) return;   // (padding of the grid: takes no part in the barriers)
    if (cg.test_leave && lb == 1) return;   // (test hook: a peer that never arrives)
    const bool writer = lb == 0 && threadIdx.x == 0;
    const int* hd = lhdr + 8 * (size_t)lb;
    const int nown = hd[1], nnode = hd[2], elem_off = hd[3], nsteps = hd[5], t = hd[6];
    double* xs = lds;               // p of the tile's nodes
    double* Xs = lds + 2 * MN;      // their coordinates
    double* acc = lds + 4 * MN;     // sums of the owned nodes
    const size_t gp = (size_t)tf.gp0 + lb;
    const int2 sv = tf.psteps[gp * 64 + (threadIdx.x & 63)];
    Q4Slot kept[KEEP];              // slot records of the first steps: registers for the whole solve where the plan stores them per tile
    const int nk = (int)tf.nsl < KEEP ? (int)tf.nsl : KEEP;
#pragma unroll
    for (int s = 0; s < KEEP; s++) {
        kept[s].fl = 0; kept[s].lam = 1.0; kept[s].mu = 1.0; kept[s].c16 = make_uint4(0, 0, 0, 0);
        if (s < nk) {
            const size_t o = (gp * tf.nsl + s) * 256 + threadIdx.x;
            kept[s].c16 = tf.pconn0[o]; kept[s].fl = tf.pfl0[o];
            kept[s].lam = tf.pgeo[(gp * tf.nsl + s) * 512 + threadIdx.x]; kept[s].mu = tf.pgeo[(gp * tf.nsl + s) * 512 + 256 + threadIdx.x];
        } else q4x_fetch(kept[s], s, nsteps, sv, elem_off, tconn, tflags, geo, geo_stride);   // (a plan that stores the first step only: the others from the lists, once)
    }
    // (node ids as 32-bit byte offsets of the 16-B entries: uniform base + lane offset, no 64-bit address per lane and vector kept alive)
    unsigned nv[NPT];
    int kb[NPO];
#pragma unroll
    for (int j = 0; j < NPT; j++) nv[j] = 16u * (unsigned)pnodes[(size_t)lb * TILE_Q4X_MAX_NODES + threadIdx.x + 256 * j];
#pragma unroll
    for (int j = 0; j < NPO; j++) kb[j] = (bc_node ? bc_node : hd)[bc_node ? nv[j] / 16u : 0];
    if (threadIdx.x == 0) bc_cnt = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPO; j++) {
        const int q = threadIdx.x + 256 * j;
        if (q >= nown || !bc_node || kb[j] < 0) continue;
        for (int d = 0; d < 2; d++)
            if (bc_dof[kb[j] + d] >= 0) {
                const int slot = atomicAdd(&bc_cnt, 1);
                if (slot < RES_BC) { bc_pos[slot] = q | (d << 16); bc_val[slot] = bc_shift[kb[j] + d]; }
            }
    }
    __syncthreads();
    const int nbc = bc_cnt;
    const bool bc_many = nbc > RES_BC;
    GridBarrierLane lane = grid_barrier_census(cg.bar, (unsigned)nlist, shu);
    // x = 0, r = b, p = 0; z = M^-1 r and the first partials.  Between the iterations only the node ids, the penalty records and the
    // first slot record stay in registers (the product needs the rest of the 168 that three waves per SIMD leave a lane): r, M^-1 and
    // x are read again where they are used -- loads that travel with the partial sums' round trip, from this XCD's L2 but for the rim
    double zz = 0.0, zr = 0.0;
#pragma unroll
    for (int j = 0; j < NPT; j++) {
        const int i = threadIdx.x + 256 * j;
        const double2 z2 = make_double2(0.0, 0.0);
        const double2 bj = (i < nnode) ? ld2(cg.b, nv[j]) : z2;
        const double2 dj = (i < nnode) ? ld2(cg.dinv, nv[j]) : z2;
        const double2 cn = (i < nnode) ? ld2(coords, nv[j]) : z2;
        if (i < nnode) { Xs[i] = cn.x; Xs[MN + i] = cn.y; xs[i] = 0.0; xs[MN + i] = 0.0; }
        if (j < NPO && i < nown) {
            st2(cg.x, nv[j], z2);
            st2(cg.r, nv[j], bj);
            const double z0 = cg_z(dj.x, bj.x), z1 = cg_z(dj.y, bj.y);
            zz += z0 * z0; zr += z0 * bj.x;
            zz += z1 * z1; zr += z1 * bj.y;
        }
    }
    block_sum2(zz, zr, sh);
    if (threadIdx.x == 0) { cg.pzz[t] = zz; cg.pzr[t] = zr; }
    if (!grid_barrier(cg.bar, lane, shu)) return;
    // the scalars of KSPSolve_CG, carried by every workgroup (all evaluate the same logic from the same sums)
    double betaold = 0.0, rnorm0 = 0.0, ttol = 0.0, aprev = 0.0, dpi_old = 0.0;
    int reason = 0, its = 0;
#ifdef FEM_RES_PROFILE
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memrealtime();
#define PH(n) { const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime(); tph[n] += tn_ - tlast; tlast = tn_; }
#else
#define PH(n)
#endif
    for (int i = 0;; i++) {
        // ---- opening of iteration i + 1: z.z, z.r over the tiles (thread k: partials k, k + 256, ...; then the fixed tree) ----
        double s0 = 0.0, s1 = 0.0;
        for (int k = threadIdx.x; k < nlist; k += 256) { s0 += cg.pzz[k]; s1 += cg.pzr[k]; }
        {
            // r and M^-1 of every node of the tile (the rim's r is the neighbours' of this iteration), x of the owned ones: asked for
            // with the partials
            double2 rr[NPT], dd[NPT], xo[NPO];
#pragma unroll
            for (int j = 0; j < NPT; j++) {
                const int q = threadIdx.x + 256 * j;
                const double2 z2 = make_double2(0.0, 0.0);
                rr[j] = (q < nnode) ? ld2(cg.r, nv[j]) : z2;
                dd[j] = (q < nnode) ? ld2(cg.dinv, nv[j]) : z2;
                if (j < NPO) xo[j] = (i > 0 && q < nown) ? ld2(cg.x, nv[j]) : z2;
            }
            zz = s0; zr = s1;
            block_sum2(zz, zr, sh);
            PH(0)
            const CgOpen o = cg_open_logic(i, zz, zr, betaold, rnorm0, ttol, cg.rtol, cg.abstol, cg.dtol, cg.maxit);
            if (i == 0) { rnorm0 = o.rnorm0; ttol = o.ttol; }
            if (o.its >= 0) its = o.its;
            if (writer) {
                cg.st->dp = o.dp;
                if (cg.hist && i < cg.hist_len) cg.hist[i] = o.dp;
            }
            // x += a p of the iteration just completed, then p = M^-1 r + b p for every node of the tile
#pragma unroll
            for (int j = 0; j < NPT; j++) {
                const int q = threadIdx.x + 256 * j;
                if (q >= nnode) continue;
                const double p0 = xs[q], p1 = xs[MN + q];
                if (j < NPO && i > 0 && q < nown) {
                    xo[j].x += aprev * p0; xo[j].y += aprev * p1;
                    st2(cg.x, nv[j], xo[j]);
                }
                if (!o.reason) {
                    xs[q] = (i > 0) ? cg_p(cg_z(dd[j].x, rr[j].x), o.b, p0) : cg_z(dd[j].x, rr[j].x);
                    xs[MN + q] = (i > 0) ? cg_p(cg_z(dd[j].y, rr[j].y), o.b, p1) : cg_z(dd[j].y, rr[j].y);
                }
                if (j < NPO && q < nown) { acc[q] = 0.0; acc[MO + q] = 0.0; }
            }
            if (o.reason) { reason = o.reason; break; }   // (uniform over the grid)
        }
        const double beta = zr;
        __syncthreads();
        PH(1)
        // ---- w = K p over the tile's steps; p.w over the owned nodes ----
        q4x_steps(kept[0], nsteps, sv, elem_off, nown, tconn, tflags, geo, geo_stride, xs, Xs, acc, g, w);
        PH(2)
        if (!bc_many && nbc > 0) {   // ApplyKBC INSERTs on the diagonal only: w_j += (PENALTY - assembled K_jj) p_j
            if ((int)threadIdx.x < nbc) {
                const int q = bc_pos[threadIdx.x] & 0xffff, d = bc_pos[threadIdx.x] >> 16;
                acc[d * MO + q] += bc_val[threadIdx.x] * xs[d * MN + q];
            }
            __syncthreads();
        }
        double dot = 0.0;
#pragma unroll
        for (int j = 0; j < NPO; j++) {
            const int q = threadIdx.x + 256 * j;
            if (q >= nown) continue;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                double sum = acc[d * MO + q];
                const double pi = xs[d * MN + q];
                if (bc_many && kb[j] >= 0 && bc_dof[kb[j] + d] >= 0) { sum += bc_shift[kb[j] + d] * pi; acc[d * MO + q] = sum; }
                dot += pi * sum;     // (acc keeps w of the owned node for the r update behind the barrier)
            }
        }
        for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off);   // (as tile_finish: the partial of the launch path)
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
        __syncthreads();
        if (threadIdx.x == 0) cg.ppw[t] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        PH(3)
        if (!grid_barrier(cg.bar, lane, shu)) { reason = 0; its = -1; break; }
        PH(4)
        // ---- a = beta / p.w; r -= a w; z = M^-1 r; partials ----
        double sp = 0.0;
        for (int k = threadIdx.x; k < nlist; k += 256) sp += cg.ppw[k];
        double2 ro[NPO], dq[NPO];
#pragma unroll
        for (int j = 0; j < NPO; j++) {
            const int q = threadIdx.x + 256 * j;
            const double2 z2 = make_double2(0.0, 0.0);
            ro[j] = (q < nown) ? ld2(cg.r, nv[j]) : z2;
            dq[j] = (q < nown) ? ld2(cg.dinv, nv[j]) : z2;
        }
        const double dpi = block_sum(sp, sh);
        PH(5)
        if (dpi == 0.0 || (i > 0 && dpi * dpi_old <= 0.0)) { reason = FEM_KSP_DIVERGED_INDEFINITE_MAT; break; }   // (x holds every completed iteration)
        const double a = beta / dpi;
        dpi_old = dpi; betaold = beta; aprev = a;
        zz = 0.0; zr = 0.0;
#pragma unroll
        for (int j = 0; j < NPO; j++) {
            const int q = threadIdx.x + 256 * j;
            if (q >= nown) continue;
            ro[j].x = ro[j].x - a * acc[q]; ro[j].y = ro[j].y - a * acc[MO + q];
            st2(cg.r, nv[j], ro[j]);
            const double z0 = cg_z(dq[j].x, ro[j].x), z1 = cg_z(dq[j].y, ro[j].y);
            zz += z0 * z0; zr += z0 * ro[j].x;
            zz += z1 * z1; zr += z1 * ro[j].y;
        }
        block_sum2(zz, zr, sh);
        if (threadIdx.x == 0) { cg.pzz[t] = zz; cg.pzr[t] = zr; }
        PH(6)
        if (!grid_barrier(cg.bar, lane, shu)) { reason = 0; its = -1; break; }
        PH(7)
    }
#ifdef FEM_RES_PROFILE
    if (threadIdx.x == 0 && (lb == 0 || lb == 100 || lb == 300 || lb == 528)) {
        const int slot = lb == 0 ? 0 : lb == 100 ? 1 : lb == 300 ? 2 : 3;
        for (int q = 0; q < 8; q++) cg.hist[cg.hist_len - 40 + slot * 8 + q] = (double)tph[q];
    }
#endif
    if (writer) {   // the verdict (x of the owned nodes is complete: the last x += a p was applied at the opening that ended the solve)
        cg.st->its = its; cg.st->it = its; cg.st->reason = reason; cg.st->done = 1; cg.st->stop = 1;
        cg.st->a = aprev; cg.st->betaold = betaold; cg.st->rnorm0 = rnorm0; cg.st->ttol = ttol;
    }
}

// the two factors of DMat2d per lane slot of a quad4 plan (k_ebe_tile_q4x): geo[k] = d12, geo[slots + k] = d44
__global__ void __launch_bounds__(256) k_tile_geom_q4(size_t ntel, const int* __restrict__ telems, const double* __restrict__ Ev,
                                                      const double* __restrict__ nuv, double* __restrict__ geo)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ntel) return;
    const int e = telems[k];
    const DCoef D = dmat_coef(e >= 0 ? Ev[e] : 1.0, e >= 0 ? nuv[e] : 0.25);
    geo[k] = D.d12; geo[ntel + k] = D.d44;
}

// the parallelepiped test of hex_forces on the host (same tolerance)
static bool host_hex_affine(const double (&X)[8][3])
{
    for (int j = 0; j < 3; j++) {
        double v[8];
        for (int a = 0; a < 8; a++) v[a] = X[a][j];
        const double s00 = v[1] + v[0], d00 = v[1] - v[0], s10 = v[2] + v[3], d10 = v[2] - v[3];
        const double s01 = v[5] + v[4], d01 = v[5] - v[4], s11 = v[6] + v[7], d11 = v[6] - v[7];
        const double ss0 = s10 + s00, sd0 = s10 - s00, ds0 = d10 + d00, dd0 = d10 - d00;
        const double ss1 = s11 + s01, sd1 = s11 - s01, ds1 = d11 + d01, dd1 = d11 - d01;
        const double c0 = 0.125 * (ss1 + ss0), cx = 0.125 * (ds1 + ds0), cy = 0.125 * (sd1 + sd0), cz = 0.125 * (ss1 - ss0);
        // (the device test scales the mixed terms by g <= 1 first: this one is the stricter of the two)
        const double cxy = 0.125 * (dd1 + dd0), cxz = 0.125 * (ds1 - ds0), cyz = 0.125 * (sd1 - sd0), cxyz = 0.125 * (dd1 - dd0);
        const double tol = 0x1p-50 * (std::fabs(c0) + std::fabs(cx) + std::fabs(cy) + std::fabs(cz));
        if (!(std::fabs(cxy) <= tol && std::fabs(cxz) <= tol && std::fabs(cyz) <= tol && std::fabs(cxyz) <= tol)) return false;
    }
    return true;
}

}  // namespace fem

// ---- set-up ---------------------------------------------------------------------------------------------------------
namespace {

struct TileOut {   // what one worker thread produced for its range of boxes (offsets local to the worker)
    std::vector<int> hdr, tnodes, telems, laff, lgen, lcart;
    std::vector<int2> steps;
    std::vector<uint4> tconn;
    std::vector<unsigned char> tflags, taff;   // taff: the slot belongs to a tile of parallelepipeds
    std::vector<unsigned char> tif;            // per tile: owns a node shared with another domain
    long long visits = 0;
    int ntiles = 0, max_nnode = 0;
    const char* why = nullptr;   // refusal
    long long why_a = 0, why_b = 0;
};

// open-addressing map int -> int for tile-sized sets (a few thousand keys), cleared by undoing its insertions
struct SmallMap {
    std::vector<int> key, val, used;
    unsigned mask;
    explicit SmallMap(unsigned cap_pow2) : key(cap_pow2, -1), val(cap_pow2, 0), mask(cap_pow2 - 1) {}
    int* slot(int k, bool& fresh)
    {
        unsigned h = ((unsigned)k * 2654435761u) & mask;
        while (key[h] != -1 && key[h] != k) h = (h + 1) & mask;
        fresh = key[h] == -1;
        if (fresh) { key[h] = k; used.push_back((int)h); }
        return &val[h];
    }
    int get(int k) const
    {
        unsigned h = ((unsigned)k * 2654435761u) & mask;
        while (key[h] != k) h = (h + 1) & mask;
        return val[h];
    }
    void clear() { for (int h : used) key[h] = -1; used.clear(); }
};

// A worker thread inherits the CPU mask of the thread that starts it, and a host program may have pinned that one to a
// single core (an OpenMP runtime with OMP_PROC_BIND does; bench.py's process is such a case): give the worker every CPU
// the process may use -- the cgroup quota still applies.
void unpin_this_thread()
{
    cpu_set_t all;
    CPU_ZERO(&all);
    for (int i = 0; i < CPU_SETSIZE; i++) CPU_SET(i, &all);
    (void)sched_setaffinity(0, sizeof(all), &all);
}

int worker_count()
{
    static const int env = std::getenv("FEM_AMD_SETUP_THREADS") ? std::atoi(std::getenv("FEM_AMD_SETUP_THREADS")) : 0;
    const int hc = (int)std::thread::hardware_concurrency();
    return std::max(1, env > 0 ? env : std::min(16, hc > 0 ? hc : 1));
}
template <class F>
void parallel_ranges(long long n, int nthreads, F f)   // f(thread, first, last)
{
    nthreads = (int)std::max<long long>(1, std::min<long long>(nthreads, n));
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; i++)
        th.emplace_back([=] { unpin_this_thread(); f(i, n * i / nthreads, n * (i + 1) / nthreads); });
    f(0, 0LL, n / nthreads);
    for (auto& x : th) x.join();
}

}  // namespace

// the tiles' records in the order of each list: [node_off, nown, nnode, elem_off, step_off, nsteps, tile, 0] per position
__global__ void __launch_bounds__(256) k_tile_list_hdr(int n, const int* __restrict__ list, const int* __restrict__ hdr, int* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n) return;
    const int t = list[i];
    for (int j = 0; j < 6; j++) out[8 * i + j] = hdr[(size_t)6 * t + j];
    out[8 * i + 6] = t; out[8 * i + 7] = 0;
}
// the listed tiles' node lists at a fixed stride (one workgroup per list position; zeros behind a tile's last node)
__global__ void __launch_bounds__(256) k_tile_list_nodes(int n, const int* __restrict__ lhdr, const int* __restrict__ nodes, int stride, int* __restrict__ out)
{
    const size_t i = blockIdx.x;
    const int node_off = lhdr[8 * i], nnode = lhdr[8 * i + 2];
    for (int k = threadIdx.x; k < stride; k += 256) out[i * stride + k] = k < nnode ? nodes[(size_t)node_off + k] : 0;
}
void ebe_tile_list_headers(EbeTilePlan& P, bool quad, hipStream_t s)
{
    auto go = [&](DBuf<int>& out, const DBuf<int>& list, int n) {
        out.alloc((size_t)std::max(n, 1) * 8);
        if (n) hipLaunchKernelGGL(k_tile_list_hdr, dim3(cdiv((size_t)n, 256)), dim3(256), 0, s, n, list.p, P.hdr.p, out.p);
    };
    go(P.lhdr_cart, P.list_cart, P.n_cart); go(P.lhdr_aff, P.list_aff, P.n_aff); go(P.lhdr_gen, P.list_gen, P.n_gen);
    // plans of the kernels that keep x of a tile in LDS (small hex8 tiles: k_ebe_tile_x; quad4 tiles: k_ebe_tile_q4x)
    P.pnode_stride = 0;
    if (!quad && P.n_cart + P.n_aff > 0 && P.max_nnode <= TILE_X_LDS_NODES && P.max_nown <= TILE_X_MAX_OWN) P.pnode_stride = TILE_X_MAX_NODES;
    else if (quad && P.max_nnode <= TILE_Q4X_LDS_NODES && P.max_nown <= TILE_Q4X_LDS_OWN) P.pnode_stride = TILE_Q4X_MAX_NODES;
    if (P.pnode_stride) {
        auto pad = [&](DBuf<int>& out, const DBuf<int>& lh, int n) {
            out.alloc((size_t)std::max(n, 1) * P.pnode_stride);
            if (n) hipLaunchKernelGGL(k_tile_list_nodes, dim3(n), dim3(256), 0, s, n, lh.p, P.nodes.p, P.pnode_stride, out.p);
        };
        pad(P.pnodes_cart, P.lhdr_cart, P.n_cart); pad(P.pnodes_aff, P.lhdr_aff, P.n_aff); pad(P.pnodes_gen, P.lhdr_gen, P.n_gen);
    }
    FEM_HIP(hipGetLastError());
}
// step table, first-step slot records and geometry of one listed tile at fixed strides (EbeTilePlan::psteps ...); NG = 0: quad4
__global__ void __launch_bounds__(256) k_tile_first(int n, const int* __restrict__ lhdr, const int2* __restrict__ steps,
                                                    const uint4* __restrict__ tconn, const unsigned char* __restrict__ tflags,
                                                    const double* __restrict__ geo, size_t geo_stride, int ng, size_t gp0, int nsl,
                                                    int2* __restrict__ psteps, uint4* __restrict__ pconn0, unsigned char* __restrict__ pfl0,
                                                    double* __restrict__ pgeo, const int* __restrict__ telems, const double* __restrict__ Ev,
                                                    const double* __restrict__ nuv, double g)
{
    const size_t i = blockIdx.x, gp = gp0 + i;
    const int* hd = lhdr + 8 * i;
    const int elem_off = hd[3], step_off = hd[4], nsteps = hd[5], uref = hd[7];
    if (threadIdx.x < 64) psteps[gp * 64 + threadIdx.x] = (int)threadIdx.x < nsteps ? steps[step_off + threadIdx.x] : make_int2(0, 0);
    for (int st = 0; st < nsl; st++) {
        const int2 s0 = st < nsteps ? steps[step_off + st] : make_int2(0, 0);
        const bool have = (int)threadIdx.x < (s0.y & 0x7fffffff);
        const size_t k = (size_t)elem_off + s0.x + threadIdx.x, o = (gp * nsl + st) * 256 + threadIdx.x;
        pconn0[o] = have ? tconn[k] : make_uint4(0, 0, 0, 0);
        pfl0[o] = have ? tflags[k] : (unsigned char)0;
        if (ng == 0) {   // quad4: lam, mu of the step's lane slots
            pgeo[(gp * nsl + st) * 512 + threadIdx.x] = (have && geo) ? geo[k] : 1.0;
            pgeo[(gp * nsl + st) * 512 + 256 + threadIdx.x] = (have && geo) ? geo[geo_stride + k] : 1.0;
        }
    }
    if (ng != 0 && (int)threadIdx.x < 16) {
        double v = (uref > 0 && (int)threadIdx.x < ng && geo) ? geo[(size_t)threadIdx.x * geo_stride + (uref - 1)] : 0.0;
        if (threadIdx.x >= 14) {   // E, nu of the tile's elements (one record: one material), for the nodal stress over the tiles
            const int e = uref > 0 ? telems[uref - 1] : -1;
            v = e >= 0 ? (threadIdx.x == 14 ? Ev[e] : nuv[e]) : (threadIdx.x == 14 ? 1.0 : 0.25);
        }
        pgeo[gp * TILE_GEO_STRIDE + threadIdx.x] = v;
    }
    if (ng == 5 && threadIdx.x == 16) {   // a tile of bricks: the coefficients of hex_forces_cart_k
        double K[CART_NK] = {};
        if (uref > 0 && geo) {
            const size_t e = (size_t)(uref - 1);
            const double a[3] = {geo[e], geo[geo_stride + e], geo[2 * geo_stride + e]};
            hex_cart_consts(a, geo[3 * geo_stride + e], geo[4 * geo_stride + e], g, K);
        }
        for (int j = 0; j < CART_NK; j++) pgeo[gp * TILE_GEO_STRIDE + 16 + j] = K[j];
    }
}
void ebe_tile_first_records(fem_mesh m, EbeTilePlan& P, bool quad, hipStream_t s)
{
    if (!P.pnode_stride) return;
    auto& grp = m->groups[0];
    const size_t npos = std::max<size_t>((size_t)P.n_cart + P.n_aff + P.n_gen, 1);
    P.pslot_steps = P.max_nsteps <= 4 ? 4 : 1;
    const size_t nsl = (size_t)P.pslot_steps;
    P.psteps.alloc(npos * 64); P.pconn0.alloc(npos * nsl * 256); P.pfl0.alloc(npos * nsl * 256);
    P.pgeo.alloc(quad ? npos * nsl * 512 : npos * TILE_GEO_STRIDE);
    auto go = [&](const DBuf<int>& lh, int n, int ng, size_t gp0) {
        if (n) hipLaunchKernelGGL(k_tile_first, dim3(n), dim3(256), 0, s, n, lh.p, P.steps.p, P.conn16.p, P.flags.p, P.geo.p, (size_t)P.slots, ng, gp0,
                                  P.pslot_steps, P.psteps.p, P.pconn0.p, P.pfl0.p, P.pgeo.p, P.elems.p, grp.E.p, grp.nu.p,
                                  std::fabs(host_elem_tables()[grp.type].xi[0][0]));
    };
    if (quad) go(P.lhdr_gen, P.n_gen, 0, 0);
    else { go(P.lhdr_cart, P.n_cart, 5, 0); go(P.lhdr_aff, P.n_aff, 11, (size_t)P.n_cart); }
    FEM_HIP(hipGetLastError());
}
// quad4 plans: the D factors of every lane slot's element (k_ebe_tile_q4x reads them with the slot record)
void ebe_tile_quad_factors(fem_mesh m, EbeTilePlan& P, hipStream_t s)
{
    auto& grp = m->groups[0];
    if (grp.type != FEM_QUA || P.slots < 1) return;
    P.geo.alloc((size_t)P.slots * 2);
    hipLaunchKernelGGL(k_tile_geom_q4, dim3((unsigned)((P.slots + 255) / 256)), dim3(256), 0, s, (size_t)P.slots, P.elems.p, grp.E.p,
                       grp.nu.p, P.geo.p);
    FEM_HIP(hipGetLastError());
}

// geometry records of the elements of the stored-geometry tiles (taff: 0 / 1 / 2 per lane slot), written once per mesh
void ebe_tile_store_geometry(fem_mesh m, EbeTilePlan& P, const unsigned char* d_taff, hipStream_t s)
{
    auto& grp = m->groups[0];
    P.geo.alloc((size_t)P.slots * (P.n_aff ? 11 : 5));   // (axis-aligned bricks alone: the diagonal of adj(J) and the two factors)
    const ElemTable& tab = host_elem_tables()[FEM_HEX];   // (only hex8 tiles are ever in those lists)
    hipLaunchKernelGGL(k_tile_geom, dim3((unsigned)((P.slots + 255) / 256)), dim3(256), 0, s, (size_t)P.slots, P.elems.p, d_taff,
                       grp.conn.p, m->coords.p, grp.E.p, grp.nu.p, tab.w[0], P.geo.p);
    // tiles with one record for all their elements (FEM_AMD_TILE_UNIFORM=0: every tile reads its records slot by slot)
    const bool uni = !(std::getenv("FEM_AMD_TILE_UNIFORM") && std::atoi(std::getenv("FEM_AMD_TILE_UNIFORM")) == 0);
    P.n_uniform = 0;
    if (uni && P.slots < 0x7fffffffLL) {
        DBuf<unsigned long long> cnt(1);
        cnt.zero(s);
        if (P.n_cart) hipLaunchKernelGGL(k_tile_geo_uniform<5>, dim3(P.n_cart), dim3(256), 0, s, P.n_cart, P.lhdr_cart.p, P.steps.p, P.elems.p, P.geo.p, (size_t)P.slots, cnt.p);
        if (P.n_aff) hipLaunchKernelGGL(k_tile_geo_uniform<11>, dim3(P.n_aff), dim3(256), 0, s, P.n_aff, P.lhdr_aff.p, P.steps.p, P.elems.p, P.geo.p, (size_t)P.slots, cnt.p);
        unsigned long long h = 0;
        cnt.download(&h, 1, s);
        FEM_HIP(hipStreamSynchronize(s));
        P.n_uniform = (long long)h;
    }
    FEM_HIP(hipGetLastError());
}

// Builds the plan.  Returns nullptr when the mesh is not one group of hex8 (or quad4) elements or a tile would not fit
// (nodes, colours): the caller keeps the two-kernel product.  The plan is built on the device (ebe_plan.hip: no copy of the
// mesh to the host, a few milliseconds at config 3); FEM_AMD_TILE_PLAN=host selects the host builder below, which produces
// the same plan with worker threads (hundreds of milliseconds) and is kept as the cross-check of the device one.
std::shared_ptr<EbeTilePlan> ebe_tile_build(fem_mesh m, hipStream_t s)
{
    const char* how = std::getenv("FEM_AMD_TILE_PLAN");
    std::shared_ptr<EbeTilePlan> P = (how && std::string(how) == "host") ? ebe_tile_build_host(m, s) : ebe_tile_build_device(m, s);
    if (P) ebe_tile_first_records(m, *P, m->pdim == 2, s);
    if (P) {
        // what one CG iteration walks through: the lists (slot records, flags, stored geometry, node lists) and five vectors, against
        // the 256 MB Infinity Cache (FEM_AMD_TILE_STREAM_MIN_MB: the threshold; measured -- 203 MB, 100^3 elements: plain loads 0.0443 ms
        // per product, streaming 0.0488; 348 MB, 120^3: 0.0790 / 0.0755; 665 MB, config 3: 0.1291 / 0.1229)
        const double min_mb = std::getenv("FEM_AMD_TILE_STREAM_MIN_MB") ? std::atof(std::getenv("FEM_AMD_TILE_STREAM_MIN_MB")) : 256.0;
        const double geo_words = P->geo.p ? (m->pdim == 2 ? 2.0 : (P->n_aff ? 11.0 : 5.0)) : 0.0;
        const double lists = (double)P->slots * (21.0 + 8.0 * geo_words) + 4.0 * (double)P->tile_nodes;
        const double vectors = 5.0 * 8.0 * (double)m->nnds * m->pdim;
        P->stream = (lists + vectors) > min_mb * 1048576.0;
    }
    return P;
}

// The host builder (worker threads over contiguous ranges of boxes; the mesh is copied to the host first).
std::shared_ptr<EbeTilePlan> ebe_tile_build_host(fem_mesh m, hipStream_t s)
{
    if (m->groups.size() != 1 || !m->pattern) return nullptr;
    const bool hex = m->groups[0].type == FEM_HEX && m->pdim == 3;
    if (!hex && !(m->groups[0].type == FEM_QUA && m->pdim == 2)) return nullptr;
    const int npe = hex ? 8 : 4, pd = hex ? 3 : 2, nfp = npe / 2;   // nodes per element, dimensions, nodes of a face
    static const bool verbose = std::getenv("FEM_AMD_EBE_VERBOSE") != nullptr;
    auto refuse = [&](const char* why, long long a = 0, long long b = 0) -> std::shared_ptr<EbeTilePlan> {
        if (verbose) std::fprintf(stderr, "[fem_amd] ebe tiles: not used (%s %lld %lld)\n", why, a, b);
        return nullptr;
    };
    auto& grp = m->groups[0];
    const int nel = grp.nel, nn = m->nnds;
    if (nel < 1) return nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const int NT = worker_count();
    auto tl = t0;
    // connectivity, coordinates and the node -> element lists of the symbolic phase (codes element << 3 | local node, ascending)
    std::vector<int> conn((size_t)nel * npe), nptr((size_t)nn + 1);
    std::vector<double> xyz((size_t)nn * pd);
    std::vector<unsigned> nlst((size_t)nel * npe);
    FEM_HIP(hipMemcpyAsync(conn.data(), grp.conn.p, conn.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipMemcpyAsync(xyz.data(), m->coords.p, xyz.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipMemcpyAsync(nptr.data(), m->pattern->n2e_ptr.p, nptr.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipMemcpyAsync(nlst.data(), m->pattern->n2e.p, nlst.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    FEM_HIP(hipStreamSynchronize(s));
    if (nptr[nn] != nel * npe) return refuse("node lists do not match the group");
    if (verbose) { const auto tn = std::chrono::steady_clock::now(); std::fprintf(stderr, "[fem_amd] ebe tiles set-up: copies to the host %.0f ms\n", std::chrono::duration<double, std::milli>(tn - tl).count()); tl = tn; }
    // grid of the mean element size over the bounding box of the nodes; node -> box of NB nodes per direction
    std::vector<unsigned char> aff((size_t)nel);
    std::vector<std::array<double, 9>> part((size_t)NT, std::array<double, 9>{1e300, 1e300, 1e300, -1e300, -1e300, -1e300, 0, 0, 0});
    std::vector<std::array<double, 12>> edges((size_t)NT, std::array<double, 12>{});   // sums of the signed mean edge vectors [k][d], of their lengths [9 + k]
    parallel_ranges(nn, NT, [&](int th, long long v0, long long v1) {
        std::array<double, 9> p = part[th];
        for (long long v = v0; v < v1; v++)
            for (int d = 0; d < pd; d++) { p[d] = std::min(p[d], xyz[(size_t)v * pd + d]); p[3 + d] = std::max(p[3 + d], xyz[(size_t)v * pd + d]); }
        part[th] = p;
    });
    parallel_ranges(nel, NT, [&](int th, long long e0, long long e1) {
        std::array<double, 9> p = part[th];      // (sums in locals: the workers' records share cache lines)
        std::array<double, 12> eg{};
        for (long long e = e0; e < e1; e++) {
            double X[8][3] = {};
            for (int a = 0; a < npe; a++)
                for (int d = 0; d < pd; d++) X[a][d] = xyz[(size_t)conn[(size_t)e * npe + a] * pd + d];
            // spacing of the grid the nodes sit on: per direction the longest of the mean edge vectors of the element
            // (exact on a regular brick, unbiased under node jitter -- the bounding box of a jittered element is not)
            for (int d = 0; d < pd; d++) {
                if (hex) {
                    const double ex = 0.25 * std::fabs((X[1][d] + X[2][d] + X[5][d] + X[6][d]) - (X[0][d] + X[3][d] + X[4][d] + X[7][d]));
                    const double ey = 0.25 * std::fabs((X[3][d] + X[2][d] + X[7][d] + X[6][d]) - (X[0][d] + X[1][d] + X[4][d] + X[5][d]));
                    const double ez = 0.25 * std::fabs((X[4][d] + X[5][d] + X[6][d] + X[7][d]) - (X[0][d] + X[1][d] + X[2][d] + X[3][d]));
                    p[6 + d] += std::max(ex, std::max(ey, ez));
                } else {
                    const double ex = 0.5 * std::fabs((X[1][d] + X[2][d]) - (X[0][d] + X[3][d]));
                    const double ey = 0.5 * std::fabs((X[3][d] + X[2][d]) - (X[0][d] + X[1][d]));
                    p[6 + d] += std::max(ex, ey);
                }
            }
            {   // the element's mean edge vectors along xi, eta, zeta (signed): the lattice the mesh sits on, if it is one
                double ev[3][3] = {};
                for (int d = 0; d < pd; d++) {
                    if (hex) {
                        ev[0][d] = 0.25 * ((X[1][d] + X[2][d] + X[5][d] + X[6][d]) - (X[0][d] + X[3][d] + X[4][d] + X[7][d]));
                        ev[1][d] = 0.25 * ((X[3][d] + X[2][d] + X[7][d] + X[6][d]) - (X[0][d] + X[1][d] + X[4][d] + X[5][d]));
                        ev[2][d] = 0.25 * ((X[4][d] + X[5][d] + X[6][d] + X[7][d]) - (X[0][d] + X[1][d] + X[2][d] + X[3][d]));
                    } else {
                        ev[0][d] = 0.5 * ((X[1][d] + X[2][d]) - (X[0][d] + X[3][d]));
                        ev[1][d] = 0.5 * ((X[3][d] + X[2][d]) - (X[0][d] + X[1][d]));
                    }
                }
                for (int k = 0; k < pd; k++) {
                    double n2 = 0.0;
                    for (int d = 0; d < pd; d++) { eg[3 * k + d] += ev[k][d]; n2 += ev[k][d] * ev[k][d]; }
                    eg[9 + k] += std::sqrt(n2);
                }
            }
            unsigned char cls = (hex && host_hex_affine(X)) ? 1 : 0;
            if (cls) {   // an axis-aligned brick: the three edges at node 0 run along x, y, z exactly
                const bool cart = X[1][1] == X[0][1] && X[1][2] == X[0][2] && X[3][0] == X[0][0] && X[3][2] == X[0][2] &&
                                  X[4][0] == X[0][0] && X[4][1] == X[0][1];
                if (cart) cls = 2;
            }
            aff[(size_t)e] = cls;
        }
        part[th] = p; edges[th] = eg;
    });
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, hsum[3] = {0, 0, 0};
    for (auto& p : part)
        for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], p[d]); hi[d] = std::max(hi[d], p[3 + d]); hsum[d] += p[6 + d]; }
    if (verbose) { const auto tn = std::chrono::steady_clock::now(); std::fprintf(stderr, "[fem_amd] ebe tiles set-up: element pass %.0f ms\n", std::chrono::duration<double, std::milli>(tn - tl).count()); tl = tn; }
    // FEM_AMD_TILE_NB="nx,ny,nz": box size (experiments; 15,15,7 won the sweep at config 3)
    int NB[3] = {hex ? 15 : 31, hex ? 15 : 31, hex ? 7 : 1};   // quad4: 32 x 32 elements, 1,089 nodes of 20 B
    if (hex) {   // a mesh of parallelepipeds (stored geometry): small tiles (16 x 8 x 8 elements), x of a tile in LDS, three (bricks) or
                 // two tiles per CU (k_ebe_tile_x)
        long long npar = 0;
        for (int e = 0; e < nel; e++) npar += aff[(size_t)e] != 0;
        if (npar * 10 >= (long long)nel * 9) { NB[1] = 7; NB[2] = 7; }
    }
    if (const char* ps = std::getenv("FEM_AMD_TILE_NB")) {
        int q[3] = {0, 0, 0};
        if (std::sscanf(ps, "%d,%d,%d", &q[0], &q[1], &q[2]) == 3 && q[0] > 0 && q[1] > 0 && q[2] > 0)
            for (int d = 0; d < pd; d++) NB[d] = q[d];
    }
    // node limits of a tile: the LDS image of the kernels it is meant for (small tiles: k_ebe_tile_x); a box that holds more (a
    // graded or distorted mesh) is halved
    const bool small_tiles = hex && NB[1] <= 7 && NB[2] <= 7;
    const int max_nodes = small_tiles ? TILE_X_LDS_NODES : TILE_MAX_NODES, max_own = small_tiles ? TILE_X_MAX_OWN : 0x7fffffff;
    long long nbox[3] = {1, 1, 1};
    double h[3] = {1.0, 1.0, 1.0};
    // A mesh whose elements all point the same way (a structured grid, sheared or rotated as it may be) is binned in the
    // coordinates of its own lattice -- L = mean edge vectors, xi = L^-1 x, unit spacing: boxes follow the grid lines, the
    // chains of x-neighbours run along the first box direction whatever the orientation in space.  Otherwise (the signed
    // edge vectors of an unstructured mesh cancel) the boxes are axis-aligned in space.
    bool lattice = false;
    {
        double Lm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, len[3] = {1, 1, 1};
        bool ok = true;
        for (int k = 0; k < pd; k++) {
            double sk[3] = {0, 0, 0}, nk = 0.0, n2 = 0.0;
            for (auto& eg : edges) { for (int d = 0; d < pd; d++) sk[d] += eg[3 * k + d]; nk += eg[9 + k]; }
            for (int d = 0; d < pd; d++) { Lm[d][k] = sk[d] / nel; n2 += Lm[d][k] * Lm[d][k]; }
            len[k] = std::sqrt(n2);
            ok = ok && nk > 0.0 && len[k] >= 0.9 * nk / nel;
        }
        const double det = Lm[0][0] * (Lm[1][1] * Lm[2][2] - Lm[1][2] * Lm[2][1]) - Lm[0][1] * (Lm[1][0] * Lm[2][2] - Lm[1][2] * Lm[2][0]) +
                           Lm[0][2] * (Lm[1][0] * Lm[2][1] - Lm[1][1] * Lm[2][0]);
        ok = ok && std::fabs(det) >= 0.1 * len[0] * len[1] * len[2];
        static const bool lattice_env = !(std::getenv("FEM_AMD_TILE_LATTICE") && std::atoi(std::getenv("FEM_AMD_TILE_LATTICE")) == 0);
        if (ok && lattice_env) {
            double Li[3][3];
            Li[0][0] = (Lm[1][1] * Lm[2][2] - Lm[1][2] * Lm[2][1]) / det; Li[0][1] = (Lm[0][2] * Lm[2][1] - Lm[0][1] * Lm[2][2]) / det;
            Li[0][2] = (Lm[0][1] * Lm[1][2] - Lm[0][2] * Lm[1][1]) / det; Li[1][0] = (Lm[1][2] * Lm[2][0] - Lm[1][0] * Lm[2][2]) / det;
            Li[1][1] = (Lm[0][0] * Lm[2][2] - Lm[0][2] * Lm[2][0]) / det; Li[1][2] = (Lm[0][2] * Lm[1][0] - Lm[0][0] * Lm[1][2]) / det;
            Li[2][0] = (Lm[1][0] * Lm[2][1] - Lm[1][1] * Lm[2][0]) / det; Li[2][1] = (Lm[0][1] * Lm[2][0] - Lm[0][0] * Lm[2][1]) / det;
            Li[2][2] = (Lm[0][0] * Lm[1][1] - Lm[0][1] * Lm[1][0]) / det;
            // the node coordinates become lattice coordinates (the copy on the host only serves the boxes from here on; the
            // element classes above were taken from the real coordinates)
            std::vector<std::array<double, 6>> bb((size_t)NT, std::array<double, 6>{1e300, 1e300, 1e300, -1e300, -1e300, -1e300});
            parallel_ranges(nn, NT, [&](int th, long long v0, long long v1) {
                std::array<double, 6> b = bb[th];
                for (long long v = v0; v < v1; v++) {
                    double X[3] = {0, 0, 0}, q[3];
                    for (int d = 0; d < pd; d++) X[d] = xyz[(size_t)v * pd + d];
                    for (int d = 0; d < pd; d++) q[d] = Li[d][0] * X[0] + Li[d][1] * X[1] + Li[d][2] * X[2];
                    for (int d = 0; d < pd; d++) { xyz[(size_t)v * pd + d] = q[d]; b[d] = std::min(b[d], q[d]); b[3 + d] = std::max(b[3 + d], q[d]); }
                }
                bb[th] = b;
            });
            for (int d = 0; d < 3; d++) { lo[d] = 1e300; hi[d] = -1e300; }
            for (auto& b : bb)
                for (int d = 0; d < pd; d++) { lo[d] = std::min(lo[d], b[d]); hi[d] = std::max(hi[d], b[3 + d]); }
            for (int d = 0; d < pd; d++) hsum[d] = (double)nel;   // unit spacing
            lattice = true;
        }
    }
    for (int d = 0; d < pd; d++) {
        h[d] = hsum[d] / nel;
        if (!(h[d] > 0.0)) return refuse("flat mesh");
        const long long npts = (long long)std::llround((hi[d] - lo[d]) / h[d]) + 1;
        nbox[d] = (npts + NB[d] - 1) / NB[d];
    }
    if ((double)nbox[0] * nbox[1] * nbox[2] > 4.0 * nn + 1024.0) return refuse("grid does not describe the mesh");
    auto box_of = [&](long long v) {
        long long b[3] = {0, 0, 0};
        for (int d = 0; d < pd; d++) {
            long long i = FEM_PLAN_GRID(xyz[(size_t)v * pd + d], lo[d], h[d]);
            b[d] = std::min(std::max(i / NB[d], 0LL), nbox[d] - 1);
        }
        return b[0] + nbox[0] * (b[1] + nbox[1] * b[2]);
    };
    // nodes sorted by box (counting sort: ascending node id inside a box)
    const long long nboxes = nbox[0] * nbox[1] * nbox[2];
    std::vector<long long> nbx((size_t)nn);
    parallel_ranges(nn, NT, [&](int, long long v0, long long v1) { for (long long v = v0; v < v1; v++) nbx[(size_t)v] = box_of(v); });
    std::vector<int> bstart((size_t)nboxes + 1, 0);
    for (int v = 0; v < nn; v++) bstart[(size_t)nbx[v] + 1]++;
    for (long long b = 0; b < nboxes; b++) bstart[b + 1] += bstart[b];
    std::vector<int> border((size_t)nn), bfill(bstart.begin(), bstart.end() - 1);
    for (int v = 0; v < nn; v++) border[(size_t)bfill[nbx[v]]++] = v;
    nbx.clear(); nbx.shrink_to_fit();
    // the boxes are dealt to the workers in contiguous ranges of about equal node counts
    std::vector<long long> bsplit((size_t)NT + 1, nboxes);
    bsplit[0] = 0;
    {
        long long b = 0;
        for (int i = 1; i < NT; i++) {
            const long long target = (long long)nn * i / NT;
            while (b < nboxes && bstart[b] < target) b++;
            bsplit[i] = b;
        }
    }
    if (verbose) { const auto tn = std::chrono::steady_clock::now(); std::fprintf(stderr, "[fem_amd] ebe tiles set-up: node boxes %.0f ms (%s, %d x %d x %d nodes per box)\n", std::chrono::duration<double, std::milli>(tn - tl).count(), lattice ? "in the coordinates of the mesh's lattice" : "axis-aligned in space", NB[0], NB[1], NB[2]); tl = tn; }
    // nodes shared with other domains (a partitioned mesh): tiles that own one are computed first, so that the halo exchange
    // of their rows can run beside the others (ksp.hip)
    std::vector<unsigned char> shared_node;
    if (Halo* hl = m->halo.get(); hl && hl->ninodes > 0) {
        std::vector<int> inodes((size_t)hl->ninodes);
        FEM_HIP(hipMemcpyAsync(inodes.data(), hl->inodes.p, inodes.size() * sizeof(int), hipMemcpyDeviceToHost, s));
        FEM_HIP(hipStreamSynchronize(s));
        shared_node.assign((size_t)nn, 0);
        for (int v : inodes) shared_node[(size_t)v] = 1;
    }
    std::vector<TileOut> outs((size_t)NT);
    static const int c_plus[4] = {1, 2, 5, 6}, c_minus[4] = {0, 3, 4, 7};
    auto work = [&](int th) {
        TileOut& o = outs[th];
        std::vector<unsigned> cmask;
        std::vector<int> el, nd, col, chain_start, lconn;   // lconn: tile-local node numbers of el's nodes
        std::vector<std::pair<unsigned long long, int>> ekey;
        SmallMap emap(1u << 15), nmap(1u << 15);            // elements / nodes of the tile in hand (far below 32 K each)
        std::vector<std::vector<int>> todo;   // owned-node sets still to be turned into tiles (a box, or the halves of one)
        std::vector<unsigned char> todo_force;   // per set: halve it before anything else
        for (long long b = bsplit[th]; b < bsplit[th + 1]; b++) {
            const int o0 = bstart[b], o1 = bstart[b + 1];
            if (o0 == o1) continue;
            todo.clear(); todo_force.clear();
            todo.emplace_back(border.begin() + o0, border.begin() + o1);
            todo_force.push_back(0);
          while (!todo.empty()) {
            std::vector<int> own = std::move(todo.back());   // ascending node ids
            todo.pop_back();
            const bool force = todo_force.back();
            todo_force.pop_back();
            // halves of a set of owned nodes: across its longest extent (in grid spacings), at the middle of that extent moved
            // down by a millionth of a spacing -- the middle of a symmetric set is a plane of nodes, whose (lattice) coordinates
            // differ in the last bits -- so that a plane of the node grid stays in one half.  Any partition of the owned nodes is
            // a valid plan; this one needs a minimum, a maximum and one comparison per node (ebe_plan.hip does the same on the
            // device).  The later half is pushed first: tiles stay in box order.  false: the nodes (all but) coincide.
            auto split_own = [&]() -> bool {
                double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
                for (int v : own)
                    for (int d = 0; d < pd; d++) { mn[d] = std::min(mn[d], xyz[(size_t)v * pd + d]); mx[d] = std::max(mx[d], xyz[(size_t)v * pd + d]); }
                int dl = 0;
                for (int d = 1; d < pd; d++)
                    if ((mx[d] - mn[d]) / h[d] > (mx[dl] - mn[dl]) / h[dl]) dl = d;
                const double eps = 1e-6 * h[dl];
                if (!(mx[dl] - mn[dl] > 2.0 * eps)) return false;
                const double cut = 0.5 * (mn[dl] + mx[dl]) - eps;
                std::vector<int> lo_half, hi_half;
                for (int v : own) (xyz[(size_t)v * pd + dl] < cut ? lo_half : hi_half).push_back(v);
                todo.push_back(std::move(hi_half)); todo_force.push_back(0);
                todo.push_back(std::move(lo_half)); todo_force.push_back(0);
                return true;
            };
            if (force && own.size() >= 2 && split_own()) continue;
            const int t = o.ntiles++;
            // the elements that touch an owned node, then the tile's nodes: owned ones first, the others of those elements
            // behind them, each part ascending -- tile-sized tables only, nothing of the size of the mesh
            emap.clear(); nmap.clear();
            el.clear();
            const int nown = (int)own.size();
            bool huge = nown > 16000;   // (a box of a mesh the grid fits badly: split before the tables fill up)
            const size_t max_elems = (size_t)PLAN_MAX_ELEMS;   // (the device builder sorts a tile's elements in LDS)
            for (int k = 0; k < nown && !huge; k++) {
                const int v = own[k];
                for (int q = nptr[v]; q < nptr[v + 1]; q++) {
                    bool fresh;
                    const int e = (int)((nlst[q] & 0x3fffffffu) >> 3);
                    emap.slot(e, fresh);
                    if (fresh) el.push_back(e);
                }
                huge = el.size() > max_elems;
            }
            // the tile's elements in the order of the grid (z, y, x of their first node, in the box coordinates), whatever the
            // numbering of the mesh: x-neighbours become consecutive, so the chains and a handful of colours are found on a
            // mesh with shuffled element ids as on one numbered along the grid (where this is the ascending order of the ids)
            ekey.resize(el.size());
            for (size_t i = 0; i < el.size(); i++) {
                const size_t v = (size_t)conn[(size_t)el[i] * npe];
                unsigned long long k = 0;
                for (int d = pd - 1; d >= 0; d--) {
                    const long long gcoord = FEM_PLAN_GRID(xyz[v * pd + d], lo[d], h[d]) + 4;
                    k = (k << 21) | (unsigned long long)(std::min(std::max(gcoord, 0LL), (1LL << 21) - 1));
                }
                ekey[i] = {k, el[i]};
            }
            std::sort(ekey.begin(), ekey.end());
            for (size_t i = 0; i < el.size(); i++) el[i] = ekey[i].second;
            nd.assign(own.begin(), own.end());
            for (int k = 0; k < nown && !huge; k++) { bool fresh; *nmap.slot(own[k], fresh) = k; }
            for (size_t i = 0; i < el.size() && !huge; i++) {
                for (int a = 0; a < npe; a++) {
                    bool fresh;
                    const int v = conn[(size_t)el[i] * npe + a];
                    int* s = nmap.slot(v, fresh);
                    if (fresh) { *s = -1; nd.push_back(v); }
                }
                huge = nd.size() > 16000;
            }
            std::sort(nd.begin() + nown, nd.end());
            for (size_t i = (size_t)nown; i < nd.size() && !huge; i++) { bool fresh; *nmap.slot(nd[i], fresh) = (int)i; }
            auto local_of = [&](int v) { return nmap.get(v); };
            if (huge || (int)nd.size() > max_nodes || nown > max_own) {
                // too many nodes for the LDS image (a box of a distorted mesh may hold more than its share): halve the set
                // of owned nodes and try again
                if (nown < 2) { o.why = "tile with too many nodes"; o.why_a = (long long)nd.size(); o.why_b = nown; return; }
                o.ntiles--;
                if (!split_own()) { o.why = "coincident nodes"; o.why_a = nown; return; }
                continue;
            }
            lconn.resize(el.size() * (size_t)npe);
            for (size_t i = 0; i < el.size(); i++)
                for (int a = 0; a < npe; a++) lconn[i * npe + a] = local_of(conn[(size_t)el[i] * npe + a]);
            // chains: runs of elements (in that order) whose +x face is the next one's -x face, at most 16 long -- a chain
            // sits in consecutive lanes of one 16-lane row and passes its shared-face values along the row (k_ebe_tile)
            chain_start.clear();
            for (size_t i = 0; i < el.size(); i++) {
                bool linked = false;
                if (i > 0 && (int)(i - chain_start.back()) < 16) {
                    linked = true;
                    for (int k = 0; k < nfp; k++)
                        linked = linked && conn[(size_t)el[i - 1] * npe + c_plus[k]] == conn[(size_t)el[i] * npe + c_minus[k]];
                }
                if (!linked) chain_start.push_back((int)i);
            }
            chain_start.push_back((int)el.size());
            const int nchain = (int)chain_start.size() - 1;
            // greedy colouring of the chains (two chains of one colour share no node)
            cmask.assign(nd.size(), 0u);
            col.resize((size_t)nchain);
            int ncol = 0;
            bool all_aff = true, all_cart = true;
            for (int ch = 0; ch < nchain; ch++) {
                unsigned used = 0;
                for (int i = chain_start[ch]; i < chain_start[ch + 1]; i++)
                    for (int a = 0; a < npe; a++) used |= cmask[lconn[(size_t)i * npe + a]];
                int c = 0;
                while (c < TILE_MAXC && ((used >> c) & 1u)) c++;
                if (c >= TILE_MAXC) { o.why = "too many colours"; return; }
                col[ch] = c; ncol = std::max(ncol, c + 1);
                for (int i = chain_start[ch]; i < chain_start[ch + 1]; i++) {
                    for (int a = 0; a < npe; a++) cmask[lconn[(size_t)i * npe + a]] |= 1u << c;
                    all_aff = all_aff && aff[el[i]]; all_cart = all_cart && aff[el[i]] == 2;
                }
            }
            const int elem_off = (int)o.telems.size(), step_off = (int)o.steps.size(), tnodes_off = (int)o.tnodes.size();
            o.tnodes.insert(o.tnodes.end(), nd.begin(), nd.end());
            auto pad = [&] { o.telems.push_back(-1); o.tconn.push_back(make_uint4(0, 0, 0, 0)); o.tflags.push_back(0); };
            for (int c = 0; c < ncol; c++) {
                const int first = (int)o.telems.size() - elem_off;
                for (int ch = 0; ch < nchain; ch++) {
                    if (col[ch] != c) continue;
                    const int len = chain_start[ch + 1] - chain_start[ch];
                    const int used_in_row = ((int)o.telems.size() - elem_off - first) % 16;
                    if (used_in_row + len > 16)   // a chain does not straddle rows: pad the row
                        for (int p = used_in_row; p < 16; p++) pad();
                    for (int i = chain_start[ch]; i < chain_start[ch + 1]; i++) {
                        o.telems.push_back(el[i]);
                        unsigned l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        for (int a = 0; a < npe; a++) l[a] = (unsigned)lconn[(size_t)i * npe + a];
                        o.tconn.push_back(make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)));
                        o.tflags.push_back((unsigned char)(2 | (i + 1 < chain_start[ch + 1] ? 1 : 0)));   // bit 1: an element, bit 0: linked to the next lane
                    }
                }
                while (((int)o.telems.size() - elem_off - first) % 16) pad();
                const int cnt = (int)o.telems.size() - elem_off - first;
                for (int r = 0; r < cnt; r += 256) {
                    const int n = std::min(256, cnt - r);
                    o.steps.push_back(make_int2(first + r, (r + 256 >= cnt) ? (int)((unsigned)n | 0x80000000u) : n));
                }
            }
            o.visits += (long long)el.size();
            o.hdr.push_back(tnodes_off); o.hdr.push_back(nown); o.hdr.push_back((int)nd.size()); o.hdr.push_back(elem_off);
            o.hdr.push_back(step_off); o.hdr.push_back((int)o.steps.size() - step_off);
            o.max_nnode = std::max(o.max_nnode, (int)nd.size());
            (all_cart ? o.lcart : all_aff ? o.laff : o.lgen).push_back(t);
            o.taff.resize(o.telems.size(), all_cart ? 2 : all_aff ? 1 : 0);
            bool iface = false;
            if (!shared_node.empty())
                for (int k = 0; k < nown && !iface; k++) iface = shared_node[(size_t)nd[k]] != 0;
            o.tif.push_back(iface ? 1 : 0);
          }
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < NT; i++) th.emplace_back([&work, i] { unpin_this_thread(); work(i); });
        work(0);
        for (auto& x : th) x.join();
    }
    for (auto& o : outs)
        if (o.why) return refuse(o.why, o.why_a, o.why_b);
    if (verbose) { const auto tn = std::chrono::steady_clock::now(); std::fprintf(stderr, "[fem_amd] ebe tiles set-up: tiles %.0f ms\n", std::chrono::duration<double, std::milli>(tn - tl).count()); tl = tn; }
    // the workers' pieces go to the device side by side (tiles keep the order of the boxes); only the small per-tile
    // records are renumbered on the host
    auto plan = std::make_shared<EbeTilePlan>();
    std::vector<size_t> toff((size_t)NT + 1, 0), noff((size_t)NT + 1, 0), eoff((size_t)NT + 1, 0), soff((size_t)NT + 1, 0);
    long long visits = 0;
    int max_nnode = 0;
    for (int i = 0; i < NT; i++) {
        toff[i + 1] = toff[i] + (size_t)outs[i].ntiles; noff[i + 1] = noff[i] + outs[i].tnodes.size();
        eoff[i + 1] = eoff[i] + outs[i].telems.size(); soff[i + 1] = soff[i] + outs[i].steps.size();
        visits += outs[i].visits; max_nnode = std::max(max_nnode, outs[i].max_nnode);
    }
    const int ntiles = (int)toff[NT];
    if (eoff[NT] > 0x7fffffffULL || noff[NT] > 0x7fffffffULL) return refuse("too many lane slots");
    std::vector<int> hdr, laff, lgen, lcart;
    std::vector<unsigned char> tif;
    hdr.reserve((size_t)ntiles * 6);
    for (int i = 0; i < NT; i++) {
        const TileOut& o = outs[i];
        for (size_t k = 0; k < o.hdr.size(); k += 6) {
            hdr.push_back(o.hdr[k] + (int)noff[i]); hdr.push_back(o.hdr[k + 1]); hdr.push_back(o.hdr[k + 2]);
            hdr.push_back(o.hdr[k + 3] + (int)eoff[i]); hdr.push_back(o.hdr[k + 4] + (int)soff[i]); hdr.push_back(o.hdr[k + 5]);
            plan->max_nsteps = std::max(plan->max_nsteps, o.hdr[k + 5]);
            plan->max_nown = std::max(plan->max_nown, o.hdr[k + 1]);
        }
        for (int t : o.laff) laff.push_back(t + (int)toff[i]);
        for (int t : o.lcart) lcart.push_back(t + (int)toff[i]);
        for (int t : o.lgen) lgen.push_back(t + (int)toff[i]);
        tif.insert(tif.end(), o.tif.begin(), o.tif.end());
    }
    auto iface_first = [&](std::vector<int>& l) {
        return (int)(std::stable_partition(l.begin(), l.end(), [&](int t) { return tif[(size_t)t] != 0; }) - l.begin());
    };
    plan->n_aff_if = iface_first(laff); plan->n_gen_if = iface_first(lgen); plan->n_cart_if = iface_first(lcart);
    plan->halo_seen = m->halo.get();
    plan->ntiles = ntiles; plan->n_aff = (int)laff.size(); plan->n_gen = (int)lgen.size(); plan->n_cart = (int)lcart.size();
    plan->max_nnode = max_nnode;
    plan->tile_elements = visits; plan->slots = (long long)eoff[NT]; plan->tile_nodes = (long long)noff[NT];
    auto up = [&](auto& dbuf, const auto& vec) {
        dbuf.alloc(std::max<size_t>(vec.size(), 1));
        if (!vec.empty())
            FEM_HIP(hipMemcpyAsync(dbuf.p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice, s));
    };
    up(plan->hdr, hdr); up(plan->list_aff, laff); up(plan->list_gen, lgen); up(plan->list_cart, lcart);
    plan->steps.alloc(std::max<size_t>(soff[NT], 1)); plan->nodes.alloc(std::max<size_t>(noff[NT], 1));
    plan->elems.alloc(std::max<size_t>(eoff[NT], 1)); plan->conn16.alloc(std::max<size_t>(eoff[NT], 1));
    plan->flags.alloc(std::max<size_t>(eoff[NT], 1));
    DBuf<unsigned char> d_taff(std::max<size_t>(eoff[NT], 1));
    auto piece = [&](auto* dst, const auto& vec, size_t off) {
        if (!vec.empty())
            FEM_HIP(hipMemcpyAsync(dst + off, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice, s));
    };
    for (int i = 0; i < NT; i++) {
        const TileOut& o = outs[i];
        piece(plan->steps.p, o.steps, soff[i]); piece(plan->nodes.p, o.tnodes, noff[i]); piece(plan->elems.p, o.telems, eoff[i]);
        piece(plan->conn16.p, o.tconn, eoff[i]); piece(plan->flags.p, o.tflags, eoff[i]); piece(d_taff.p, o.taff, eoff[i]);
    }
    FEM_HIP(hipStreamSynchronize(s));
    ebe_tile_list_headers(*plan, !hex, s);
    if (!laff.empty() || !lcart.empty()) ebe_tile_store_geometry(m, *plan, d_taff.p, s);
    if (!hex) ebe_tile_quad_factors(m, *plan, s);
    FEM_HIP(hipStreamSynchronize(s));   // (d_taff goes out of scope)
    plan->t_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (verbose)
        std::fprintf(stderr, "[fem_amd] ebe tiles: %d (%d affine, %d Cartesian), <= %d nodes, %lld element visits in %lld lane slots for %d elements, %zu steps, set-up %.0f ms (%d threads)\n",
                     ntiles, plan->n_aff, plan->n_cart, max_nnode, plan->tile_elements, plan->slots, nel, soff[NT], plan->t_setup_ms, NT);
    return plan;
}

int ebe_tile_partials(const EbeTilePlan& P) { return P.ntiles; }
void ebe_tile_info(const EbeTilePlan& P, long long* tiles, long long* affine_tiles, long long* visits, long long* slots,
                   long long* nodes, double* ms)
{
    *tiles = P.ntiles; *affine_tiles = P.n_aff + P.n_cart; *visits = P.tile_elements; *slots = P.slots; *nodes = P.tile_nodes; *ms = P.t_setup_ms;
}

void ebe_tile_geometry_info(const EbeTilePlan& P, long long* uniform_tiles, int* per_tile_records, int* streaming_lists)
{
    if (uniform_tiles) *uniform_tiles = P.n_uniform;
    // 1: one geometry record per tile; 2: that, and every list of a tile at a fixed stride per list position with the slot records of
    // all (at most four) steps (k_ebe_tile_x4)
    const bool uni = P.n_cart + P.n_aff > 0 && P.n_uniform == (long long)P.n_cart + P.n_aff;
    if (per_tile_records) *per_tile_records = !uni ? 0 : (P.pnode_stride == TILE_X_MAX_NODES && P.pslot_steps == 4) ? 2 : 1;
    if (streaming_lists) *streaming_lists = P.stream ? 1 : 0;
}

static void hex_tile_kernel_attributes(int device)
{
    static OncePerDevice once;
    if (!once.first(device)) return;
    for (const void* f : {reinterpret_cast<const void*>(k_ebe_tile<0>), reinterpret_cast<const void*>(k_ebe_tile<1>),
                          reinterpret_cast<const void*>(k_ebe_tile<2>), reinterpret_cast<const void*>(k_ebe_tile<0, 1>),
                          reinterpret_cast<const void*>(k_ebe_tile<0, 2>), reinterpret_cast<const void*>(k_ebe_tile<0, 0, true>),
                          reinterpret_cast<const void*>(k_ebe_tile<1, 0, true>), reinterpret_cast<const void*>(k_ebe_tile<2, 0, true>)})
        FEM_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_MAX_NODES * 28));
}

// the plans whose product can open the CG iteration itself: quad4 with x of the tile in LDS (2-D: a tile's rim is 13 % of its nodes)
bool ebe_tile_can_fuse(fem_mat A, const EbeTilePlan& P)
{
    static const bool qx_env = !(std::getenv("FEM_AMD_TILE_X") && std::atoi(std::getenv("FEM_AMD_TILE_X")) == 0);
    return A->mesh->groups.size() == 1 && A->mesh->groups[0].type == FEM_QUA && qx_env && P.geo.p && P.max_nsteps <= 64 &&
           P.max_nnode <= TILE_Q4X_LDS_NODES && P.max_nown <= TILE_Q4X_LDS_OWN && P.n_gen > 0;
}

// KSPCG / PCJACOBI of a quad4 plan as one resident launch (k_cg_resident_q4): cg == nullptr asks whether this Mat's plan qualifies --
// the plans whose product can open the iteration (ebe_tile_can_fuse) with every tile resident at once; otherwise the kernel is
// enqueued on the context's stream (barrier words zeroed first) and the caller reads the verdict from the KspState and cg->bar.err.
bool ebe_tile_cg_resident(fem_mat A, const EbeTilePlan& P, const CgResidentArgs* cg)
{
    if (!ebe_tile_can_fuse(A, P) || P.n_gen != P.ntiles) return false;
    constexpr size_t shqx = (size_t)(4 * TILE_Q4X_LDS_NODES + 2 * TILE_Q4X_LDS_OWN) * sizeof(double);
    static std::mutex mu;
    static std::map<int, int> cap_of;   // device -> workgroups of the kernel that are resident at once
    int cap = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cap_of.find(A->ctx->device);
        if (it == cap_of.end()) {
            FEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cg_resident_q4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shqx));
            int per_cu = 0;
            FEM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cg_resident_q4, 256, shqx));
            // (the occupancy query can be one workgroup per CU high for kernels with many scalar registers, MI355X_MICROARCH.md: what the
            // LDS image and the launch bounds allow -- three -- is the figure trusted here, and a grid that turns out not to be resident
            // ends in the barrier's bounded spin, not in a hang)
            cap = std::min(per_cu, 3) * A->ctx->num_cu;
            cap_of[A->ctx->device] = cap;
        } else cap = it->second;
    }
    if (P.n_gen > cap || (size_t)A->mesh->nnds * 16 >= (1ull << 32)) return false;
    if (!cg) return true;
    fem_mesh m = A->mesh;
    hipStream_t s = A->ctx->stream;
    auto& grp = m->groups[0];
    const ElemTable& tab = host_elem_tables()[grp.type];
    const double g = std::fabs(tab.xi[0][0]), w = tab.w[0];
    const int* bcn = A->bc_saved ? A->bc_node.p : nullptr;
    FEM_HIP(hipMemsetAsync(cg->bar.w, 0, GridBarrier::WORDS * sizeof(unsigned), s));
    FEM_HIP(hipMemsetAsync(cg->bar.err, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_cg_resident_q4, dim3((P.n_gen + 7) / 8 * 8), dim3(256), shqx, s, P.n_gen, P.lhdr_gen.p, P.pnodes_gen.p, P.conn16.p,
                       P.flags.p, P.geo.p, (size_t)P.slots, m->coords.p, bcn, A->bc_dof.p, A->bc_shift.p, g, w, *cg,
                       TileFirst{P.psteps.p, P.pconn0.p, P.pfl0.p, P.pgeo.p, 0ull, (unsigned)P.pslot_steps});
    FEM_HIP(hipGetLastError());
    return true;
}

// y = K x, p.Kp partials into part[0 .. ntiles)
// phase -1: every tile; 0: the tiles that own nodes shared with other domains; 1: the others
// cg != nullptr (quad4 plans that pass ebe_tile_can_fuse, phase -1): x is not read -- the kernel forms p itself (cg_device.h)
void ebe_tile_mult(fem_mat A, const EbeTilePlan& P, const double* x, double* y, double* part, const int* done, int phase,
                   const CgFuseArgs* cg)
{
    fem_mesh m = A->mesh;
    hipStream_t s = A->ctx->stream;
    auto& grp = m->groups[0];
    const ElemTable& tab = host_elem_tables()[grp.type];
    const double g = std::fabs(tab.xi[0][0]), w = tab.w[0];
    const int* bcn = A->bc_saved ? A->bc_node.p : nullptr;
    // the caller wants the sum of the partials (A->red_out, set by the CG): the workgroup that finishes last delivers it
    unsigned* rcount = A->red_out ? A->red_counter.p : nullptr;
    if (grp.type == FEM_QUA) {
        const size_t shq = (size_t)P.max_nnode * (2 * sizeof(double) + sizeof(int));
        static OncePerDevice qonce;
        if (qonce.first(A->ctx->device))
            FEM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ebe_tile_q4), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        TILE_MAX_NODES * 20));
        const int n0 = phase == 1 ? P.n_gen_if : 0, n1 = phase == 0 ? P.n_gen_if : P.n_gen;
        // x and the coordinates of a tile in LDS where its nodes fit (FEM_AMD_TILE_X=0: the gathering kernel)
        static const bool qx_env = !(std::getenv("FEM_AMD_TILE_X") && std::atoi(std::getenv("FEM_AMD_TILE_X")) == 0);
        if (qx_env && P.geo.p && P.max_nsteps <= 64 && P.max_nnode <= TILE_Q4X_LDS_NODES && P.max_nown <= TILE_Q4X_LDS_OWN) {
            constexpr size_t shqx = (size_t)(4 * TILE_Q4X_LDS_NODES + 2 * TILE_Q4X_LDS_OWN) * sizeof(double);
            static OncePerDevice qxonce;
            if (qxonce.first(A->ctx->device))
                for (const void* f : {reinterpret_cast<const void*>(k_ebe_tile_q4x<false>), reinterpret_cast<const void*>(k_ebe_tile_q4x<true>)})
                    FEM_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shqx));
            if (n1 > n0) {
                auto launch = [&](auto kern, const CgFuseArgs& ca) {
                    hipLaunchKernelGGL(kern, dim3((n1 - n0 + 7) / 8 * 8), dim3(256), shqx, s, n1 - n0, P.lhdr_gen.p + 8 * (size_t)n0,
                                       P.steps.p, P.pnodes_gen.p + (size_t)n0 * TILE_Q4X_MAX_NODES, P.conn16.p, P.flags.p, P.geo.p, (size_t)P.slots, m->coords.p, x, bcn,
                                       A->bc_dof.p, A->bc_shift.p, g, w, y, part, done, rcount, P.ntiles, A->red_out, ca,
                                       TileFirst{P.psteps.p, P.pconn0.p, P.pfl0.p, P.pgeo.p, (unsigned long long)n0, (unsigned)P.pslot_steps});
                };
                if (cg) launch(k_ebe_tile_q4x<true>, *cg); else launch(k_ebe_tile_q4x<false>, CgFuseArgs{});
            }
            return;
        }
        FEM_CHECK(!cg, 7, "ebe_tile_mult: the fused CG opening needs the quad4 kernel with x in LDS");
        if (n1 > n0)
            hipLaunchKernelGGL(k_ebe_tile_q4, dim3((n1 - n0 + 7) / 8 * 8), dim3(256), shq, s, n1 - n0, P.lhdr_gen.p + 8 * (size_t)n0,
                               P.steps.p, P.nodes.p, P.elems.p, P.conn16.p, P.flags.p, m->coords.p, grp.E.p, grp.nu.p, x, bcn,
                               A->bc_dof.p, A->bc_shift.p, g, w, P.max_nnode, y, part, done, rcount, P.ntiles, A->red_out);
        return;
    }
    FEM_CHECK(!cg, 7, "ebe_tile_mult: the fused CG opening is a quad4 path");
    const size_t shmem = (size_t)P.max_nown * 3 * sizeof(double) + (size_t)P.max_nnode * sizeof(int);
    hex_tile_kernel_attributes(m->ctx->device);
    auto go = [&](auto kern, int n, const int* list) {
        if (!n) return;
        hipLaunchKernelGGL(kern, dim3((n + 7) / 8 * 8), dim3(256), shmem, s, n, list, P.steps.p, P.nodes.p, P.elems.p,
                           P.conn16.p, P.flags.p, m->coords.p, grp.E.p, grp.nu.p, x, bcn, A->bc_dof.p, A->bc_shift.p, P.geo.p,
                           (size_t)P.slots, g, w, P.max_nown, y, part, done, (const double*)nullptr, (size_t)0, rcount, P.ntiles, A->red_out);
    };
    const int c0 = phase == 1 ? P.n_cart_if : 0, c1 = phase == 0 ? P.n_cart_if : P.n_cart;
    const int a0 = phase == 1 ? P.n_aff_if : 0, a1 = phase == 0 ? P.n_aff_if : P.n_aff;
    const int g0 = phase == 1 ? P.n_gen_if : 0, g1 = phase == 0 ? P.n_gen_if : P.n_gen;
    // bricks: x of the tile in LDS where the plan's tiles are small enough for three workgroups per CU (FEM_AMD_TILE_X=0: never)
    static const bool x_env = !(std::getenv("FEM_AMD_TILE_X") && std::atoi(std::getenv("FEM_AMD_TILE_X")) == 0);
    constexpr size_t shx = (size_t)(TILE_X_LDS_NODES + TILE_X_MAX_OWN) * 3 * sizeof(double), shx_max = shx;
    if (x_env && P.max_nsteps <= 64 && P.max_nnode <= TILE_X_LDS_NODES && P.max_nown <= TILE_X_MAX_OWN) {
        static OncePerDevice xonce;
        if (xonce.first(A->ctx->device))
            for (const void* f : {reinterpret_cast<const void*>(k_ebe_tile_x<1>), reinterpret_cast<const void*>(k_ebe_tile_x<2>),
                                  reinterpret_cast<const void*>(k_ebe_tile_x<1, true>), reinterpret_cast<const void*>(k_ebe_tile_x<2, true>),
                                  reinterpret_cast<const void*>(k_ebe_tile_x<1, false, true>), reinterpret_cast<const void*>(k_ebe_tile_x<2, false, true>),
                                  reinterpret_cast<const void*>(k_ebe_tile_x<1, true, true>), reinterpret_cast<const void*>(k_ebe_tile_x<2, true, true>),
                                  reinterpret_cast<const void*>(k_ebe_tile_x4<1, false>), reinterpret_cast<const void*>(k_ebe_tile_x4<2, false>),
                                  reinterpret_cast<const void*>(k_ebe_tile_x4<1, true>), reinterpret_cast<const void*>(k_ebe_tile_x4<2, true>)})
                FEM_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shx_max));
        auto gox = [&](auto kern, int n, const int* list, const int* pnodes, size_t gp0) {
            if (!n) return;
            hipLaunchKernelGGL(kern, dim3((n + 7) / 8 * 8), dim3(256), shx, s, n, list, P.steps.p, pnodes, P.conn16.p, P.flags.p,
                               x, bcn, A->bc_dof.p, A->bc_shift.p, P.geo.p, (size_t)P.slots, g, P.max_nnode, P.max_nown, y, part, done, rcount,
                               P.ntiles, A->red_out, TileFirst{P.psteps.p, P.pconn0.p, P.pfl0.p, P.pgeo.p, (unsigned long long)gp0, (unsigned)P.pslot_steps});
        };
        // (a run-time test of the header word inside one kernel cost more than the records: 0.131 against 0.124 ms; so the variant is
        // per plan -- a mesh with a graded region or a second material keeps the records slot by slot)
        const bool uni = P.n_uniform == (long long)P.n_cart + P.n_aff;
        if (uni && P.pslot_steps == 4) {   // (every slot record of a tile at the fixed stride: none is loaded inside the steps)
            if (P.stream) {
                gox(k_ebe_tile_x4<2, true>, c1 - c0, P.lhdr_cart.p + 8 * (size_t)c0, P.pnodes_cart.p + (size_t)c0 * TILE_X_MAX_NODES, (size_t)c0);
                gox(k_ebe_tile_x4<1, true>, a1 - a0, P.lhdr_aff.p + 8 * (size_t)a0, P.pnodes_aff.p + (size_t)a0 * TILE_X_MAX_NODES, (size_t)P.n_cart + a0);
            } else {
                gox(k_ebe_tile_x4<2, false>, c1 - c0, P.lhdr_cart.p + 8 * (size_t)c0, P.pnodes_cart.p + (size_t)c0 * TILE_X_MAX_NODES, (size_t)c0);
                gox(k_ebe_tile_x4<1, false>, a1 - a0, P.lhdr_aff.p + 8 * (size_t)a0, P.pnodes_aff.p + (size_t)a0 * TILE_X_MAX_NODES, (size_t)P.n_cart + a0);
            }
        } else if (P.stream && uni) {
            gox(k_ebe_tile_x<2, true, true>, c1 - c0, P.lhdr_cart.p + 8 * (size_t)c0, P.pnodes_cart.p + (size_t)c0 * TILE_X_MAX_NODES, (size_t)c0);
            gox(k_ebe_tile_x<1, true, true>, a1 - a0, P.lhdr_aff.p + 8 * (size_t)a0, P.pnodes_aff.p + (size_t)a0 * TILE_X_MAX_NODES, (size_t)P.n_cart + a0);
        } else if (P.stream) {
            gox(k_ebe_tile_x<2, true>, c1 - c0, P.lhdr_cart.p + 8 * (size_t)c0, P.pnodes_cart.p + (size_t)c0 * TILE_X_MAX_NODES, (size_t)c0);
            gox(k_ebe_tile_x<1, true>, a1 - a0, P.lhdr_aff.p + 8 * (size_t)a0, P.pnodes_aff.p + (size_t)a0 * TILE_X_MAX_NODES, (size_t)P.n_cart + a0);
        } else if (uni) {
            gox(k_ebe_tile_x<2, false, true>, c1 - c0, P.lhdr_cart.p + 8 * (size_t)c0, P.pnodes_cart.p + (size_t)c0 * TILE_X_MAX_NODES, (size_t)c0);
            gox(k_ebe_tile_x<1, false, true>, a1 - a0, P.lhdr_aff.p + 8 * (size_t)a0, P.pnodes_aff.p + (size_t)a0 * TILE_X_MAX_NODES, (size_t)P.n_cart + a0);
        } else {
            gox(k_ebe_tile_x<2>, c1 - c0, P.lhdr_cart.p + 8 * (size_t)c0, P.pnodes_cart.p + (size_t)c0 * TILE_X_MAX_NODES, (size_t)c0);
            gox(k_ebe_tile_x<1>, a1 - a0, P.lhdr_aff.p + 8 * (size_t)a0, P.pnodes_aff.p + (size_t)a0 * TILE_X_MAX_NODES, (size_t)P.n_cart + a0);
        }
    } else if (P.stream) {
        go(k_ebe_tile<2, 0, true>, c1 - c0, P.lhdr_cart.p + 8 * (size_t)c0);
        go(k_ebe_tile<1, 0, true>, a1 - a0, P.lhdr_aff.p + 8 * (size_t)a0);
    } else {
        go(k_ebe_tile<2>, c1 - c0, P.lhdr_cart.p + 8 * (size_t)c0);
        go(k_ebe_tile<1>, a1 - a0, P.lhdr_aff.p + 8 * (size_t)a0);
    }
    if (P.stream) go(k_ebe_tile<0, 0, true>, g1 - g0, P.lhdr_gen.p + 8 * (size_t)g0);
    else go(k_ebe_tile<0>, g1 - g0, P.lhdr_gen.p + 8 * (size_t)g0);
}
// stress_node(nnds, 6) of a hex8 mesh over the tiles of the product (RecoverNodalStress per element, the sum over the
// elements of a node src/main.F90:258-268 in LDS, the division by the valence :277-279 on the way out): two passes of
// three components, every tile through the kernel of the distorted elements (it reads the coordinates)
bool ebe_tile_stress(fem_mesh m, const EbeTilePlan& P, const double* U, double* S, const double* valence)
{
    auto& grp = m->groups[0];
    if (m->groups.size() != 1 || grp.type != FEM_HEX) return false;
    hipStream_t s = m->ctx->stream;
    const ElemTable& tab = host_elem_tables()[grp.type];
    const double g = std::fabs(tab.xi[0][0]), w = tab.w[0];
    const size_t shmem = (size_t)P.max_nown * 3 * sizeof(double) + (size_t)P.max_nnode * sizeof(int), nn = (size_t)m->nnds;
    hex_tile_kernel_attributes(m->ctx->device);
    auto go = [&](auto kern, int n, const int* list, double* out) {
        if (!n) return;
        hipLaunchKernelGGL(kern, dim3((n + 7) / 8 * 8), dim3(256), shmem, s, n, list, P.steps.p, P.nodes.p, P.elems.p,
                           P.conn16.p, P.flags.p, m->coords.p, grp.E.p, grp.nu.p, U, (const int*)nullptr, (const int*)nullptr,
                           (const double*)nullptr, P.geo.p, (size_t)P.slots, g, w, P.max_nown, out, (double*)nullptr,
                           (const int*)nullptr, valence, nn, (unsigned*)nullptr, 0, (double*)nullptr);
    };
    // tiles of parallelepipeds: one pass, from the stored geometry (FEM_AMD_STRESS_CART=0: the two passes below for them too)
    static const bool cart_env = !(std::getenv("FEM_AMD_STRESS_CART") && std::atoi(std::getenv("FEM_AMD_STRESS_CART")) == 0);
    const bool cart_one_pass = cart_env && (P.n_cart > 0 || P.n_aff > 0) && P.max_nsteps <= 64 && P.max_nnode <= TILE_X_LDS_NODES &&
                               P.max_nown <= TILE_X_MAX_OWN && P.pnode_stride == TILE_X_MAX_NODES;
    if (cart_one_pass) {
        constexpr size_t shs = (size_t)(3 * TILE_X_LDS_NODES + 6 * TILE_X_MAX_OWN) * sizeof(double);
        static OncePerDevice sonce;
        if (sonce.first(m->ctx->device))
            for (const void* f : {reinterpret_cast<const void*>(k_tile_stress_cart<1, false>), reinterpret_cast<const void*>(k_tile_stress_cart<2, false>),
                                  reinterpret_cast<const void*>(k_tile_stress_cart<1, true>), reinterpret_cast<const void*>(k_tile_stress_cart<2, true>),
                                  reinterpret_cast<const void*>(k_tile_stress_cart<1, true, true>), reinterpret_cast<const void*>(k_tile_stress_cart<2, true, true>)})
                FEM_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shs));
        const bool uni = P.n_uniform == (long long)P.n_cart + P.n_aff;    // (as the product: every tile has one record)
        auto launch = [&](auto kern, int n, const int* list, const int* pnodes, size_t gp0) {
            hipLaunchKernelGGL(kern, dim3((n + 7) / 8 * 8), dim3(256), shs, s, n, list, P.steps.p, P.nodes.p, P.elems.p, P.conn16.p, P.flags.p, U,
                               P.geo.p, (size_t)P.slots, grp.E.p, grp.nu.p, w, S, valence, nn, pnodes,
                               TileFirst{P.psteps.p, P.pconn0.p, P.pfl0.p, P.pgeo.p, (unsigned long long)gp0, (unsigned)P.pslot_steps});
        };
        if (uni && P.pslot_steps == 4) {
            if (P.n_cart) launch(k_tile_stress_cart<2, true, true>, P.n_cart, P.lhdr_cart.p, P.pnodes_cart.p, 0);
            if (P.n_aff) launch(k_tile_stress_cart<1, true, true>, P.n_aff, P.lhdr_aff.p, P.pnodes_aff.p, (size_t)P.n_cart);
        } else {
            if (P.n_cart) { if (uni) launch(k_tile_stress_cart<2, true>, P.n_cart, P.lhdr_cart.p, P.pnodes_cart.p, 0); else launch(k_tile_stress_cart<2, false>, P.n_cart, P.lhdr_cart.p, P.pnodes_cart.p, 0); }
            if (P.n_aff) { if (uni) launch(k_tile_stress_cart<1, true>, P.n_aff, P.lhdr_aff.p, P.pnodes_aff.p, (size_t)P.n_cart); else launch(k_tile_stress_cart<1, false>, P.n_aff, P.lhdr_aff.p, P.pnodes_aff.p, (size_t)P.n_cart); }
        }
    }
    for (int half = 0; half < 2; half++) {
        double* out = S + (size_t)(3 * half) * nn;
        for (auto [n, list] : {std::pair<int, const int*>{cart_one_pass ? 0 : P.n_cart, P.lhdr_cart.p}, {cart_one_pass ? 0 : P.n_aff, P.lhdr_aff.p}, {P.n_gen, P.lhdr_gen.p}}) {
            if (half == 0) go(k_ebe_tile<0, 1>, n, list, out);
            else go(k_ebe_tile<0, 2>, n, list, out);
        }
    }
    return true;
}
bool ebe_tile_split_valid(fem_mat A, const EbeTilePlan& P) { return P.halo_seen == (const void*)A->mesh->halo.get(); }

// fem_mesh_tile_plan_digest: the plan built afresh by the named builder, hashed array by array (cross-check of the builders)
void mesh_tile_plan_digest_impl(fem_mesh m, const char* builder, unsigned long long* digest, double* setup_ms)
{
    m->ctx->bind();
    hipStream_t s = m->ctx->stream;
    FEM_CHECK(m->pattern, 5, "fem_mesh_tile_plan_digest: no Mat was created from this mesh yet (the symbolic phase supplies the node lists)");
    const std::string b = builder;
    FEM_CHECK(b == "device" || b == "host", 6, "fem_mesh_tile_plan_digest: builder must be device|host");
    std::shared_ptr<EbeTilePlan> P = b == "host" ? ebe_tile_build_host(m, s) : ebe_tile_build_device(m, s);
    for (int i = 0; i < 12; i++) digest[i] = 0;
    if (setup_ms) *setup_ms = P ? P->t_setup_ms : 0.0;
    if (!P) return;
    auto fnv = [&](const void* dptr, size_t bytes, unsigned long long h = 1469598103934665603ULL) {
        std::vector<unsigned char> buf(bytes);
        if (bytes) FEM_HIP(hipMemcpy(buf.data(), dptr, bytes, hipMemcpyDeviceToHost));
        for (unsigned char c : buf) { h ^= c; h *= 1099511628211ULL; }
        return h;
    };
    FEM_HIP(hipStreamSynchronize(s));
    digest[0] = (unsigned long long)P->ntiles; digest[1] = (unsigned long long)P->slots;
    digest[2] = (unsigned long long)P->tile_nodes; digest[3] = (unsigned long long)P->tile_elements;
    digest[4] = fnv(P->hdr.p, (size_t)P->ntiles * 6 * sizeof(int));
    // (the number of steps is the sum of the headers' sixth entries: take it from there)
    std::vector<int> hdr((size_t)P->ntiles * 6);
    if (P->ntiles) FEM_HIP(hipMemcpy(hdr.data(), P->hdr.p, hdr.size() * sizeof(int), hipMemcpyDeviceToHost));
    size_t nsteps = 0;
    for (int t = 0; t < P->ntiles; t++) nsteps += (size_t)hdr[(size_t)6 * t + 5];
    digest[5] = fnv(P->steps.p, nsteps * sizeof(int2));
    digest[6] = fnv(P->nodes.p, (size_t)P->tile_nodes * sizeof(int));
    digest[7] = fnv(P->elems.p, (size_t)P->slots * sizeof(int));
    digest[8] = fnv(P->conn16.p, (size_t)P->slots * sizeof(uint4));
    digest[9] = fnv(P->flags.p, (size_t)P->slots);
    const int counts[9] = {P->n_cart, P->n_cart_if, P->n_aff, P->n_aff_if, P->n_gen, P->n_gen_if, P->max_nnode, P->max_nsteps, P->max_nown};
    unsigned long long h = 1469598103934665603ULL;
    for (int c : counts) { h ^= (unsigned long long)(unsigned)c; h *= 1099511628211ULL; }
    h = fnv(P->list_cart.p, (size_t)P->n_cart * sizeof(int), h);
    h = fnv(P->list_aff.p, (size_t)P->n_aff * sizeof(int), h);
    digest[10] = fnv(P->list_gen.p, (size_t)P->n_gen * sizeof(int), h);
    // stored geometry: only the entries of real elements in stored-geometry tiles are ever written
    if (P->geo.p && (P->n_aff || P->n_cart)) {
        std::vector<double> geo((size_t)P->slots * (P->n_aff ? 11 : 5));
        std::vector<int> el((size_t)P->slots);
        FEM_HIP(hipMemcpy(geo.data(), P->geo.p, geo.size() * sizeof(double), hipMemcpyDeviceToHost));
        FEM_HIP(hipMemcpy(el.data(), P->elems.p, el.size() * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> lists;
        std::vector<unsigned char> cls((size_t)P->ntiles, 0);
        auto mark = [&](const DBuf<int>& l, int n, unsigned char c) {
            std::vector<int> hl((size_t)n);
            if (n) FEM_HIP(hipMemcpy(hl.data(), l.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
            for (int t : hl) cls[(size_t)t] = c;
        };
        mark(P->list_aff, P->n_aff, 1); mark(P->list_cart, P->n_cart, 2);
        unsigned long long g = 1469598103934665603ULL;
        for (int t = 0; t < P->ntiles; t++) {
            if (!cls[(size_t)t]) continue;
            const int e0 = hdr[(size_t)6 * t + 3], e1 = (t + 1 < P->ntiles) ? hdr[(size_t)6 * (t + 1) + 3] : (int)P->slots;
            const int ng = cls[(size_t)t] == 2 ? 5 : 11;
            for (int k = e0; k < e1; k++) {
                if (el[(size_t)k] < 0) continue;
                for (int j = 0; j < ng; j++) {
                    unsigned long long bits;
                    std::memcpy(&bits, &geo[(size_t)j * (size_t)P->slots + (size_t)k], 8);
                    g ^= bits; g *= 1099511628211ULL;
                }
            }
        }
        digest[11] = g;
    }
}
