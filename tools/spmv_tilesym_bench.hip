// tools/spmv_tilesym_bench.hip -- design-time experiment (not product code).
// Tile-local symmetric SpMV: rows are grouped into spatially compact tiles of
// 256 nodes (one workgroup); a block K_ij whose two rows live in the SAME tile
// is stored once (with the lower-indexed row) and applied both ways -- the
// transposed product goes to row j through an LDS staging buffer in a fixed
// order (deterministic, no atomics).  Blocks that cross a tile boundary are
// stored in both rows as today.  Vectors keep the external numbering: only the
// matrix rows are permuted (rowmap), x is gathered and y scattered by node id.
//
//   hipcc -O3 --offload-arch=gfx950 tools/spmv_tilesym_bench.hip -o tools/spmv_tilesym_bench
//   tools/spmv_tilesym_bench [n=150] [reps=20] [TX TY TZ]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static inline double hash01(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (double)(k >> 11) / 9007199254740992.0 - 0.5;
}
static inline double aval(int64_t i, int64_t j, int r, int c) {  // A[i][j][r][c] == A[j][i][c][r]
    if (i > j || (i == j && r > c)) { std::swap(i, j); std::swap(r, c); }
    return hash01((uint64_t)i * 1000003ULL * 9 + (uint64_t)(j - i + 40000) * 9 + r * 3 + c + 12345);
}

constexpr int TILE = 256;  // rows per tile = threads per workgroup (4 slices of 64)

// goff/soff: per slice offsets (in block columns) of the general / symmetric parts.
//   gcol[(goff[s]+k)*64+lane]  gval[((goff[s]+k)*9+e)*64+lane]
//   scol, sval likewise; sdst[(soff[s]+k)*64+lane] = staging slot of the transposed product (0xffff = none)
// inc[r], inc[r+1] (per internal row, tile-local offsets): the row's incoming staging range.
__global__ void __launch_bounds__(TILE) k_tilesym(int ntiles, const int* __restrict__ rowmap,
                                                  const int* __restrict__ goff, const int* __restrict__ gcol, const double* __restrict__ gval,
                                                  const int* __restrict__ soff, const int* __restrict__ scol, const unsigned short* __restrict__ sdst,
                                                  const double* __restrict__ sval, const unsigned short* __restrict__ inc,
                                                  const double* __restrict__ x, double* __restrict__ y)
{
    extern __shared__ double stage[];
    const int t = blockIdx.x, l = threadIdx.x, lane = l & 63;
    const int s = t * (TILE / 64) + (l >> 6);
    const int r = t * TILE + l;
    const int ext = rowmap[r];
    double y0 = 0, y1 = 0, y2 = 0;
    double xi0 = 0, xi1 = 0, xi2 = 0;
    if (ext >= 0) { xi0 = x[3 * (size_t)ext]; xi1 = x[3 * (size_t)ext + 1]; xi2 = x[3 * (size_t)ext + 2]; }
    {
        const int o0 = goff[s], W = goff[s + 1] - o0;
        const int* bc = gcol + (size_t)o0 * 64 + lane;
        const double* v = gval + (size_t)o0 * 9 * 64 + lane;
        for (int k = 0; k < W; k++) {
            const int c = bc[(size_t)k * 64];
            const double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
            const double* a = v + (size_t)k * 9 * 64;
            y0 += a[0] * x0 + a[64] * x1 + a[128] * x2;
            y1 += a[192] * x0 + a[256] * x1 + a[320] * x2;
            y2 += a[384] * x0 + a[448] * x1 + a[512] * x2;
        }
    }
    {
        const int o0 = soff[s], W = soff[s + 1] - o0;
        const int* bc = scol + (size_t)o0 * 64 + lane;
        const unsigned short* bd = sdst + (size_t)o0 * 64 + lane;
        const double* v = sval + (size_t)o0 * 9 * 64 + lane;
        for (int k = 0; k < W; k++) {
            const int c = bc[(size_t)k * 64];
            const unsigned d = bd[(size_t)k * 64];
            const double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
            const double* a = v + (size_t)k * 9 * 64;
            const double a0 = a[0], a1 = a[64], a2 = a[128], a3 = a[192], a4 = a[256], a5 = a[320], a6 = a[384], a7 = a[448], a8 = a[512];
            y0 += a0 * x0 + a1 * x1 + a2 * x2;
            y1 += a3 * x0 + a4 * x1 + a5 * x2;
            y2 += a6 * x0 + a7 * x1 + a8 * x2;
            if (d != 0xffffu) {   // transposed product for the partner row, delivered through LDS
                stage[3 * d] = a0 * xi0 + a3 * xi1 + a6 * xi2;
                stage[3 * d + 1] = a1 * xi0 + a4 * xi1 + a7 * xi2;
                stage[3 * d + 2] = a2 * xi0 + a5 * xi1 + a8 * xi2;
            }
        }
    }
    __syncthreads();
    const int q0 = inc[(size_t)t * (TILE + 1) + l], q1 = inc[(size_t)t * (TILE + 1) + l + 1];
    for (int q = q0; q < q1; q++) { y0 += stage[3 * q]; y1 += stage[3 * q + 1]; y2 += stage[3 * q + 2]; }
    if (ext >= 0) { y[3 * (size_t)ext] = y0; y[3 * (size_t)ext + 1] = y1; y[3 * (size_t)ext + 2] = y2; }
}

// baseline: full sliced ELL in natural order
__global__ void __launch_bounds__(256) k_full(int nslices, int nrows, const int* __restrict__ boff, const int* __restrict__ bcol,
                                              const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, s = blockIdx.x * 4 + wave;
    if (s >= nslices) return;
    int o0 = boff[s], W = boff[s + 1] - o0;
    const int* bc = bcol + (size_t)o0 * 64 + lane;
    const double* v = val + (size_t)o0 * 9 * 64 + lane;
    double y0 = 0, y1 = 0, y2 = 0;
    for (int j = 0; j < W; j++) {
        int c = bc[(size_t)j * 64];
        double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
        const double* a = v + (size_t)j * 9 * 64;
        y0 += a[0] * x0 + a[64] * x1 + a[128] * x2;
        y1 += a[192] * x0 + a[256] * x1 + a[320] * x2;
        y2 += a[384] * x0 + a[448] * x1 + a[512] * x2;
    }
    int row = s * 64 + lane;
    if (row < nrows) { y[3 * (size_t)row] = y0; y[3 * (size_t)row + 1] = y1; y[3 * (size_t)row + 2] = y2; }
}

template <typename F>
static double time_ms(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 150, reps = argc > 2 ? atoi(argv[2]) : 20;
    int TX = argc > 5 ? atoi(argv[3]) : 8, TY = argc > 5 ? atoi(argv[4]) : 8, TZ = argc > 5 ? atoi(argv[5]) : 4;
    if (TX * TY * TZ != TILE) { printf("tile must have %d nodes\n", TILE); return 1; }
    size_t nn = (size_t)n * n * n, ndof = 3 * nn;
    printf("nodes/edge %d nodes %zu dof %zu  tile %dx%dx%d\n", n, nn, ndof, TX, TY, TZ); fflush(stdout);
    auto nid = [&](int i, int j, int k) { return (size_t)i + (size_t)n * (j + (size_t)n * k); };
    std::vector<double> x(ndof);
    for (size_t i = 0; i < ndof; i++) x[i] = hash01(i + 0x9e3779b97f4a7c15ULL);
    auto neighbours = [&](size_t v, std::vector<int>& out) {
        out.clear();
        int i = (int)(v % n), j = (int)((v / n) % n), k = (int)(v / ((size_t)n * n));
        for (int dk = -1; dk <= 1; dk++) for (int dj = -1; dj <= 1; dj++) for (int di = -1; di <= 1; di++) {
            int ii = i + di, jj = j + dj, kk = k + dk;
            if (ii < 0 || jj < 0 || kk < 0 || ii >= n || jj >= n || kk >= n) continue;
            out.push_back((int)nid(ii, jj, kk));
        }
    };
    std::vector<int> nb;
    auto yref = [&](size_t v, int r) { std::vector<int> q; neighbours(v, q); double s = 0; for (int u : q) for (int c = 0; c < 3; c++) s += aval(v, u, r, c) * x[3 * (size_t)u + c]; return s; };
    double *dx, *dy; CK(hipMalloc(&dx, ndof * 8)); CK(hipMalloc(&dy, ndof * 8));
    CK(hipMemcpy(dx, x.data(), ndof * 8, hipMemcpyHostToDevice));
    std::vector<double> yh(ndof);
    auto check = [&](const char* name) {
        CK(hipMemcpy(yh.data(), dy, ndof * 8, hipMemcpyDeviceToHost));
        double mx = 0;
        for (size_t t = 0; t < 4000; t++) { size_t v = (t * 7919 * 131) % nn; for (int r = 0; r < 3; r++) mx = std::max(mx, fabs(yh[3 * v + r] - yref(v, r))); }
        for (size_t v : {(size_t)0, (size_t)63, (size_t)64, nn - 1, nid(7, 7, 3), nid(8, 8, 4), nid(n - 1, 0, n - 1)}) for (int r = 0; r < 3; r++) mx = std::max(mx, fabs(yh[3 * v + r] - yref(v, r)));
        printf("   check %-14s max abs err %.3e\n", name, mx); fflush(stdout);
        CK(hipMemset(dy, 0, ndof * 8));
    };
    double csr_bytes = 0;
    // ---------------- baseline ----------------
    {
        int nsl = (int)((nn + 63) / 64);
        std::vector<int> boff(nsl + 1, 0);
        std::vector<int> rl(nn);
        size_t nblocks = 0;
        for (size_t v = 0; v < nn; v++) { neighbours(v, nb); rl[v] = (int)nb.size(); nblocks += nb.size(); }
        csr_bytes = (double)nblocks * 9 * 12 + (ndof + 1) * 4.0 + ndof * 16.0;
        for (int s = 0; s < nsl; s++) { int W = 0; for (size_t v = (size_t)s * 64; v < std::min(nn, (size_t)(s + 1) * 64); v++) W = std::max(W, rl[v]); boff[s + 1] = boff[s] + W; }
        size_t tb = (size_t)boff[nsl];
        std::vector<int> sc(tb * 64, 0); std::vector<double> sv(tb * 64 * 9, 0.0);
        for (size_t v = 0; v < nn; v++) {
            neighbours(v, nb);
            int s = (int)(v >> 6), l = (int)(v & 63), W = boff[s + 1] - boff[s];
            for (int j = 0; j < W; j++) {
                size_t o = (size_t)boff[s] + j;
                bool has = j < (int)nb.size();
                sc[o * 64 + l] = has ? nb[j] : (int)v;
                if (has) for (int e = 0; e < 9; e++) sv[(o * 9 + e) * 64 + l] = aval(v, nb[j], e / 3, e % 3);
            }
        }
        int *dboff, *dsc; double* dsv;
        CK(hipMalloc(&dboff, (nsl + 1) * 4)); CK(hipMalloc(&dsc, sc.size() * 4)); CK(hipMalloc(&dsv, sv.size() * 8));
        CK(hipMemcpy(dboff, boff.data(), (nsl + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsv, sv.data(), sv.size() * 8, hipMemcpyHostToDevice));
        double fmt = tb * 64 * 76.0 + ndof * 16.0;
        double ms = time_ms([&] { k_full<<<(nsl + 3) / 4, 256>>>(nsl, (int)nn, dboff, dsc, dsv, dx, dy); }, reps);
        printf("%-22s %.3f ms  stored %.2f GB -> %.2f TB/s  csr-alg %.2f TB/s\n", "full sliced ELL", ms, fmt / 1e9, fmt / ms / 1e9, csr_bytes / ms / 1e9);
        check("full");
        CK(hipFree(dboff)); CK(hipFree(dsc)); CK(hipFree(dsv));
    }
    // ---------------- tile-symmetric ----------------
    {
        int ntx = (n + TX - 1) / TX, nty = (n + TY - 1) / TY, ntz = (n + TZ - 1) / TZ;
        int ntiles = ntx * nty * ntz;
        size_t nrows_i = (size_t)ntiles * TILE;
        std::vector<int> rowmap(nrows_i, -1), irow(nn);
        for (int tz = 0; tz < ntz; tz++) for (int ty = 0; ty < nty; ty++) for (int tx = 0; tx < ntx; tx++) {
            size_t t = tx + (size_t)ntx * (ty + (size_t)nty * tz);
            for (int lz = 0; lz < TZ; lz++) for (int ly = 0; ly < TY; ly++) for (int lx = 0; lx < TX; lx++) {
                int i = tx * TX + lx, j = ty * TY + ly, k = tz * TZ + lz;
                if (i >= n || j >= n || k >= n) continue;
                size_t r = t * TILE + lx + TX * (ly + TY * lz);
                rowmap[r] = (int)nid(i, j, k); irow[nid(i, j, k)] = (int)r;
            }
        }
        int nsl = (int)(nrows_i / 64);
        // per internal row: general list G, symmetric list S (in-tile partner with larger internal index)
        std::vector<int> goff(nsl + 1, 0), soff(nsl + 1, 0);
        std::vector<int> gcnt(nrows_i, 0), scnt(nrows_i, 0), icnt(nrows_i, 0);
        for (size_t r = 0; r < nrows_i; r++) {
            if (rowmap[r] < 0) continue;
            neighbours(rowmap[r], nb);
            for (int u : nb) {
                size_t ru = irow[u];
                if (ru / TILE == r / TILE && ru != r) { if (ru > r) { scnt[r]++; icnt[ru]++; } }
                else gcnt[r]++;
            }
        }
        for (int s = 0; s < nsl; s++) {
            int WG = 0, WS = 0;
            for (int l = 0; l < 64; l++) { WG = std::max(WG, gcnt[(size_t)s * 64 + l]); WS = std::max(WS, scnt[(size_t)s * 64 + l]); }
            goff[s + 1] = goff[s] + WG; soff[s + 1] = soff[s] + WS;
        }
        size_t tg = (size_t)goff[nsl], ts = (size_t)soff[nsl];
        std::vector<int> gcol(tg * 64, 0), scol(ts * 64, 0);
        std::vector<unsigned short> sdst(ts * 64, 0xffff), inc((size_t)ntiles * (TILE + 1), 0);
        std::vector<double> gval(tg * 64 * 9, 0.0), sval(ts * 64 * 9, 0.0);
        size_t max_stage = 0, true_blocks = 0;
        std::vector<int> fill(TILE);
        for (int t = 0; t < ntiles; t++) {
            int acc = 0;
            for (int l = 0; l < TILE; l++) { inc[(size_t)t * (TILE + 1) + l] = (unsigned short)acc; fill[l] = acc; acc += icnt[(size_t)t * TILE + l]; }
            inc[(size_t)t * (TILE + 1) + TILE] = (unsigned short)acc;
            max_stage = std::max(max_stage, (size_t)acc);
            for (int l = 0; l < TILE; l++) {
                size_t r = (size_t)t * TILE + l;
                int s = (int)(r >> 6), lane = (int)(r & 63);
                if (rowmap[r] < 0) {
                    for (int k = 0; k < goff[s + 1] - goff[s]; k++) gcol[((size_t)goff[s] + k) * 64 + lane] = 0;
                    for (int k = 0; k < soff[s + 1] - soff[s]; k++) scol[((size_t)soff[s] + k) * 64 + lane] = 0;
                    continue;
                }
                neighbours(rowmap[r], nb);
                int kg = 0, ks = 0;
                for (int u : nb) {
                    size_t ru = irow[u];
                    bool intile = (ru / TILE == r / TILE && ru != r);
                    if (intile && ru < r) continue;   // stored with the partner
                    if (intile) {
                        size_t o = (size_t)soff[s] + ks;
                        scol[o * 64 + lane] = u;
                        sdst[o * 64 + lane] = (unsigned short)(fill[ru % TILE]++);
                        for (int e = 0; e < 9; e++) sval[(o * 9 + e) * 64 + lane] = aval(rowmap[r], u, e / 3, e % 3);
                        ks++;
                    } else {
                        size_t o = (size_t)goff[s] + kg;
                        gcol[o * 64 + lane] = u;
                        for (int e = 0; e < 9; e++) gval[(o * 9 + e) * 64 + lane] = aval(rowmap[r], u, e / 3, e % 3);
                        kg++;
                    }
                    true_blocks++;
                }
                for (; kg < goff[s + 1] - goff[s]; kg++) gcol[((size_t)goff[s] + kg) * 64 + lane] = rowmap[r];
                for (; ks < soff[s + 1] - soff[s]; ks++) scol[((size_t)soff[s] + ks) * 64 + lane] = rowmap[r];
            }
        }
        double fmt = (tg + ts) * 64 * 76.0 + ts * 64 * 2.0 + ndof * 16.0 + nrows_i * 6.0;
        printf("tile-symmetric: tiles %d, stored blocks %zu (true %zu, %.1f per row), max staging %zu entries (%.1f KB LDS), stored %.2f GB\n",
               ntiles, (tg + ts) * 64, true_blocks, (double)true_blocks / nn, max_stage, max_stage * 24 / 1024.0, fmt / 1e9); fflush(stdout);
        int *drm, *dgo, *dgc, *dso, *dsc; unsigned short *dsd, *dinc; double *dgv, *dsv;
        CK(hipMalloc(&drm, rowmap.size() * 4)); CK(hipMalloc(&dgo, goff.size() * 4)); CK(hipMalloc(&dso, soff.size() * 4));
        CK(hipMalloc(&dgc, gcol.size() * 4)); CK(hipMalloc(&dsc, scol.size() * 4)); CK(hipMalloc(&dsd, sdst.size() * 2)); CK(hipMalloc(&dinc, inc.size() * 2));
        CK(hipMalloc(&dgv, gval.size() * 8)); CK(hipMalloc(&dsv, sval.size() * 8));
        CK(hipMemcpy(drm, rowmap.data(), rowmap.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgo, goff.data(), goff.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dso, soff.data(), soff.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgc, gcol.data(), gcol.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsc, scol.data(), scol.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsd, sdst.data(), sdst.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dinc, inc.data(), inc.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dgv, gval.data(), gval.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsv, sval.data(), sval.size() * 8, hipMemcpyHostToDevice));
        size_t lds = max_stage * 24;
        CK(hipFuncSetAttribute((const void*)k_tilesym, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        double ms = time_ms([&] { k_tilesym<<<ntiles, TILE, lds>>>(ntiles, drm, dgo, dgc, dgv, dso, dsc, dsd, dsv, dinc, dx, dy); }, reps);
        printf("%-22s %.3f ms  stored %.2f GB -> %.2f TB/s  csr-alg %.2f TB/s\n", "tile-symmetric", ms, fmt / 1e9, fmt / ms / 1e9, csr_bytes / ms / 1e9);
        check("tilesym");
    }
    return 0;
}
