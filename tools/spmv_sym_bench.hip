// tools/spmv_sym_bench.hip -- design-time experiment (not product code).
// Question: K is symmetric; can y = K x stream only the upper block triangle
// from HBM and pick the lower blocks (= transposed upper blocks of earlier
// rows) out of the 256 MiB Infinity Cache / L2, where they still sit from
// ~23 MB of traffic ago?  Compares the full sliced block-ELL kernel with a
// "symmetric gather" variant on the C3-shaped operator.
//
//   hipcc -O3 --offload-arch=gfx950 tools/spmv_sym_bench.hip -o tools/spmv_sym_bench
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
// symmetric operator: A[i][j][r][c] == A[j][i][c][r]
static inline double aval(int64_t i, int64_t j, int r, int c) {
    if (i > j || (i == j && r > c)) { std::swap(i, j); std::swap(r, c); }
    return hash01((uint64_t)i * 1000003ULL * 9 + (uint64_t)(j - i + 40000) * 9 + r * 3 + c + 12345);
}

// full sliced block-ELL (the product kernel's structure)
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

// symmetric gather: upper blocks (col >= row) streamed; lower blocks read as the
// transposed upper block of the earlier row through a precomputed base index.
//   ucol/uval : sliced ELL of the upper triangle        (uoff)
//   lcol/lbase: sliced ELL of the lower neighbours: column node id and index of
//               entry (0,0) of the stored upper block (j,i) in uval (loff)
template <bool XCD>
__global__ void __launch_bounds__(256) k_sym(int nslices, int nrows, const int* __restrict__ uoff, const int* __restrict__ ucol,
                                             const double* __restrict__ uval, const int* __restrict__ loff,
                                             const int* __restrict__ lcol, const int* __restrict__ lbase,
                                             const double* __restrict__ x, double* __restrict__ y) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b = blockIdx.x;
    if (XCD) { int nb = gridDim.x; int q = nb >> 3, r = nb & 7; int xcd = b & 7, i = b >> 3;
               b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i; }
    int s = b * 4 + wave;
    if (s >= nslices) return;
    double y0 = 0, y1 = 0, y2 = 0;
    {   // lower part first: these lines were fetched long ago (cache hits hoped for)
        int o0 = loff[s], W = loff[s + 1] - o0;
        const int* lc = lcol + (size_t)o0 * 64 + lane;
        const int* lb = lbase + (size_t)o0 * 64 + lane;
        for (int j = 0; j < W; j++) {
            int c = lc[(size_t)j * 64];
            int base = lb[(size_t)j * 64];
            if (base < 0) continue;   // padding
            double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
            const double* a = uval + base;   // block (c,row); we need its transpose
            y0 += a[0] * x0 + a[192] * x1 + a[384] * x2;
            y1 += a[64] * x0 + a[256] * x1 + a[448] * x2;
            y2 += a[128] * x0 + a[320] * x1 + a[512] * x2;
        }
    }
    {
        int o0 = uoff[s], W = uoff[s + 1] - o0;
        const int* bc = ucol + (size_t)o0 * 64 + lane;
        const double* v = uval + (size_t)o0 * 9 * 64 + lane;
        for (int j = 0; j < W; j++) {
            int c = bc[(size_t)j * 64];
            double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
            const double* a = v + (size_t)j * 9 * 64;
            y0 += a[0] * x0 + a[64] * x1 + a[128] * x2;
            y1 += a[192] * x0 + a[256] * x1 + a[320] * x2;
            y2 += a[384] * x0 + a[448] * x1 + a[512] * x2;
        }
    }
    int row = s * 64 + lane;
    if (row < nrows) { y[3 * (size_t)row] = y0; y[3 * (size_t)row + 1] = y1; y[3 * (size_t)row + 2] = y2; }
}

// same, but workgroup b processes the slices sched[4*pos .. 4*pos+3] where pos is
// the XCD-chunked position of b: each XCD walks a contiguous part of a tiled sweep
// so that the previous plane's upper blocks are still in ITS 4 MiB L2.
__global__ void __launch_bounds__(256) k_sym_sched(int nslices, int nrows, const int* __restrict__ sched,
                                                   const int* __restrict__ uoff, const int* __restrict__ ucol,
                                                   const double* __restrict__ uval, const int* __restrict__ loff,
                                                   const int* __restrict__ lcol, const int* __restrict__ lbase,
                                                   const double* __restrict__ x, double* __restrict__ y) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b = blockIdx.x;
    { int nb = gridDim.x; int q = nb >> 3, r = nb & 7; int xcd = b & 7, i = b >> 3;
      b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i; }
    int pos = b * 4 + wave;
    if (pos >= nslices) return;
    int s = sched[pos];
    double y0 = 0, y1 = 0, y2 = 0;
    {
        int o0 = loff[s], W = loff[s + 1] - o0;
        const int* lc = lcol + (size_t)o0 * 64 + lane;
        const int* lb = lbase + (size_t)o0 * 64 + lane;
        for (int j = 0; j < W; j++) {
            int c = lc[(size_t)j * 64];
            int base = lb[(size_t)j * 64];
            if (base < 0) continue;
            double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
            const double* a = uval + base;
            y0 += a[0] * x0 + a[192] * x1 + a[384] * x2;
            y1 += a[64] * x0 + a[256] * x1 + a[448] * x2;
            y2 += a[128] * x0 + a[320] * x1 + a[512] * x2;
        }
    }
    {
        int o0 = uoff[s], W = uoff[s + 1] - o0;
        const int* bc = ucol + (size_t)o0 * 64 + lane;
        const double* v = uval + (size_t)o0 * 9 * 64 + lane;
        for (int j = 0; j < W; j++) {
            int c = bc[(size_t)j * 64];
            double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
            const double* a = v + (size_t)j * 9 * 64;
            y0 += a[0] * x0 + a[64] * x1 + a[128] * x2;
            y1 += a[192] * x0 + a[256] * x1 + a[320] * x2;
            y2 += a[384] * x0 + a[448] * x1 + a[512] * x2;
        }
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
    size_t nn = (size_t)n * n * n, ndof = 3 * nn;
    printf("nodes/edge %d nodes %zu dof %zu\n", n, nn, ndof);
    std::vector<int> bptr(nn + 1); std::vector<int> bcol; bcol.reserve(nn * 27);
    for (int k = 0; k < n; k++) for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
        size_t v = i + (size_t)n * (j + (size_t)n * k);
        bptr[v] = (int)bcol.size();
        for (int dk = -1; dk <= 1; dk++) for (int dj = -1; dj <= 1; dj++) for (int di = -1; di <= 1; di++) {
            int ii = i + di, jj = j + dj, kk = k + dk;
            if (ii < 0 || jj < 0 || kk < 0 || ii >= n || jj >= n || kk >= n) continue;
            bcol.push_back((int)(ii + (size_t)n * (jj + (size_t)n * kk)));
        }
    }
    bptr[nn] = (int)bcol.size();
    size_t nb = bcol.size();
    double csr_bytes = (double)nb * 9 * 12 + (ndof + 1) * 4.0 + ndof * 16.0;
    std::vector<double> x(ndof);
    for (size_t i = 0; i < ndof; i++) x[i] = hash01(i + 0x9e3779b97f4a7c15ULL);
    auto yref = [&](size_t v, int r) { double s = 0; for (int b = bptr[v]; b < bptr[v + 1]; b++) for (int c = 0; c < 3; c++) s += aval(v, bcol[b], r, c) * x[3 * (size_t)bcol[b] + c]; return s; };
    double *dx, *dy; CK(hipMalloc(&dx, ndof * 8)); CK(hipMalloc(&dy, ndof * 8));
    CK(hipMemcpy(dx, x.data(), ndof * 8, hipMemcpyHostToDevice));
    std::vector<double> yh(ndof);
    auto check = [&](const char* name) {
        CK(hipMemcpy(yh.data(), dy, ndof * 8, hipMemcpyDeviceToHost));
        double mx = 0;
        for (size_t t = 0; t < 3000; t++) { size_t v = (t * 7919 * 131) % nn; for (int r = 0; r < 3; r++) mx = std::max(mx, fabs(yh[3 * v + r] - yref(v, r))); }
        for (size_t v : {(size_t)0, (size_t)63, (size_t)64, nn - 1}) for (int r = 0; r < 3; r++) mx = std::max(mx, fabs(yh[3 * v + r] - yref(v, r)));
        printf("   check %-14s max abs err %.3e\n", name, mx);
        CK(hipMemset(dy, 0, ndof * 8));
    };
    int nsl = (int)((nn + 63) / 64);
    // ---- full SELL ----
    {
        std::vector<int> boff(nsl + 1, 0);
        for (int s = 0; s < nsl; s++) { int W = 0; for (size_t v = (size_t)s * 64; v < std::min(nn, (size_t)(s + 1) * 64); v++) W = std::max(W, bptr[v + 1] - bptr[v]); boff[s + 1] = boff[s] + W; }
        size_t tb = (size_t)boff[nsl];
        std::vector<int> sc(tb * 64); std::vector<double> sv(tb * 64 * 9, 0.0);
        for (int s = 0; s < nsl; s++) for (int l = 0; l < 64; l++) {
            size_t v = (size_t)s * 64 + l; int W = boff[s + 1] - boff[s];
            for (int j = 0; j < W; j++) {
                size_t o = (size_t)boff[s] + j;
                bool has = v < nn && j < bptr[v + 1] - bptr[v];
                sc[o * 64 + l] = has ? bcol[bptr[v] + j] : (int)std::min(v, nn - 1);
                if (has) for (int e = 0; e < 9; e++) sv[(o * 9 + e) * 64 + l] = aval(v, bcol[bptr[v] + j], e / 3, e % 3);
            }
        }
        int *dboff, *dsc; double* dsv;
        CK(hipMalloc(&dboff, (nsl + 1) * 4)); CK(hipMalloc(&dsc, sc.size() * 4)); CK(hipMalloc(&dsv, sv.size() * 8));
        CK(hipMemcpy(dboff, boff.data(), (nsl + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsv, sv.data(), sv.size() * 8, hipMemcpyHostToDevice));
        double fmt = tb * 64 * 76.0 + ndof * 16.0;
        double ms = time_ms([&] { k_full<<<(nsl + 3) / 4, 256>>>(nsl, (int)nn, dboff, dsc, dsv, dx, dy); }, reps);
        printf("%-22s %.3f ms  fmt %.2f TB/s (%.2f GB)  csr-alg %.2f TB/s\n", "full sliced ELL", ms, fmt / ms / 1e9, fmt / 1e9, csr_bytes / ms / 1e9);
        check("full");
        CK(hipFree(dboff)); CK(hipFree(dsc)); CK(hipFree(dsv));
    }
    // ---- symmetric gather ----
    {
        std::vector<int> uoff(nsl + 1, 0), loff(nsl + 1, 0);
        for (int s = 0; s < nsl; s++) {
            int WU = 0, WL = 0;
            for (size_t v = (size_t)s * 64; v < std::min(nn, (size_t)(s + 1) * 64); v++) {
                int nu = 0, nl = 0;
                for (int b = bptr[v]; b < bptr[v + 1]; b++) { if ((size_t)bcol[b] >= v) nu++; else nl++; }
                WU = std::max(WU, nu); WL = std::max(WL, nl);
            }
            uoff[s + 1] = uoff[s] + WU; loff[s + 1] = loff[s] + WL;
        }
        size_t tu = (size_t)uoff[nsl], tl = (size_t)loff[nsl];
        std::vector<int> uc(tu * 64), lc(tl * 64), lb(tl * 64, -1);
        std::vector<double> uv(tu * 64 * 9, 0.0);
        // slot of column j in the upper row of i
        auto uslot = [&](size_t i, size_t j) { int k = 0; for (int b = bptr[i]; b < bptr[i + 1]; b++) { if ((size_t)bcol[b] < i) continue; if ((size_t)bcol[b] == j) return k; k++; } return -1; };
        for (int s = 0; s < nsl; s++) for (int l = 0; l < 64; l++) {
            size_t v = (size_t)s * 64 + l;
            int WU = uoff[s + 1] - uoff[s], WL = loff[s + 1] - loff[s], ku = 0, kl = 0;
            if (v < nn) for (int b = bptr[v]; b < bptr[v + 1]; b++) {
                size_t j = bcol[b];
                if (j >= v) {
                    size_t o = (size_t)uoff[s] + ku;
                    uc[o * 64 + l] = (int)j;
                    for (int e = 0; e < 9; e++) uv[(o * 9 + e) * 64 + l] = aval(v, j, e / 3, e % 3);
                    ku++;
                } else {
                    size_t o = (size_t)loff[s] + kl;
                    lc[o * 64 + l] = (int)j;
                    int sl = uslot(j, v);
                    size_t sj = j >> 6; int lj = (int)(j & 63);
                    lb[o * 64 + l] = (int)((((size_t)uoff[sj] + sl) * 9) * 64 + lj);
                    kl++;
                }
            }
            for (; ku < WU; ku++) uc[((size_t)uoff[s] + ku) * 64 + l] = (int)std::min(v, nn - 1);
            for (; kl < WL; kl++) lc[((size_t)loff[s] + kl) * 64 + l] = (int)std::min(v, nn - 1);
        }
        int *duoff, *dloff, *duc, *dlc, *dlb; double* duv;
        CK(hipMalloc(&duoff, (nsl + 1) * 4)); CK(hipMalloc(&dloff, (nsl + 1) * 4));
        CK(hipMalloc(&duc, uc.size() * 4)); CK(hipMalloc(&dlc, lc.size() * 4)); CK(hipMalloc(&dlb, lb.size() * 4)); CK(hipMalloc(&duv, uv.size() * 8));
        CK(hipMemcpy(duoff, uoff.data(), (nsl + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dloff, loff.data(), (nsl + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(duc, uc.data(), uc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlc, lc.data(), lc.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dlb, lb.data(), lb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(duv, uv.data(), uv.size() * 8, hipMemcpyHostToDevice));
        double fmt = tu * 64 * 76.0 + tl * 64 * 8.0 + ndof * 16.0;
        printf("symmetric storage: upper blocks %zu (padded), lower refs %zu; HBM-side bytes if lower blocks hit cache: %.2f GB\n", tu * 64, tl * 64, fmt / 1e9);
        double ms = time_ms([&] { k_sym<false><<<(nsl + 3) / 4, 256>>>(nsl, (int)nn, duoff, duc, duv, dloff, dlc, dlb, dx, dy); }, reps);
        printf("%-22s %.3f ms  csr-alg %.2f TB/s\n", "symmetric gather", ms, csr_bytes / ms / 1e9);
        check("sym");
        ms = time_ms([&] { k_sym<true><<<(nsl + 3) / 4, 256>>>(nsl, (int)nn, duoff, duc, duv, dloff, dlc, dlb, dx, dy); }, reps);
        printf("%-22s %.3f ms  csr-alg %.2f TB/s\n", "symmetric gather xcd", ms, csr_bytes / ms / 1e9);
        check("sym xcd");
        int* dsched; CK(hipMalloc(&dsched, (size_t)nsl * 4));
        for (int TJ : {5, 10, 19, 38, 150}) {
            std::vector<long long> key(nsl); std::vector<int> sched(nsl);
            for (int s = 0; s < nsl; s++) {
                size_t v0 = (size_t)s * 64; int i = (int)(v0 % n), j = (int)((v0 / n) % n), k = (int)(v0 / ((size_t)n * n));
                key[s] = (((long long)(j / TJ) * n + k) * n + j) * n + i; sched[s] = s;
            }
            std::sort(sched.begin(), sched.end(), [&](int a, int b) { return key[a] < key[b]; });
            CK(hipMemcpy(dsched, sched.data(), (size_t)nsl * 4, hipMemcpyHostToDevice));
            ms = time_ms([&] { k_sym_sched<<<(nsl + 3) / 4, 256>>>(nsl, (int)nn, dsched, duoff, duc, duv, dloff, dlc, dlb, dx, dy); }, reps);
            char nm[64]; snprintf(nm, 64, "sym tiled TJ=%d", TJ);
            printf("%-22s %.3f ms  csr-alg %.2f TB/s\n", nm, ms, csr_bytes / ms / 1e9);
            check(nm);
        }
    }
    return 0;
}
