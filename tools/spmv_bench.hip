// tools/spmv_bench.hip -- design-time microbenchmark (not product code, not a test).
// Compares candidate storage formats / kernels for y = K x on the C3-shaped
// operator (n^3 nodes, 27-point node stencil, 3x3 fp64 blocks) so that the
// product format is chosen from measurements on gfx950.
//
//   hipcc -O3 --offload-arch=gfx950 tools/spmv_bench.hip -o tools/spmv_bench -lrocsparse
// The rocSPARSE rows (rocsparse_spmv on scalar CSR, adaptive and row-split algorithms; rocsparse_dbsrmv on 3x3 BSR)
// are the library reference point north_star names; the product uses its own sliced block-ELL kernel.
//   tools/spmv_bench [nodes_per_edge=150] [reps=20]
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>

#define RS(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { printf("rocSPARSE error %d at %s:%d\n", (int)s_, __FILE__, __LINE__); exit(1);} } while (0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static inline double hash01(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return (double)(k >> 11) / 9007199254740992.0 - 0.5;
}

// ---------------- kernels ----------------

// streaming copy (measured HBM ceiling on this box)
__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
__global__ void k_read(const double2* __restrict__ a, double* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += st) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e-300) out[0] = s;
}

// (A) sliced block-ELL, slice = 64 block rows, lane = block row.
//   bcol[(boff[s]+j)*64 + lane], val[((boff[s]+j)*9 + e)*64 + lane]
template <int UNROLL, bool XCD>
__global__ void __launch_bounds__(256) k_sell3(int nslices, int nrows, const int* __restrict__ boff,
                                               const int* __restrict__ bcol, const double* __restrict__ val,
                                               const double* __restrict__ x, double* __restrict__ y,
                                               double* __restrict__ partial) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b = blockIdx.x;
    if (XCD) { int nb = gridDim.x; int q = nb >> 3, r = nb & 7; int xcd = b & 7, i = b >> 3;
               b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i; }
    int s = b * 4 + wave;
    double dot = 0.0;
    if (s < nslices) {
        int o0 = boff[s], o1 = boff[s + 1];
        const int* bc = bcol + (size_t)o0 * 64 + lane;
        const double* v = val + (size_t)o0 * 9 * 64 + lane;
        double y0 = 0, y1 = 0, y2 = 0;
        int W = o1 - o0, j = 0;
        for (; j + UNROLL <= W; j += UNROLL) {
            int c[UNROLL]; double a[UNROLL][9]; double xx[UNROLL][3];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) c[u] = bc[(size_t)(j + u) * 64];
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int e = 0; e < 9; e++) a[u][e] = v[((size_t)(j + u) * 9 + e) * 64];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) { xx[u][0] = x[3 * (size_t)c[u]]; xx[u][1] = x[3 * (size_t)c[u] + 1]; xx[u][2] = x[3 * (size_t)c[u] + 2]; }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                y0 += a[u][0] * xx[u][0] + a[u][1] * xx[u][1] + a[u][2] * xx[u][2];
                y1 += a[u][3] * xx[u][0] + a[u][4] * xx[u][1] + a[u][5] * xx[u][2];
                y2 += a[u][6] * xx[u][0] + a[u][7] * xx[u][1] + a[u][8] * xx[u][2];
            }
        }
        for (; j < W; j++) {
            int c = bc[(size_t)j * 64];
            double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
            const double* a = v + (size_t)j * 9 * 64;
            y0 += a[0] * x0 + a[64] * x1 + a[128] * x2;
            y1 += a[192] * x0 + a[256] * x1 + a[320] * x2;
            y2 += a[384] * x0 + a[448] * x1 + a[512] * x2;
        }
        int row = s * 64 + lane;
        if (row < nrows) {
            y[3 * (size_t)row] = y0; y[3 * (size_t)row + 1] = y1; y[3 * (size_t)row + 2] = y2;
            dot = y0 * x[3 * (size_t)row] + y1 * x[3 * (size_t)row + 1] + y2 * x[3 * (size_t)row + 2];
        }
    }
    // fused p.w partial (block reduce)
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o);
    __shared__ double red[4];
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// (B) sliced block-ELL with pair-packed 16-byte loads: two consecutive blocks
// (j, j+1) = 18 doubles stored as 9 double2 per lane: val2[((boff/2.. )]
//   layout: pairs p = 0..W/2-1 : val[(pbase + p*9 + k)*64 + lane] as double2 (k=0..8)
//   odd tail block stored as 9 doubles [e][lane] after the pairs.
__global__ void __launch_bounds__(256) k_sell3_x4(int nslices, int nrows, const int* __restrict__ boff,
                                                  const int* __restrict__ bcol, const double* __restrict__ val,
                                                  const double* __restrict__ x, double* __restrict__ y,
                                                  double* __restrict__ partial) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int b = blockIdx.x;
    { int nb = gridDim.x; int q = nb >> 3, r = nb & 7; int xcd = b & 7, i = b >> 3;
      b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i; }
    int s = b * 4 + wave;
    double dot = 0.0;
    if (s < nslices) {
        int o0 = boff[s], o1 = boff[s + 1];
        const int* bc = bcol + (size_t)o0 * 64 + lane;
        const double* vbase = val + (size_t)o0 * 9 * 64;
        double y0 = 0, y1 = 0, y2 = 0;
        int W = o1 - o0, np = W >> 1;
        for (int p = 0; p < np; p++) {
            int c0 = bc[(size_t)(2 * p) * 64], c1 = bc[(size_t)(2 * p + 1) * 64];
            const double2* v2 = reinterpret_cast<const double2*>(vbase + (size_t)p * 18 * 64) + lane;
            double2 a[9];
#pragma unroll
            for (int k = 0; k < 9; k++) a[k] = v2[(size_t)k * 64];
            double x0 = x[3 * (size_t)c0], x1 = x[3 * (size_t)c0 + 1], x2 = x[3 * (size_t)c0 + 2];
            double u0 = x[3 * (size_t)c1], u1 = x[3 * (size_t)c1 + 1], u2 = x[3 * (size_t)c1 + 2];
            // 18-vector: block0 e0..e8, block1 e0..e8 ; a[k] = (v[2k], v[2k+1])
            y0 += a[0].x * x0 + a[0].y * x1 + a[1].x * x2;
            y1 += a[1].y * x0 + a[2].x * x1 + a[2].y * x2;
            y2 += a[3].x * x0 + a[3].y * x1 + a[4].x * x2;
            y0 += a[4].y * u0 + a[5].x * u1 + a[5].y * u2;
            y1 += a[6].x * u0 + a[6].y * u1 + a[7].x * u2;
            y2 += a[7].y * u0 + a[8].x * u1 + a[8].y * u2;
        }
        if (W & 1) {
            int c = bc[(size_t)(W - 1) * 64];
            const double* a = vbase + (size_t)np * 18 * 64 + lane;
            double x0 = x[3 * (size_t)c], x1 = x[3 * (size_t)c + 1], x2 = x[3 * (size_t)c + 2];
            y0 += a[0] * x0 + a[64] * x1 + a[128] * x2;
            y1 += a[192] * x0 + a[256] * x1 + a[320] * x2;
            y2 += a[384] * x0 + a[448] * x1 + a[512] * x2;
        }
        int row = s * 64 + lane;
        if (row < nrows) {
            y[3 * (size_t)row] = y0; y[3 * (size_t)row + 1] = y1; y[3 * (size_t)row + 2] = y2;
            dot = y0 * x[3 * (size_t)row] + y1 * x[3 * (size_t)row + 1] + y2 * x[3 * (size_t)row + 2];
        }
    }
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o);
    __shared__ double red[4];
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// (C) scalar CSR, "vector" kernel: LPR lanes per row.
template <int LPR>
__global__ void __launch_bounds__(256) k_csr(int n, const int* __restrict__ rowptr, const int* __restrict__ colind,
                                             const double* __restrict__ val, const double* __restrict__ x,
                                             double* __restrict__ y) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int row = gid / LPR, l = gid % LPR;
    if (row >= n) return;
    double s = 0;
    for (int k = rowptr[row] + l; k < rowptr[row + 1]; k += LPR) s += val[k] * x[colind[k]];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_down(s, o, LPR);
    if (l == 0) y[row] = s;
}

// (D) BSR 3x3, one wave per block row (node), standard blocked storage
//   val[(brow_ptr[v]+j)*9 + e], bcol[brow_ptr[v]+j]
__global__ void __launch_bounds__(256) k_bsr3_wave(int nrows, const int* __restrict__ bptr, const int* __restrict__ bcol,
                                                   const double* __restrict__ val, const double* __restrict__ x,
                                                   double* __restrict__ y) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= nrows) return;
    int b0 = bptr[wave], b1 = bptr[wave + 1];
    int nent = (b1 - b0) * 9;
    const double* v = val + (size_t)b0 * 9;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int i = lane; i < nent; i += 64) {
        int blk = i / 9, e = i - blk * 9, r = e / 3, c = e - r * 3;
        double t = v[i] * x[3 * (size_t)bcol[b0 + blk] + c];
        a0 += (r == 0) ? t : 0.0; a1 += (r == 1) ? t : 0.0; a2 += (r == 2) ? t : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_down(a0, o); a1 += __shfl_down(a1, o); a2 += __shfl_down(a2, o); }
    if (lane == 0) { y[3 * (size_t)wave] = a0; y[3 * (size_t)wave + 1] = a1; y[3 * (size_t)wave + 2] = a2; }
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
    // block pattern
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
    size_t nb = bcol.size(), nnz = nb * 9;
    double csr_bytes = (double)nnz * 12 + (ndof + 1) * 4.0 + ndof * 16.0;
    printf("blocks %zu nnz %zu  CSR-algorithmic bytes %.3f GB\n", nb, nnz, csr_bytes / 1e9);
    std::vector<double> bval(nnz), x(ndof);
    for (size_t i = 0; i < nnz; i++) bval[i] = hash01(i * 2654435761ULL + 17);
    for (size_t i = 0; i < ndof; i++) x[i] = hash01(i + 0x9e3779b97f4a7c15ULL);
    // reference y on a sample of rows
    auto yref = [&](size_t v, int r) { double s = 0; for (int b = bptr[v]; b < bptr[v + 1]; b++) for (int c = 0; c < 3; c++) s += bval[(size_t)b * 9 + r * 3 + c] * x[3 * (size_t)bcol[b] + c]; return s; };

    double *dx, *dy, *dpart; CK(hipMalloc(&dx, ndof * 8)); CK(hipMalloc(&dy, ndof * 8)); CK(hipMalloc(&dpart, 1 << 20));
    CK(hipMemcpy(dx, x.data(), ndof * 8, hipMemcpyHostToDevice));
    std::vector<double> yh(ndof);
    auto check = [&](const char* name) {
        CK(hipMemcpy(yh.data(), dy, ndof * 8, hipMemcpyDeviceToHost));
        double mx = 0;
        for (size_t t = 0; t < 2000; t++) { size_t v = (t * 7919 * 131) % nn; for (int r = 0; r < 3; r++) mx = std::max(mx, fabs(yh[3 * v + r] - yref(v, r))); }
        size_t v = nn - 1; for (int r = 0; r < 3; r++) mx = std::max(mx, fabs(yh[3 * v + r] - yref(v, r)));
        printf("   check %-12s max abs err %.3e\n", name, mx);
        CK(hipMemset(dy, 0, ndof * 8));
    };

    // ---- HBM ceiling ----
    {
        size_t nbytes = (size_t)4 << 30; double2 *a, *b; CK(hipMalloc(&a, nbytes)); CK(hipMalloc(&b, nbytes));
        CK(hipMemset(a, 1, nbytes));
        double ms = time_ms([&] { k_copy<<<2048 * 4, 256>>>(a, b, nbytes / 16); }, reps);
        printf("copy 4GiB->4GiB: %.3f ms  %.2f TB/s (r+w)\n", ms, 2.0 * nbytes / ms / 1e9);
        ms = time_ms([&] { k_read<<<2048 * 4, 256>>>(a, dpart, nbytes / 16); }, reps);
        printf("read 4GiB: %.3f ms  %.2f TB/s\n", ms, 1.0 * nbytes / ms / 1e9);
        CK(hipFree(a)); CK(hipFree(b));
    }

    // ---- (A) SELL-64 ----
    {
        int nsl = (int)((nn + 63) / 64);
        std::vector<int> boff(nsl + 1, 0);
        for (int s = 0; s < nsl; s++) { int W = 0; for (size_t v = (size_t)s * 64; v < std::min(nn, (size_t)(s + 1) * 64); v++) W = std::max(W, bptr[v + 1] - bptr[v]); boff[s + 1] = boff[s] + W; }
        size_t tb = (size_t)boff[nsl];
        printf("SELL-64: slices %d padded blocks %zu (%.2f%% pad)  format bytes %.3f GB\n", nsl, tb * 64, 100.0 * (tb * 64 - nb) / nb,
               (tb * 64 * 76.0 + ndof * 16.0) / 1e9);
        std::vector<int> sc(tb * 64); std::vector<double> sv(tb * 64 * 9, 0.0);
        for (int s = 0; s < nsl; s++) for (int l = 0; l < 64; l++) {
            size_t v = (size_t)s * 64 + l; int W = boff[s + 1] - boff[s];
            for (int j = 0; j < W; j++) {
                size_t o = (size_t)boff[s] + j;
                bool has = v < nn && j < bptr[v + 1] - bptr[v];
                sc[o * 64 + l] = has ? bcol[bptr[v] + j] : (int)std::min(v, nn - 1);
                for (int e = 0; e < 9; e++) {
                    double a = has ? bval[((size_t)bptr[v] + j) * 9 + e] : 0.0;
                    sv[(o * 9 + e) * 64 + l] = a;
                }
            }
        }
        int *dboff, *dsc; double* dsv;
        CK(hipMalloc(&dboff, (nsl + 1) * 4)); CK(hipMalloc(&dsc, sc.size() * 4)); CK(hipMalloc(&dsv, sv.size() * 8));
        CK(hipMemcpy(dboff, boff.data(), (nsl + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsv, sv.data(), sv.size() * 8, hipMemcpyHostToDevice));
        int grid = (nsl + 3) / 4;
        double fmt_bytes = tb * 64 * 76.0 + ndof * 16.0;
#define RUN(NAME, ...) { double ms = time_ms([&] { __VA_ARGS__; }, reps); printf("%-28s %.3f ms  fmt %.2f TB/s  csr-alg %.2f TB/s\n", NAME, ms, fmt_bytes / ms / 1e9, csr_bytes / ms / 1e9); check(NAME); }
        RUN("sell3 u1", (k_sell3<1, false><<<grid, 256>>>(nsl, (int)nn, dboff, dsc, dsv, dx, dy, dpart)));
        RUN("sell3 u1 xcd", (k_sell3<1, true><<<grid, 256>>>(nsl, (int)nn, dboff, dsc, dsv, dx, dy, dpart)));
        RUN("sell3 u2 xcd", (k_sell3<2, true><<<grid, 256>>>(nsl, (int)nn, dboff, dsc, dsv, dx, dy, dpart)));
        RUN("sell3 u3 xcd", (k_sell3<3, true><<<grid, 256>>>(nsl, (int)nn, dboff, dsc, dsv, dx, dy, dpart)));
        RUN("sell3 u3", (k_sell3<3, false><<<grid, 256>>>(nsl, (int)nn, dboff, dsc, dsv, dx, dy, dpart)));
        // pair-packed layout (rebuilt in place of sv to bound host memory)
        std::fill(sv.begin(), sv.end(), 0.0);
        for (int s = 0; s < nsl; s++) for (int l = 0; l < 64; l++) {
            size_t v = (size_t)s * 64 + l; int W = boff[s + 1] - boff[s]; int np = W >> 1; size_t base = (size_t)boff[s] * 9 * 64;
            for (int j = 0; j < W; j++) {
                bool has = v < nn && j < bptr[v + 1] - bptr[v];
                for (int e = 0; e < 9; e++) {
                    double a = has ? bval[((size_t)bptr[v] + j) * 9 + e] : 0.0;
                    if (j < 2 * np) { int p = j >> 1, idx = (j & 1) * 9 + e; int k = idx >> 1, h = idx & 1; sv[base + ((size_t)p * 18 + 2 * k) * 64 + 2 * l + h] = a; }
                    else sv[base + (size_t)np * 18 * 64 + (size_t)e * 64 + l] = a;
                }
            }
        }
        CK(hipMemcpy(dsv, sv.data(), sv.size() * 8, hipMemcpyHostToDevice));
        RUN("sell3 pair16 xcd", (k_sell3_x4<<<grid, 256>>>(nsl, (int)nn, dboff, dsc, dsv, dx, dy, dpart)));
        CK(hipFree(dboff)); CK(hipFree(dsc)); CK(hipFree(dsv));
    }
    // ---- (D) BSR wave per row ----
    {
        int *dbp, *dbc; double* dbv;
        CK(hipMalloc(&dbp, (nn + 1) * 4)); CK(hipMalloc(&dbc, nb * 4)); CK(hipMalloc(&dbv, nnz * 8));
        CK(hipMemcpy(dbp, bptr.data(), (nn + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbc, bcol.data(), nb * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbv, bval.data(), nnz * 8, hipMemcpyHostToDevice));
        double fmt_bytes = nb * 76.0 + ndof * 16.0;
        RUN("bsr3 wave/row", (k_bsr3_wave<<<(unsigned)((nn * 64 + 255) / 256), 256>>>((int)nn, dbp, dbc, dbv, dx, dy)));
        {   // rocSPARSE BSR, block dimension 3, row-major blocks
            rocsparse_handle h; rocsparse_mat_descr descr; rocsparse_mat_info info;
            RS(rocsparse_create_handle(&h)); RS(rocsparse_create_mat_descr(&descr)); RS(rocsparse_create_mat_info(&info));
            const double one = 1.0, zero = 0.0;
            RS(rocsparse_dbsrmv_analysis(h, rocsparse_direction_row, rocsparse_operation_none, (rocsparse_int)nn, (rocsparse_int)nn,
                                         (rocsparse_int)nb, descr, dbv, dbp, dbc, 3, info));
            RUN("rocsparse_dbsrmv bs3", RS(rocsparse_dbsrmv(h, rocsparse_direction_row, rocsparse_operation_none, (rocsparse_int)nn,
                                                            (rocsparse_int)nn, (rocsparse_int)nb, &one, descr, dbv, dbp, dbc, 3, info,
                                                            dx, &zero, dy)));
            RS(rocsparse_destroy_mat_info(info)); RS(rocsparse_destroy_mat_descr(descr)); RS(rocsparse_destroy_handle(h));
        }
        CK(hipFree(dbp)); CK(hipFree(dbc)); CK(hipFree(dbv));
    }
    // ---- (C) scalar CSR ----
    {
        std::vector<int> rp(ndof + 1), ci(nnz); std::vector<double> cv(nnz);
        size_t k = 0;
        for (size_t v = 0; v < nn; v++) for (int r = 0; r < 3; r++) {
            rp[3 * v + r] = (int)k;
            for (int b = bptr[v]; b < bptr[v + 1]; b++) for (int c = 0; c < 3; c++) { ci[k] = 3 * bcol[b] + c; cv[k] = bval[(size_t)b * 9 + r * 3 + c]; k++; }
        }
        rp[ndof] = (int)k;
        int *drp, *dci; double* dcv;
        CK(hipMalloc(&drp, (ndof + 1) * 4)); CK(hipMalloc(&dci, nnz * 4)); CK(hipMalloc(&dcv, nnz * 8));
        CK(hipMemcpy(drp, rp.data(), (ndof + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dcv, cv.data(), nnz * 8, hipMemcpyHostToDevice));
        double fmt_bytes = csr_bytes;
        RUN("csr 8 lanes/row", (k_csr<8><<<(unsigned)((ndof * 8 + 255) / 256), 256>>>((int)ndof, drp, dci, dcv, dx, dy)));
        RUN("csr 16 lanes/row", (k_csr<16><<<(unsigned)((ndof * 16 + 255) / 256), 256>>>((int)ndof, drp, dci, dcv, dx, dy)));
        RUN("csr 32 lanes/row", (k_csr<32><<<(unsigned)((ndof * 32 + 255) / 256), 256>>>((int)ndof, drp, dci, dcv, dx, dy)));
        {   // rocSPARSE generic SpMV on the scalar CSR matrix (what "rocSPARSE CSR SpMV" in north_star means)
            rocsparse_handle h; RS(rocsparse_create_handle(&h));
            rocsparse_spmat_descr A; rocsparse_dnvec_descr vx, vy;
            RS(rocsparse_create_csr_descr(&A, (int64_t)ndof, (int64_t)ndof, (int64_t)nnz, drp, dci, dcv, rocsparse_indextype_i32,
                                          rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r));
            RS(rocsparse_create_dnvec_descr(&vx, (int64_t)ndof, dx, rocsparse_datatype_f64_r));
            RS(rocsparse_create_dnvec_descr(&vy, (int64_t)ndof, dy, rocsparse_datatype_f64_r));
            const double one = 1.0, zero = 0.0;
            const rocsparse_spmv_alg algs[2] = {rocsparse_spmv_alg_csr_adaptive, rocsparse_spmv_alg_csr_rowsplit};   // (csr_lrb returns at once without computing on this build: left out)
            const char* names[2] = {"rocsparse_spmv csr adaptive", "rocsparse_spmv csr rowsplit"};
            for (int a = 0; a < 2; a++) {
                size_t bsz = 0; void* buf = nullptr;
                RS(rocsparse_spmv(h, rocsparse_operation_none, &one, A, vx, &zero, vy, rocsparse_datatype_f64_r, algs[a],
                                  rocsparse_spmv_stage_buffer_size, &bsz, nullptr));
                CK(hipMalloc(&buf, bsz ? bsz : 8));
                RS(rocsparse_spmv(h, rocsparse_operation_none, &one, A, vx, &zero, vy, rocsparse_datatype_f64_r, algs[a],
                                  rocsparse_spmv_stage_preprocess, &bsz, buf));
                RUN(names[a], RS(rocsparse_spmv(h, rocsparse_operation_none, &one, A, vx, &zero, vy, rocsparse_datatype_f64_r, algs[a],
                                                rocsparse_spmv_stage_compute, &bsz, buf)));
                CK(hipFree(buf));
            }
            RS(rocsparse_destroy_dnvec_descr(vx)); RS(rocsparse_destroy_dnvec_descr(vy)); RS(rocsparse_destroy_spmat_descr(A));
            RS(rocsparse_destroy_handle(h));
        }
        CK(hipFree(drp)); CK(hipFree(dci)); CK(hipFree(dcv));
    }
    return 0;
}
