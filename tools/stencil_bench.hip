// tools/stencil_bench.hip -- experiment (round 5): y = K x for a mesh whose assembled block rows fall into a few classes (a
// structured brick: interior, faces, edges, corners, penalty rows), with the matrix stored as ONE copy of every distinct block row
// ("row-class dictionary") instead of 81 nonzeros per row or the element loop of -mat_mult_type ebe.
//   tiles: boxes of 15 x 7 x 7 owned nodes on the node lattice, window 17 x 9 x 9 (the rim of one node), x of the window staged in
//   LDS in lattice order, thread = owned node: 27 blocks of 3 x 3 from the dictionary (uniform class per wave: scalar loads) times
//   the 27 neighbours from LDS.  No colours, no LDS sums, no rim elements computed twice.
// Synthetic: the dictionary holds random blocks, the classes are the 27 position types of a brick; checked against a host loop.
// Build: hipcc -O3 --offload-arch=gfx950 tools/stencil_bench.hip -o tools/stencil_bench ; run: tools/stencil_bench [nodes per edge] [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e = (x);                                                                \
        if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } \
    } while (0)

constexpr int BX = 15, BY = 7, BZ = 7;                 // owned nodes of a tile
constexpr int WX = BX + 2, WY = BY + 2, WZ = BZ + 2;   // window
constexpr int NW = WX * WY * WZ, LP = 1408;            // 1377 window positions, padded
constexpr int NOWN = BX * BY * BZ, OP = 768;           // 735 owned, padded

struct alignas(8) Node3 { double a, b, c; };

template <int WPS, int MODE>   // MODE 0: one node at a time, coefficients through scalar loads; 1: the same with vector loads of the dictionary;
                               // 2: the thread's three nodes together where the whole wave has one class (coefficients loaded once for the three)
__global__ void __launch_bounds__(256, WPS) k_stencil_tile(int ntiles, const int* __restrict__ lat, const unsigned short* __restrict__ cls,
                                                           const double* __restrict__ dict, const double* __restrict__ x,
                                                           double* __restrict__ y, double* __restrict__ partial)
{
    extern __shared__ double xs[];   // [3][LP]
    const unsigned chunk = gridDim.x / 8, t = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
    if (t >= (unsigned)ntiles) return;
    const int* lt = lat + (size_t)t * LP;
    for (int p = threadIdx.x; p < NW; p += 256) {
        const int v = lt[p];
        Node3 u{0.0, 0.0, 0.0};
        if (v >= 0) u = *reinterpret_cast<const Node3*>(x + (size_t)v * 3);
        xs[p] = u.a; xs[LP + p] = u.b; xs[2 * LP + p] = u.c;
    }
    __syncthreads();
    double dot = 0.0;
    typedef const double __attribute__((address_space(4))) * cdp;
    auto uniform_ptr = [&](unsigned cu) {
        const unsigned long long ad = (unsigned long long)(dict + (size_t)cu * 243);
        return (cdp)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ad >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ad));
    };
    if constexpr (MODE == 2) {
        unsigned c3[3];
        int p3[3];
        bool same = true;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int o = threadIdx.x + 256 * r;
            c3[r] = o < NOWN ? cls[(size_t)t * OP + o] : 0xffffu;
            const int i = o % BX, j = (o / BX) % BY, k = o / (BX * BY);
            p3[r] = o < NOWN ? (i + 1) + WX * ((j + 1) + WY * (k + 1)) : 1 + WX * (1 + WY);
            same = same && (c3[r] == c3[0] || o >= NOWN);
        }
        const unsigned c0 = __builtin_amdgcn_readfirstlane(c3[0]);
        if (__all(same && c3[0] == c0 && c0 != 0xffffu)) {
            const cdp A = uniform_ptr(c0);
            double a[3][3] = {};
#pragma unroll 1
            for (int sy = 0; sy < 9; sy++) {   // (rolled: unrolled, the compiler hoists all 243 LDS reads of the three nodes and spills)
                const int offy = WX * (sy % 3 - 1) + WX * WY * (sy / 3 - 1);
#pragma unroll
                for (int sx = 0; sx < 3; sx++) {
                    const int off = offy + sx - 1, s9 = 9 * (3 * sy + sx);
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const double u0 = xs[p3[r] + off], u1 = xs[LP + p3[r] + off], u2 = xs[2 * LP + p3[r] + off];
                        a[r][0] += A[s9 + 0] * u0 + A[s9 + 1] * u1 + A[s9 + 2] * u2;
                        a[r][1] += A[s9 + 3] * u0 + A[s9 + 4] * u1 + A[s9 + 5] * u2;
                        a[r][2] += A[s9 + 6] * u0 + A[s9 + 7] * u1 + A[s9 + 8] * u2;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 3; r++) {
                if ((int)threadIdx.x + 256 * r >= NOWN) continue;
                const int v = lt[p3[r]];
                *reinterpret_cast<Node3*>(y + (size_t)v * 3) = Node3{a[r][0], a[r][1], a[r][2]};
                dot += xs[p3[r]] * a[r][0] + xs[LP + p3[r]] * a[r][1] + xs[2 * LP + p3[r]] * a[r][2];
            }
            goto finish;
        }
    }
    for (int o = threadIdx.x; o < OP; o += 256) {
        const unsigned c = o < NOWN ? cls[(size_t)t * OP + o] : 0xffffu;
        const int i = o % BX, j = (o / BX) % BY, k = o / (BX * BY);
        const int p = (i + 1) + WX * ((j + 1) + WY * (k + 1));
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        bool todo = c != 0xffffu;
        while (true) {
            const unsigned long long m = __ballot(todo);
            if (!m) break;
            const unsigned cu = __builtin_amdgcn_readlane(c, __builtin_ctzll(m));
            if (todo && c == cu) {
                // (constant address space + a wave-uniform address: the 243 coefficients arrive through scalar loads)
                auto body = [&](auto A) {
#pragma unroll
                    for (int s = 0; s < 27; s++) {
                        const int off = (s % 3 - 1) + WX * ((s / 3) % 3 - 1) + WX * WY * (s / 9 - 1);
                        const double u0 = xs[p + off], u1 = xs[LP + p + off], u2 = xs[2 * LP + p + off];
                        a0 += A[9 * s + 0] * u0 + A[9 * s + 1] * u1 + A[9 * s + 2] * u2;
                        a1 += A[9 * s + 3] * u0 + A[9 * s + 4] * u1 + A[9 * s + 5] * u2;
                        a2 += A[9 * s + 6] * u0 + A[9 * s + 7] * u1 + A[9 * s + 8] * u2;
                    }
                };
                if constexpr (MODE == 1) body(dict + (size_t)cu * 243); else body(uniform_ptr(cu));
                todo = false;
            }
        }
        if (c != 0xffffu) {
            const int v = lt[p];
            *reinterpret_cast<Node3*>(y + (size_t)v * 3) = Node3{a0, a1, a2};
            dot += xs[p] * a0 + xs[LP + p] * a1 + xs[2 * LP + p] * a2;
        }
    }
finish:
    __shared__ double sh[4];
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_down(dot, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partial[t] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 150, reps = argc > 2 ? atoi(argv[2]) : 50;
    const bool one_class = argc > 3 && atoi(argv[3]) != 0;   // every node the interior class (timing of the uniform path; the check is skipped)
    const int tb[3] = {(N + BX - 1) / BX, (N + BY - 1) / BY, (N + BZ - 1) / BZ};
    const int ntiles = tb[0] * tb[1] * tb[2];
    const size_t nn = (size_t)N * N * N;
    printf("%d^3 nodes, %d tiles of %d x %d x %d owned nodes\n", N, ntiles, BX, BY, BZ);
    std::vector<int> lat((size_t)ntiles * LP, -1);
    std::vector<unsigned short> cls((size_t)ntiles * OP, 0xffff);
    auto type_of = [&](int i) { return i == 0 ? 0 : i == N - 1 ? 2 : 1; };
    for (int tz = 0, t = 0; tz < tb[2]; tz++)
        for (int ty = 0; ty < tb[1]; ty++)
            for (int tx = 0; tx < tb[0]; tx++, t++) {
                for (int k = 0; k < WZ; k++)
                    for (int j = 0; j < WY; j++)
                        for (int i = 0; i < WX; i++) {
                            const int gi = tx * BX - 1 + i, gj = ty * BY - 1 + j, gk = tz * BZ - 1 + k;
                            if (gi < 0 || gj < 0 || gk < 0 || gi >= N || gj >= N || gk >= N) continue;
                            lat[(size_t)t * LP + i + WX * (j + WY * k)] = gi + N * (gj + N * gk);
                        }
                for (int o = 0; o < NOWN; o++) {
                    const int gi = tx * BX + o % BX, gj = ty * BY + (o / BX) % BY, gk = tz * BZ + o / (BX * BY);
                    if (gi >= N || gj >= N || gk >= N) continue;
                    cls[(size_t)t * OP + o] = one_class ? (unsigned short)13 : (unsigned short)(type_of(gi) + 3 * (type_of(gj) + 3 * type_of(gk)));
                }
            }
    std::vector<double> dict(27 * 243), x(nn * 3), yref;
    srand(7);
    for (auto& v : dict) v = (double)rand() / RAND_MAX - 0.5;
    // (a class that has no neighbour in a direction has zero blocks there, like an assembled row)
    for (int c = 0; c < 27; c++)
        for (int s = 0; s < 27; s++) {
            const int d[3] = {s % 3 - 1, (s / 3) % 3 - 1, s / 9 - 1}, ty[3] = {c % 3, (c / 3) % 3, c / 9};
            bool out = false;
            for (int a = 0; a < 3; a++) out = out || (ty[a] == 0 && d[a] < 0) || (ty[a] == 2 && d[a] > 0);
            if (out) for (int q = 0; q < 9; q++) dict[(size_t)c * 243 + 9 * s + q] = 0.0;
        }
    for (size_t i = 0; i < x.size(); i++) x[i] = sin(0.001 * (double)i);
    int *d_lat; unsigned short* d_cls; double *d_dict, *d_x, *d_y, *d_part;
    CK(hipMalloc(&d_lat, lat.size() * 4)); CK(hipMalloc(&d_cls, cls.size() * 2)); CK(hipMalloc(&d_dict, dict.size() * 8));
    CK(hipMalloc(&d_x, x.size() * 8)); CK(hipMalloc(&d_y, x.size() * 8)); CK(hipMalloc(&d_part, (size_t)ntiles * 8));
    CK(hipMemcpy(d_lat, lat.data(), lat.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_cls, cls.data(), cls.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dict, dict.data(), dict.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_x, x.data(), x.size() * 8, hipMemcpyHostToDevice));
    const size_t shm = 3 * LP * sizeof(double);
    auto run = [&](auto kern, const char* name) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        CK(hipMemset(d_y, 0, x.size() * 8));
        hipLaunchKernelGGL(kern, dim3((ntiles + 7) / 8 * 8), dim3(256), shm, 0, ntiles, d_lat, d_cls, d_dict, d_x, d_y, d_part);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++)
            hipLaunchKernelGGL(kern, dim3((ntiles + 7) / 8 * 8), dim3(256), shm, 0, ntiles, d_lat, d_cls, d_dict, d_x, d_y, d_part);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.4f ms per product (%d back to back)\n", name, ms / reps, reps);
    };
    run(k_stencil_tile<4, 1>, "vector loads of the dictionary, 4 workgroups per CU");
    run(k_stencil_tile<3, 0>, "scalar loads, 3 workgroups per CU");
    run(k_stencil_tile<4, 0>, "scalar loads, 4 workgroups per CU");
    run(k_stencil_tile<3, 2>, "three nodes per thread where the wave has one class, 3 workgroups per CU");
    run(k_stencil_tile<4, 2>, "three nodes per thread where the wave has one class, 4 workgroups per CU");
    // check a sample of rows against the host
    std::vector<double> y(x.size());
    CK(hipMemcpy(y.data(), d_y, y.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t v = 0; v < nn; v += 9973) {
        const int gi = (int)(v % N), gj = (int)((v / N) % N), gk = (int)(v / ((size_t)N * N));
        const int c = type_of(gi) + 3 * (type_of(gj) + 3 * type_of(gk));
        double acc[3] = {0, 0, 0};
        for (int s = 0; s < 27; s++) {
            const int ni = gi + s % 3 - 1, nj = gj + (s / 3) % 3 - 1, nk = gk + s / 9 - 1;
            if (ni < 0 || nj < 0 || nk < 0 || ni >= N || nj >= N || nk >= N) continue;
            const size_t w = (size_t)ni + (size_t)N * (nj + (size_t)N * nk);
            for (int q = 0; q < 3; q++)
                for (int r = 0; r < 3; r++) acc[q] += dict[(size_t)c * 243 + 9 * s + 3 * q + r] * x[w * 3 + r];
        }
        for (int q = 0; q < 3; q++) worst = fmax(worst, fabs(acc[q] - y[v * 3 + q]));
    }
    if (!one_class) printf("largest difference against the host on sampled rows: %.3e\n", worst);
    return 0;
}
