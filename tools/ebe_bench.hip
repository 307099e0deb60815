// tools/ebe_bench.hip -- experiment: element-by-element (matrix-free) application of the hex8
// stiffness operator, w = K u, against the assembled sliced-ELL SpMV (1.17-1.29 ms at config 3).
//   V1  thread per element, registers only, fp64 atomics into w
//   V2  thread per element writes its 24 nodal forces to an element-major buffer, a second kernel
//       gathers per node in fixed order (deterministic, no atomics)
// Build: hipcc -O3 --offload-arch=gfx950 tools/ebe_bench.hip -o tools/ebe_bench ; run: tools/ebe_bench [n]
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

// reference node order of the hex8 (src/m_elems.F90: (-,-,-),(+,-,-),(+,+,-),(-,+,-),(-,-,+),...)
__device__ __constant__ const double SX[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
__device__ __constant__ const double SY[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
__device__ __constant__ const double SZ[8] = {-1, -1, -1, -1, 1, 1, 1, 1};

template <bool ATOMIC>
__global__ void __launch_bounds__(256) k_ebe_hex(int nel, const int* __restrict__ conn, const double* __restrict__ coords,
                                                 const double* __restrict__ E, const double* __restrict__ nu,
                                                 const double* __restrict__ u, double* __restrict__ w,
                                                 double* __restrict__ fbuf)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nel) return;
    constexpr double sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1},
                     sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    int c[8];
    double x[8][3], d[8][3], f[8][3];
#pragma unroll
    for (int a = 0; a < 8; a++) {
        c[a] = conn[(size_t)e * 8 + a];
#pragma unroll
        for (int k = 0; k < 3; k++) { x[a][k] = coords[(size_t)c[a] * 3 + k]; d[a][k] = u[(size_t)c[a] * 3 + k]; f[a][k] = 0.0; }
    }
    const double Ee = E[e], ne = nu[e];
    const double lam = Ee * ne / ((1.0 + ne) * (1.0 - 2.0 * ne)), mu = 0.5 * Ee / (1.0 + ne);
    constexpr double g = 0.57735026918962576451;
#pragma unroll
    for (int gp = 0; gp < 8; gp++) {
        const double xi = sx[gp] * g, et = sy[gp] * g, ze = sz[gp] * g;
        double dn[8][3];
#pragma unroll
        for (int a = 0; a < 8; a++) {
            dn[a][0] = 0.125 * sx[a] * (1.0 + et * sy[a]) * (1.0 + ze * sz[a]);
            dn[a][1] = 0.125 * sy[a] * (1.0 + xi * sx[a]) * (1.0 + ze * sz[a]);
            dn[a][2] = 0.125 * sz[a] * (1.0 + xi * sx[a]) * (1.0 + et * sy[a]);
        }
        double J[3][3] = {}, G[3][3] = {};   // J[k][m] = d x_m / d xi_k,  G[k][l] = d u_l / d xi_k
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int k = 0; k < 3; k++)
#pragma unroll
                for (int m = 0; m < 3; m++) { J[k][m] += dn[a][k] * x[a][m]; G[k][m] += dn[a][k] * d[a][m]; }
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
                     c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, id = 1.0 / det;
        double Ji[3][3];   // Ji[m][k] = d xi_k / d x_m
        Ji[0][0] = c00 * id; Ji[1][0] = c01 * id; Ji[2][0] = c02 * id;
        Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
        Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
        Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
        Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
        Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
        Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
        double H[3][3];    // H[l][m] = d u_l / d x_m
#pragma unroll
        for (int l = 0; l < 3; l++)
#pragma unroll
            for (int m = 0; m < 3; m++) H[l][m] = Ji[m][0] * G[0][l] + Ji[m][1] * G[1][l] + Ji[m][2] * G[2][l];
        const double tr = H[0][0] + H[1][1] + H[2][2];
        double S[3][3];    // stress * w * det  (w = 1)
#pragma unroll
        for (int l = 0; l < 3; l++)
#pragma unroll
            for (int m = 0; m < 3; m++) S[l][m] = det * (mu * (H[l][m] + H[m][l]) + ((l == m) ? lam * tr : 0.0));
        double T[3][3];    // T[l][k] = sum_m S[l][m] Ji[m][k]
#pragma unroll
        for (int l = 0; l < 3; l++)
#pragma unroll
            for (int k = 0; k < 3; k++) T[l][k] = S[l][0] * Ji[0][k] + S[l][1] * Ji[1][k] + S[l][2] * Ji[2][k];
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int l = 0; l < 3; l++) f[a][l] += T[l][0] * dn[a][0] + T[l][1] * dn[a][1] + T[l][2] * dn[a][2];
    }
    if (ATOMIC) {
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int l = 0; l < 3; l++) unsafeAtomicAdd(&w[(size_t)c[a] * 3 + l], f[a][l]);
    } else {
        // element-major buffer, component-major inside so that consecutive threads write consecutive words
#pragma unroll
        for (int a = 0; a < 8; a++)
#pragma unroll
            for (int l = 0; l < 3; l++) fbuf[((size_t)(a * 3 + l)) * nel + e] = f[a][l];
    }
}

// node gather: thread per node, incident (element, local node) pairs in ascending element order
__global__ void __launch_bounds__(256) k_gather(int nn, int nel, const int* __restrict__ ptr, const unsigned* __restrict__ lst,
                                                const double* __restrict__ fbuf, double* __restrict__ w)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nn) return;
    double s0 = 0, s1 = 0, s2 = 0;
    for (int k = ptr[v]; k < ptr[v + 1]; k++) {
        const unsigned code = lst[k];
        const size_t e = code >> 3, a = code & 7u;
        s0 += fbuf[(a * 3 + 0) * nel + e];
        s1 += fbuf[(a * 3 + 1) * nel + e];
        s2 += fbuf[(a * 3 + 2) * nel + e];
    }
    w[(size_t)v * 3] = s0; w[(size_t)v * 3 + 1] = s1; w[(size_t)v * 3 + 2] = s2;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 149;
    const int np = n + 1, nn = np * np * np, nel = n * n * n;
    std::vector<int> conn((size_t)nel * 8);
    std::vector<double> coords((size_t)nn * 3), u((size_t)nn * 3), E(nel, 3e10), nu(nel, 0.25);
    for (int k = 0; k < np; k++) for (int j = 0; j < np; j++) for (int i = 0; i < np; i++) {
        const size_t v = i + (size_t)np * (j + (size_t)np * k);
        coords[v * 3] = i + 0.1 * sin(1.3 * v); coords[v * 3 + 1] = j + 0.1 * sin(2.1 * v + 1); coords[v * 3 + 2] = k + 0.1 * sin(0.7 * v + 2);
        // linear displacement field + rigid rotation: interior nodal forces must vanish
        u[v * 3] = 1e-3 * coords[v * 3] + 2e-3 * coords[v * 3 + 1]; u[v * 3 + 1] = -2e-3 * coords[v * 3] + 5e-4 * coords[v * 3 + 2];
        u[v * 3 + 2] = 1e-3 * coords[v * 3 + 2];
    }
    std::vector<int> cnt(nn, 0);
    for (int k = 0; k < n; k++) for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) {
        const size_t e = i + (size_t)n * (j + (size_t)n * k);
        auto id = [&](int a, int b, int c) { return (int)((i + a) + (size_t)np * ((j + b) + (size_t)np * (k + c))); };
        const int cc[8] = {id(0, 0, 0), id(1, 0, 0), id(1, 1, 0), id(0, 1, 0), id(0, 0, 1), id(1, 0, 1), id(1, 1, 1), id(0, 1, 1)};
        for (int a = 0; a < 8; a++) { conn[e * 8 + a] = cc[a]; cnt[cc[a]]++; }
    }
    std::vector<int> ptr(nn + 1, 0);
    for (int v = 0; v < nn; v++) ptr[v + 1] = ptr[v] + cnt[v];
    std::vector<unsigned> lst(ptr[nn]);
    std::vector<int> cur(ptr.begin(), ptr.end() - 1);
    for (size_t e = 0; e < (size_t)nel; e++) for (int a = 0; a < 8; a++) lst[cur[conn[e * 8 + a]]++] = (unsigned)(e * 8 + a);

    int *dconn, *dptr; unsigned* dlst; double *dco, *du, *dE, *dnu, *dw1, *dw2, *dfb;
    CK(hipMalloc(&dconn, conn.size() * 4)); CK(hipMalloc(&dptr, ptr.size() * 4)); CK(hipMalloc(&dlst, lst.size() * 4));
    CK(hipMalloc(&dco, coords.size() * 8)); CK(hipMalloc(&du, u.size() * 8)); CK(hipMalloc(&dE, nel * 8)); CK(hipMalloc(&dnu, nel * 8));
    CK(hipMalloc(&dw1, u.size() * 8)); CK(hipMalloc(&dw2, u.size() * 8)); CK(hipMalloc(&dfb, (size_t)nel * 24 * 8));
    CK(hipMemcpy(dconn, conn.data(), conn.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dptr, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dlst, lst.data(), lst.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dco, coords.data(), coords.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(du, u.data(), u.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dE, E.data(), nel * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dnu, nu.data(), nel * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20; float ms;
    const unsigned ge = (nel + 255) / 256, gn = (nn + 255) / 256;
    // V1
    for (int r = 0; r < reps + 2; r++) {
        if (r == 2) CK(hipEventRecord(e0));
        CK(hipMemsetAsync(dw1, 0, u.size() * 8));
        hipLaunchKernelGGL(k_ebe_hex<true>, dim3(ge), dim3(256), 0, 0, nel, dconn, dco, dE, dnu, du, dw1, (double*)nullptr);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("n=%d  V1 (registers + fp64 atomics, incl. memset): %.3f ms per application\n", n, ms / reps);
    // V2
    for (int r = 0; r < reps + 2; r++) {
        if (r == 2) CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_ebe_hex<false>, dim3(ge), dim3(256), 0, 0, nel, dconn, dco, dE, dnu, du, (double*)nullptr, dfb);
        hipLaunchKernelGGL(k_gather, dim3(gn), dim3(256), 0, 0, nn, nel, dptr, dlst, dfb, dw2);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("n=%d  V2 (element buffer + ordered node gather):    %.3f ms per application\n", n, ms / reps);
    // element kernel alone
    for (int r = 0; r < reps + 2; r++) {
        if (r == 2) CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_ebe_hex<false>, dim3(ge), dim3(256), 0, 0, nel, dconn, dco, dE, dnu, du, (double*)nullptr, dfb);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("n=%d  element kernel of V2 alone:                  %.3f ms\n", n, ms / reps);
    std::vector<double> w1(u.size()), w2(u.size());
    CK(hipMemcpy(w1.data(), dw1, u.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(w2.data(), dw2, u.size() * 8, hipMemcpyDeviceToHost));
    double dmax = 0, wmax = 0, interior = 0;
    for (size_t i = 0; i < w1.size(); i++) { dmax = fmax(dmax, fabs(w1[i] - w2[i])); wmax = fmax(wmax, fabs(w2[i])); }
    for (int k = 1; k < n; k++) for (int j = 1; j < n; j++) for (int i = 1; i < n; i++) {
        const size_t v = i + (size_t)np * (j + (size_t)np * k);
        for (int l = 0; l < 3; l++) interior = fmax(interior, fabs(w2[v * 3 + l]));
    }
    printf("max|w| %.3e  max|V1-V2| %.3e  max interior force for a linear field %.3e (patch test)\n", wmax, dmax, interior);
    return 0;
}
