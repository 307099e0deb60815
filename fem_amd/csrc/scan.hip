// fem_amd/csrc/scan.hip -- three-pass exclusive scan (chunk sums, scan of sums,
// chunk-local scan + offset).  Integer work, symbolic phase only.
#include "scan.h"

namespace fem {

constexpr int CH = 1024;  // elements per workgroup (256 threads x 4)

__global__ void __launch_bounds__(256) k_chunk_sums(const int* __restrict__ in, size_t n, int* __restrict__ sums)
{
    const size_t base = (size_t)blockIdx.x * CH;
    int s = 0;
    for (int k = threadIdx.x; k < CH; k += 256) {
        const size_t i = base + k;
        if (i < n) s += in[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    __shared__ int red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// single workgroup: in-place exclusive scan of the chunk sums; sums[nchunks] = total
__global__ void __launch_bounds__(256) k_scan_sums(int* __restrict__ sums, int nchunks)
{
    __shared__ int sh[256];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nchunks; base += 256) {
        const int i = base + threadIdx.x;
        const int v = (i < nchunks) ? sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int t = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = sh[threadIdx.x];
        if (i < nchunks) sums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nchunks] = carry;
}

__global__ void __launch_bounds__(256) k_chunk_scan(const int* __restrict__ in, int* __restrict__ out, size_t n,
                                                    const int* __restrict__ sums)
{
    // each thread owns 4 consecutive elements of the chunk
    const size_t base = (size_t)blockIdx.x * CH + (size_t)threadIdx.x * 4;
    int v[4], t = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (base + k < n) ? in[base + k] : 0; t += v[k]; }
    __shared__ int sh[256];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int u = (threadIdx.x >= o) ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += u;
        __syncthreads();
    }
    int run = sums[blockIdx.x] + sh[threadIdx.x] - t;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = sums[gridDim.x];
}

size_t exclusive_scan_tmp_ints(size_t n) { return (n + CH - 1) / CH + 2; }

void exclusive_scan_i32(const int* in, int* out, size_t n, int* tmp, hipStream_t s)
{
    if (n == 0) {
        FEM_HIP(hipMemsetAsync(out, 0, sizeof(int), s));
        return;
    }
    const int nch = (int)((n + CH - 1) / CH);
    hipLaunchKernelGGL(k_chunk_sums, dim3(nch), dim3(256), 0, s, in, n, tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, s, tmp, nch);
    hipLaunchKernelGGL(k_chunk_scan, dim3(nch), dim3(256), 0, s, in, out, n, tmp);
    FEM_HIP(hipGetLastError());
}

}  // namespace fem
