// tools/pmc_calib.hip -- known-byte-count kernels to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the product uses
// (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 for 16 B/lane streams; other
// widths must be calibrated in one's own pattern).  Each kernel moves exactly
// 4 GiB once (buffers far larger than the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void calib_read_b64(const double* __restrict__ a, double* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0; for (; i < n; i += st) s += a[i];
    if (s == 1.2345e-300) out[0] = s;
}
__global__ void calib_read_b128(const double2* __restrict__ a, double* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double s = 0; for (; i < n; i += st) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e-300) out[0] = s;
}
__global__ void calib_write_b64(double* __restrict__ a, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) a[i] = 1.0;
}
int main() {
    const size_t bytes = (size_t)4 << 30;
    double *a, *o; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 0, bytes)); CK(hipDeviceSynchronize());
    for (int r = 0; r < 3; r++) {
        calib_read_b64<<<8192, 256>>>(a, o, bytes / 8);
        calib_read_b128<<<8192, 256>>>((const double2*)a, o, bytes / 16);
        calib_write_b64<<<8192, 256>>>(a, bytes / 8);
    }
    CK(hipDeviceSynchronize());
    printf("moved %zu bytes per kernel\n", bytes);
    return 0;
}
