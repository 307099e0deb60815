// tools/fp64_rate.hip -- issue rate of the fp64 vector instructions on gfx950: v_fma_f64 / v_add_f64 / v_mul_f64
// with ILP independent chains per lane, W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/fp64_rate.hip -o fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP, int ILP>
__global__ void __launch_bounds__(256) k(double* out, double a, double b, int iters)
{
    double v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) v[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (OP == 0) v[i] = fma(v[i], a, b);
                else if (OP == 1) v[i] = v[i] + b;
                else if (OP == 2) v[i] = v[i] * a;
                else if (OP == 3) v[i] = fma(v[i], v[(i + 1) % ILP], v[(i + 2) % ILP]);   // three VGPR sources
                else v[i] = v[i] + v[(i + 1) % ILP];                                      // two VGPR sources
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += v[i];
    if (s == 12345.678) out[0] = s;
}
template <int OP, int ILP>
void run(const char* name, int waves_per_simd)
{
    double* out; hipMalloc(&out, 8);
    const int iters = 2000;
    // 256 CUs x 4 SIMDs x waves_per_simd waves: blocks of 256 threads = 4 waves, one per SIMD
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, ILP><<<blocks, 256>>>(out, 1.0000001, 1e-9, 10);
    hipEventRecord(e0);
    k<OP, ILP><<<blocks, 256>>>(out, 1.0000001, 1e-9, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)iters * 8 * ILP * waves_per_simd;      // per SIMD
    const double clk = ms * 1e-3 * 2.4e9;
    printf("%-10s ilp %2d waves/simd %d: %.3f ms  %.2f cycles per wave-instruction  (%.1f Ginst/s/SIMD)\n", name, ILP, waves_per_simd, ms,
           clk / inst, inst / ms * 1e-6);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0, 1>("fma", w); run<0, 4>("fma", w); run<0, 8>("fma", w);
        run<1, 1>("add", w); run<1, 4>("add", w); run<1, 8>("add", w);
        run<2, 4>("mul", w); run<2, 8>("mul", w);
        run<3, 8>("fma3v", w); run<4, 8>("add2v", w);
    }
    return 0;
}
