// tools/grid_sync_bench.hip -- cost of a grid-wide barrier on one MI355X with every workgroup resident (the question behind a CG whose
// iterations stay inside one kernel at a million dofs): cooperative_groups grid.sync() and a hand-written generation barrier
// (one agent-scope counter, workgroup 0's wave 0 releases), N workgroups of 256 threads, R barriers back to back.
//   hipcc --offload-arch=gfx950 -O3 -o tools/grid_sync_bench tools/grid_sync_bench.hip && tools/grid_sync_bench [workgroups] [reps]
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256, 3) k_cg_sync(int reps, double* out)
{
    cg::grid_group g = cg::this_grid();
    double v = threadIdx.x;
    for (int r = 0; r < reps; r++) { v = v * 1.0000001 + 1.0; g.sync(); }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
// generation barrier: arrive = one agent-scope add per workgroup; the last arrival resets the count and bumps the generation.
// Bounded spin: a stuck barrier raises *err and everybody leaves.
__device__ __forceinline__ bool grid_barrier(unsigned* count, unsigned* gen, unsigned n, unsigned& my_gen, int* err)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned prev = __hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == n - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000L) { *err = 1; ok = false; break; }
                if (*(volatile int*)err) { ok = false; break; }
            }
        }
        __threadfence();
    }
    my_gen++;
    __syncthreads();
    return ok;
}
__global__ void __launch_bounds__(256, 3) k_own_sync(int reps, unsigned* count, unsigned* gen, int* err, double* out)
{
    unsigned my_gen = 0;
    double v = threadIdx.x;
    for (int r = 0; r < reps; r++) {
        v = v * 1.0000001 + 1.0;
        grid_barrier(count, gen, gridDim.x, my_gen, err);
        if (*(volatile int*)err) break;
    }
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main(int argc, char** argv)
{
    int nwg = argc > 1 ? atoi(argv[1]) : 529, reps = argc > 2 ? atoi(argv[2]) : 2000;
    double* out; unsigned* cnt; int* err;
    CK(hipMalloc(&out, nwg * sizeof(double))); CK(hipMalloc(&cnt, 2 * sizeof(unsigned))); CK(hipMalloc(&err, sizeof(int)));
    CK(hipMemset(cnt, 0, 2 * sizeof(unsigned))); CK(hipMemset(err, 0, sizeof(int)));
    int per_cu = 0; hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_own_sync, 256, 0));
    printf("device %s, %d CUs, %d workgroups of 256 per CU -> %d resident; asked %d\n", pr.name, pr.multiProcessorCount, per_cu, per_cu * pr.multiProcessorCount, nwg);
    if (nwg > per_cu * pr.multiProcessorCount) { printf("too many workgroups\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int pass = 0; pass < 2; pass++) {
        void* args[] = {&reps, &out};
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel((void*)k_cg_sync, dim3(nwg), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("cooperative_groups grid.sync(): %.3f us per barrier (%d workgroups, %d barriers)\n", ms * 1e3 / reps, nwg, reps);
    }
    for (int pass = 0; pass < 2; pass++) {
        CK(hipMemset(cnt, 0, 2 * sizeof(unsigned)));
        unsigned* count = cnt; unsigned* gen = cnt + 1;
        void* args[] = {&reps, &count, &gen, &err, &out};
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel((void*)k_own_sync, dim3(nwg), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        int herr = 0; CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
        printf("generation barrier: %.3f us per barrier (%d workgroups, %d barriers)%s\n", ms * 1e3 / reps, nwg, reps, herr ? "  STUCK" : "");
    }
    return 0;
}
