// tools/grid_sync_bench.hip -- cost of a grid-wide barrier on one MI355X with every workgroup resident (the question behind a CG whose
// iterations stay inside one kernel at a million dofs): cooperative_groups grid.sync(), a hand-written generation barrier
// (one agent-scope counter, acquire-load polling, two __threadfence()) and the XCD-hierarchical barrier of MI355X_MICROARCH.md's price
// table (row barrier-xcd: per-XCC counter -> the XCD's last arriver releases and adds to a top counter -> it polls the top counter,
// acquires and bumps the XCC's generation word; everybody else polls that word with relaxed sc1 loads + s_sleep and acquires once),
// N workgroups of 256 threads, R barriers back to back; the hierarchical one also with a 128-B record per workgroup published before
// every barrier and a neighbour's record checked after it.
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
// ---- XCD-hierarchical barrier ----------------------------------------------------------------------------------------------------
// words (each on a 128-B line of its own): xcnt[8] arrivals per XCC (monotonic), top arrivals of XCC leaders (monotonic), xgen[8]
// generation per XCC, xmem[8] members per XCC (census, first use), nx number of XCCs that hold workgroups
struct XcdBar { unsigned* xcnt; unsigned* top; unsigned* xgen; unsigned* xmem; };
constexpr int LINE = 32;   // words per 128-B line
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu; }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// gen: number of barriers this workgroup has passed; members: workgroups on this XCC; nxcc: XCCs in use.  publish: this workgroup
// has stored data others will read (plain stores: the XCD's leader writes the L2 back)
__device__ __forceinline__ bool xcd_barrier(const XcdBar& b, unsigned xcc, unsigned members, unsigned nxcc, unsigned& gen, int* err)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave's stores have reached this XCD's L2
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned target = gen + 1;
        const unsigned prev = __hip_atomic_fetch_add(b.xcnt + xcc * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        if (prev == members * target - 1) {          // the XCD's last arriver: leader of this round
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_relaxed(b.top) < nxcc * target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000L) { *err = 1; ok = false; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(b.xgen + xcc * LINE, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(b.xgen + xcc * LINE) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000L) { *err = 1; ok = false; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    gen++;
    __syncthreads();
    return ok;
}
// census (once per launch): members per XCC and the number of XCCs in use, agreed upon through a plain counter barrier
__device__ __forceinline__ void xcd_census(const XcdBar& b, unsigned* census_done, unsigned nwg, unsigned& xcc, unsigned& members, unsigned& nxcc)
{
    __shared__ unsigned s_x[3];
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        __hip_atomic_fetch_add(b.xmem + x * LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(census_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        for (long spins = 0; ld_relaxed(census_done) < nwg && spins < 4000000L; spins++) __builtin_amdgcn_s_sleep(2);   // (bounded: a grid that is not resident ends with wrong counts and a STUCK barrier, not a hang)
        unsigned n = 0;
        for (int i = 0; i < 8; i++) n += ld_relaxed(b.xmem + i * LINE) != 0;
        s_x[0] = x; s_x[1] = ld_relaxed(b.xmem + x * LINE); s_x[2] = n;
    }
    __syncthreads();
    xcc = s_x[0]; members = s_x[1]; nxcc = s_x[2];
}
template <bool PUBLISH>
__global__ void __launch_bounds__(256, 3) k_xcd_sync(int reps, XcdBar b, unsigned* census_done, int* err, double* rec, unsigned* bad, double* out)
{
    unsigned xcc, members, nxcc, gen = 0;
    xcd_census(b, census_done, gridDim.x, xcc, members, nxcc);
    double v = threadIdx.x;
    unsigned nbad = 0;
    for (int r = 0; r < reps; r++) {
        v = v * 1.0000001 + 1.0;
        if (PUBLISH && threadIdx.x < 16) rec[(size_t)blockIdx.x * 16 + threadIdx.x] = (double)(r + 1) + threadIdx.x;   // a 128-B record, plain stores
        if (!xcd_barrier(b, xcc, members, nxcc, gen, err)) break;
        if (PUBLISH && threadIdx.x < 16) {
            const unsigned peer = (blockIdx.x + 17u * (r + 1)) % gridDim.x;                        // somebody else's record, every lane its word
            nbad += rec[(size_t)peer * 16 + threadIdx.x] != (double)(r + 1) + threadIdx.x;
        }
        if (PUBLISH) { if (!xcd_barrier(b, xcc, members, nxcc, gen, err)) break; }                  // (nobody overwrites before everybody has read)
    }
    if (PUBLISH && nbad) atomicAdd(bad, nbad);
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
    // XCD-hierarchical barrier
    unsigned* words; CK(hipMalloc(&words, (8 + 1 + 8 + 8 + 2) * LINE * sizeof(unsigned)));
    double* rec; CK(hipMalloc(&rec, (size_t)nwg * 16 * sizeof(double)));
    XcdBar xb{words, words + 8 * LINE, words + 9 * LINE, words + 17 * LINE};
    unsigned* census = words + 25 * LINE; unsigned* bad = words + 26 * LINE;
    for (int publish = 0; publish < 2; publish++)
        for (int pass = 0; pass < 2; pass++) {
            CK(hipMemset(words, 0, (8 + 1 + 8 + 8 + 2) * LINE * sizeof(unsigned))); CK(hipMemset(err, 0, sizeof(int)));
            CK(hipMemset(rec, 0, (size_t)nwg * 16 * sizeof(double)));
            CK(hipEventRecord(e0));
            if (publish) hipLaunchKernelGGL(k_xcd_sync<true>, dim3(nwg), dim3(256), 0, 0, reps, xb, census, err, rec, bad, out);
            else hipLaunchKernelGGL(k_xcd_sync<false>, dim3(nwg), dim3(256), 0, 0, reps, xb, census, err, rec, bad, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            int herr = 0; unsigned hbad = 0, mem[8];
            CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, sizeof(unsigned), hipMemcpyDeviceToHost));
            for (int i = 0; i < 8; i++) CK(hipMemcpy(&mem[i], xb.xmem + i * LINE, sizeof(unsigned), hipMemcpyDeviceToHost));
            const int nbar = publish ? 2 * reps : reps;
            printf("XCD-hierarchical barrier%s: %.3f us per barrier (%d workgroups [%u %u %u %u %u %u %u %u per XCC], %d barriers)%s",
                   publish ? ", 128-B record per workgroup published and a peer's read back" : "", ms * 1e3 / nbar, nwg, mem[0], mem[1], mem[2], mem[3],
                   mem[4], mem[5], mem[6], mem[7], nbar, herr ? "  STUCK" : "");
            if (publish) printf("  stale words read: %u", hbad);
            printf("\n");
        }
    return 0;
}
