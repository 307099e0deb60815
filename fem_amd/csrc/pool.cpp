// fem_amd/csrc/pool.cpp -- size-keyed cache of released device blocks (see common.h).
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace fem {

namespace {
std::mutex g_mu;
std::map<std::pair<int, size_t>, std::vector<void*>> g_free;   // (device, bytes) -> blocks
size_t g_cached = 0;

size_t pool_limit()
{
    static const size_t lim = [] {
        const char* e = std::getenv("FEM_AMD_POOL_MAX_GB");
        const double gb = e ? std::atof(e) : 96.0;
        return (size_t)((gb > 0.0 ? gb : 0.0) * 1073741824.0);
    }();
    return lim;
}

void trim_locked(int dev)
{
    for (auto it = g_free.begin(); it != g_free.end();) {
        if (it->first.first == dev) {
            for (void* p : it->second) { (void)hipFree(p); g_cached -= it->first.second; }
            it = g_free.erase(it);
        } else {
            ++it;
        }
    }
}
}  // namespace

void* pool_alloc(size_t bytes)
{
    int dev = 0;
    FEM_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_free.find({dev, bytes});
        if (it != g_free.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            g_cached -= bytes;
            return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {          // make room: drop the cache of this device and try once more
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        { std::lock_guard<std::mutex> lk(g_mu); trim_locked(dev); }
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
        throw Error(70, std::string("HIP: ") + hipGetErrorString(e) + " allocating " + std::to_string(bytes) + " bytes");
    return p;
}

void pool_release(void* p, size_t bytes) noexcept
{
    if (!p) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipFree(p); return; }
    // same guarantee as hipFree: nothing in flight can still touch the block when it is handed out again
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_mu);
    if (bytes > pool_limit()) { (void)hipFree(p); return; }
    // A full cache makes room for the block that is released NOW -- the sizes in use now are the ones asked for next -- by giving its
    // largest blocks back to the driver (down to half the limit, so that this is rare).  Before round 5 a full cache -- the big blocks
    // of earlier problems in a long-lived process -- sent every later release to hipFree and every later request to hipMalloc.
    if (g_cached + bytes > pool_limit()) {
        for (auto it = g_free.end(); it != g_free.begin() && g_cached + bytes > pool_limit() / 2;) {
            --it;
            if (it->first.first != dev) continue;
            while (!it->second.empty() && g_cached + bytes > pool_limit() / 2) {
                (void)hipFree(it->second.back());
                it->second.pop_back();
                g_cached -= it->first.second;
            }
        }
    }
    g_free[{dev, bytes}].push_back(p);
    g_cached += bytes;
}

void pool_trim() noexcept
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_mu);
    trim_locked(dev);
}

size_t pool_cached_bytes()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_cached;
}

}  // namespace fem
