// fem_amd/csrc/pool.cpp -- size-keyed cache of released device blocks (see common.h).
//
// Release is stream-ordered, not a device-wide stall: a released block takes one event per stream the library runs on that device
// (the contexts register their two streams), recorded at the moment of the release -- everything that can still touch the block was
// submitted before them -- and goes into the cache at once.  A later request of that size prefers a block whose events have
// completed and otherwise waits for the events of the one it takes: only for the work that was in flight when that block was
// released, never for the device.  (Before round 6 every release called hipDeviceSynchronize(): a host that frees inside its
// increment loop stalled the device each time.)  With more than MAX_EV streams registered on a device (many in-process domains)
// a block is stamped with the release count instead and the first request that meets such a block synchronises the device once,
// which clears every block released before that moment.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace fem {

namespace {
constexpr int MAX_EV = 4;
struct Block {
    void* p = nullptr;
    hipEvent_t ev[MAX_EV] = {nullptr, nullptr, nullptr, nullptr};
    int nev = 0;
    unsigned long long epoch = 0;   // != 0: released without events; clean once g_clean[dev] >= epoch
};
std::mutex g_mu;
std::map<std::pair<int, size_t>, std::vector<Block>> g_free;   // (device, bytes) -> blocks
std::map<int, std::vector<hipStream_t>> g_streams;             // device -> streams of the live contexts
std::map<int, unsigned long long> g_clean;                     // device -> every block stamped <= this is clean
std::vector<hipEvent_t> g_events;                              // events not in use
unsigned long long g_epoch = 0;
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

void give_back_events_locked(Block& b)
{
    for (int i = 0; i < b.nev; i++) g_events.push_back(b.ev[i]);
    b.nev = 0;
}
// hipFree waits for the device itself: no event of the block has to be looked at
void free_block_locked(Block& b, size_t bytes)
{
    (void)hipFree(b.p);
    give_back_events_locked(b);
    g_cached -= bytes;
}
void trim_locked(int dev)
{
    for (auto it = g_free.begin(); it != g_free.end();) {
        if (it->first.first == dev) {
            for (Block& b : it->second) free_block_locked(b, it->first.second);
            it = g_free.erase(it);
        } else {
            ++it;
        }
    }
}
bool ready_locked(const Block& b, int dev)
{
    if (b.epoch && g_clean[dev] < b.epoch) return false;
    for (int i = 0; i < b.nev; i++)
        if (hipEventQuery(b.ev[i]) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}
}  // namespace

void pool_register_stream(hipStream_t s)
{
    int dev = 0;
    FEM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams[dev].push_back(s);
}
void pool_unregister_stream(hipStream_t s) noexcept
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto& v = g_streams[dev];
    v.erase(std::remove(v.begin(), v.end(), s), v.end());
}

void* pool_alloc(size_t bytes)
{
    int dev = 0;
    FEM_HIP(hipGetDevice(&dev));
    {
        std::unique_lock<std::mutex> lk(g_mu);
        auto it = g_free.find({dev, bytes});
        if (it != g_free.end() && !it->second.empty()) {
            auto& v = it->second;
            size_t pick = v.size() - 1;
            for (size_t i = v.size(); i-- > 0;)
                if (ready_locked(v[i], dev)) { pick = i; break; }
            Block b = v[pick];
            v.erase(v.begin() + (long)pick);
            g_cached -= bytes;
            const bool need_dev = b.epoch && g_clean[dev] < b.epoch;
            const unsigned long long upto = g_epoch;
            lk.unlock();
            // what was in flight when this block was released: its events, or (no events taken) the device as a whole
            for (int i = 0; i < b.nev; i++) (void)hipEventSynchronize(b.ev[i]);
            if (need_dev) (void)hipDeviceSynchronize();
            lk.lock();
            if (need_dev) g_clean[dev] = std::max(g_clean[dev], upto);
            give_back_events_locked(b);
            return b.p;
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
                free_block_locked(it->second.back(), it->first.second);
                it->second.pop_back();
            }
        }
    }
    Block b;
    b.p = p;
    const auto& streams = g_streams[dev];
    bool stamped = streams.empty() || streams.size() > (size_t)MAX_EV;
    if (!stamped) {
        for (hipStream_t s : streams) {
            hipEvent_t ev = nullptr;
            if (!g_events.empty()) { ev = g_events.back(); g_events.pop_back(); }
            else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); stamped = true; break; }
            if (hipEventRecord(ev, s) != hipSuccess) { (void)hipGetLastError(); g_events.push_back(ev); stamped = true; break; }
            b.ev[b.nev++] = ev;
        }
    }
    if (stamped) b.epoch = ++g_epoch;   // (the events taken so far stay with the block: waiting for them as well is harmless)
    g_free[{dev, bytes}].push_back(b);
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
