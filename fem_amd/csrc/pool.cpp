// fem_amd/csrc/pool.cpp -- size-keyed cache of released device blocks (see common.h).
//
// Release costs nothing on the device: a released block is stamped with the device's release count and goes into the cache at once
// (before round 6 every release called hipDeviceSynchronize(): a host that frees inside its increment loop stalled the device each
// time).  What orders a block's reuse behind the work that could still touch it is a BATCH, taken when a request first meets a block
// no batch covers yet: one event per stream the library runs on that device (the contexts register their two streams), recorded at
// that moment -- later than the release of every block stamped so far, so it covers them all -- and waited for by the request.  A
// loop that frees and re-creates its temporaries pays one batch per turn, not two event records per block (a first version did that:
// config 5's Newton loop, a chain of small launches on an idle device, went from 9.3 to 11.6 ms) -- and nothing at all where the
// streams are idle when the block is asked for again (two stream queries).  Requests prefer blocks that a
// completed batch covers.  With more than MAX_EV streams registered on a device (many in-process domains) a batch is a device
// synchronisation.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
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
    unsigned long long seq = 0;     // the device's release count when the block came back
};
struct Batch {                      // covers every block of its device with seq <= this one's
    unsigned long long seq = 0;
    hipEvent_t ev[MAX_EV] = {nullptr, nullptr, nullptr, nullptr};
    int nev = 0;
    bool whole_device = false;      // no events: the wait is hipDeviceSynchronize()
};
struct DeviceState {
    std::vector<hipStream_t> streams;   // streams of the live contexts
    unsigned long long seq = 0;         // releases so far
    unsigned long long done_upto = 0;   // every block stamped <= this is free of in-flight work
    std::deque<Batch> batches;          // ascending seq, none of them known to be complete
};
std::mutex g_mu;
std::map<std::pair<int, size_t>, std::vector<Block>> g_free;   // (device, bytes) -> blocks
std::map<int, DeviceState> g_dev;
std::vector<hipEvent_t> g_events;                              // events not in use
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

void drop_batch_locked(Batch& b)
{
    for (int i = 0; i < b.nev; i++) g_events.push_back(b.ev[i]);
    b.nev = 0;
}
// hipFree waits for the device itself
void free_block_locked(Block& b, size_t bytes)
{
    static const bool verbose = std::getenv("FEM_AMD_POOL_VERBOSE") && std::atoi(std::getenv("FEM_AMD_POOL_VERBOSE")) != 0;
    if (verbose) std::fprintf(stderr, "[fem_amd pool] hipFree %zu B\n", bytes);
    (void)hipFree(b.p);
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
// a batch that covers everything released on this device so far (taken now)
Batch take_batch_locked(DeviceState& d)
{
    Batch b;
    b.seq = d.seq;
    b.whole_device = d.streams.empty() || d.streams.size() > (size_t)MAX_EV;
    if (!b.whole_device) {
        for (hipStream_t s : d.streams) {
            hipEvent_t ev = nullptr;
            if (!g_events.empty()) { ev = g_events.back(); g_events.pop_back(); }
            else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); b.whole_device = true; break; }
            if (hipEventRecord(ev, s) != hipSuccess) { (void)hipGetLastError(); g_events.push_back(ev); b.whole_device = true; break; }
            b.ev[b.nev++] = ev;
        }
    }
    return b;
}
}  // namespace

void pool_register_stream(hipStream_t s)
{
    int dev = 0;
    FEM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dev].streams.push_back(s);
}
void pool_unregister_stream(hipStream_t s) noexcept
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto& v = g_dev[dev].streams;
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
            DeviceState& d = g_dev[dev];
            // the oldest block: the one most likely to be covered by a completed batch
            size_t pick = 0;
            for (size_t i = 1; i < v.size(); i++)
                if (v[i].seq < v[pick].seq) pick = i;
            const Block b = v[pick];
            v.erase(v.begin() + (long)pick);
            g_cached -= bytes;
            if (b.seq <= d.done_upto) return b.p;
            // the first batch that covers the block, or a new one
            Batch wait;
            bool found = false;
            for (const Batch& q : d.batches)
                if (q.seq >= b.seq) { wait = q; found = true; break; }
            if (!found) {
                // nothing in flight on any of the device's streams (the usual state behind a call that ended on a stream
                // synchronisation -- a dot product, a norm read back): every block released so far is clean, no events needed
                bool idle = !d.streams.empty() && d.streams.size() <= (size_t)MAX_EV;
                const unsigned long long upto = d.seq;
                for (size_t i = 0; idle && i < d.streams.size(); i++)
                    if (hipStreamQuery(d.streams[i]) != hipSuccess) { (void)hipGetLastError(); idle = false; }
                if (idle) { d.done_upto = upto; return b.p; }
                // Work is in flight and the only cached blocks of this size came back too recently to be known idle.  A host that
                // runs ahead of the device (a Newton loop enqueueing small launches) would stop here and drain its queue for every
                // temporary it creates -- config 5's connected strip took 45 instead of 9 ms inside bench.py, where the sizes of its
                // temporaries had exactly as many cached blocks as the loop keeps in flight.  A small block is cheaper to get from the
                // driver (5-30 us) than to wait for: the old one goes back into the cache, the next request finds it clean.
                if (bytes <= ((size_t)4 << 20) && v.size() < 64) {
                    v.push_back(b);
                    g_cached += bytes;
                    goto fresh;
                }
                wait = take_batch_locked(d); d.batches.push_back(wait);
            }
            lk.unlock();
            for (int i = 0; i < wait.nev; i++) (void)hipEventSynchronize(wait.ev[i]);
            if (wait.whole_device) (void)hipDeviceSynchronize();
            lk.lock();
            DeviceState& d2 = g_dev[dev];
            if (wait.seq > d2.done_upto) d2.done_upto = wait.seq;
            while (!d2.batches.empty() && d2.batches.front().seq <= d2.done_upto) {   // (events go back once: the deque owns them)
                drop_batch_locked(d2.batches.front());
                d2.batches.pop_front();
            }
            return b.p;
        }
    }
fresh:
    void* p = nullptr;
    // FEM_AMD_POOL_VERBOSE=1 (diagnostic): every request the cache cannot serve, with what the driver took for it
    static const bool verbose = std::getenv("FEM_AMD_POOL_VERBOSE") && std::atoi(std::getenv("FEM_AMD_POOL_VERBOSE")) != 0;
    const auto tv0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, bytes);
    if (verbose)
        std::fprintf(stderr, "[fem_amd pool] hipMalloc %zu B: %.1f us (cached %.2f GB)\n", bytes,
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tv0).count(), (double)g_cached / 1073741824.0);
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
    // FEM_AMD_POOL_SYNC=1 (diagnostic): the release of rounds 1-5 -- synchronise the device; the block is clean at once
    static const bool sync_env = std::getenv("FEM_AMD_POOL_SYNC") && std::atoi(std::getenv("FEM_AMD_POOL_SYNC")) != 0;
    if (sync_env) (void)hipDeviceSynchronize();
    b.seq = sync_env ? 0 : ++g_dev[dev].seq;
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
    DeviceState& d = g_dev[dev];
    d.done_upto = d.seq;
    for (Batch& b : d.batches) drop_batch_locked(b);
    d.batches.clear();
}

size_t pool_cached_bytes()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_cached;
}

}  // namespace fem
