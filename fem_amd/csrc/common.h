// fem_amd/csrc/common.h -- shared internals of libfem_amd.so (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fem_amd.h"

namespace fem {

constexpr int WAVE = 64;          // CDNA wavefront
constexpr int SLICE = 64;         // block rows per SELL slice = one wavefront
constexpr int MAX_NPE = 8;
constexpr double PENALTY_PARAM = (double)1.0e30f;  // src/m_global.F90:15 (default-real literal)

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& m);

#define FEM_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            throw ::fem::Error(70, std::string("HIP: ") + hipGetErrorString(e__) + " in " #expr " at " + \
                                       __FILE__ + ":" + std::to_string(__LINE__));                      \
    } while (0)

#define FEM_CHECK(cond, code, msg)                        \
    do {                                                  \
        if (!(cond)) throw ::fem::Error((code), (msg));   \
    } while (0)

// Device-memory pool behind DBuf (pool.cpp).  The path re-creates its large objects every run
// (Mat_K: 7 GB at config 3, the symbolic-phase scratch, the multigrid hierarchy); going to the
// driver for each of them costs from a few to >100 ms per step depending on the machine.  Released
// blocks are kept by exact size and handed out again once the work that was in flight at their release has
// completed (one event per registered stream, taken at the release: no device-wide synchronisation); the
// cache is bounded (FEM_AMD_POOL_MAX_GB, default 96) and is emptied when an allocation fails and when a
// context is destroyed.
void pool_register_stream(hipStream_t s);              // a context's streams: what a released block may still be in use by
void pool_unregister_stream(hipStream_t s) noexcept;
void* pool_alloc(size_t bytes);
void pool_release(void* p, size_t bytes) noexcept;
void pool_trim() noexcept;      // give every cached block of the current device back to the driver
size_t pool_cached_bytes();

// device buffer (owned)
template <typename T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    DBuf() = default;
    explicit DBuf(size_t n_) { alloc(n_); }
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DBuf& operator=(DBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DBuf() { release(); }
    void alloc(size_t n_) {
        release();
        n = n_;
        if (n) p = static_cast<T*>(pool_alloc(n * sizeof(T)));
    }
    void release() {
        if (p) pool_release(p, n * sizeof(T));
        p = nullptr; n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
    void upload(const T* h, size_t cnt, hipStream_t s) {
        if (cnt) FEM_HIP(hipMemcpyAsync(p, h, cnt * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload(const std::vector<T>& h, hipStream_t s) {
        if (n < h.size()) alloc(h.size());
        upload(h.data(), h.size(), s);
    }
    void download(T* h, size_t cnt, hipStream_t s) const {
        if (cnt) FEM_HIP(hipMemcpyAsync(h, p, cnt * sizeof(T), hipMemcpyDeviceToHost, s));
    }
    void zero(hipStream_t s) {
        if (n) FEM_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
};

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// One-shot set-up that the HIP runtime keeps per device (a kernel's raised dynamic-LDS limit): first(dev) is true exactly once
// per device, also when several domain threads of one process arrive together.
struct OncePerDevice {
    std::mutex mu;
    unsigned long long done = 0;
    bool first(int dev) {
        std::lock_guard<std::mutex> lk(mu);
        const unsigned long long bit = 1ULL << (dev & 63);
        if (done & bit) return false;
        done |= bit;
        return true;
    }
};

}  // namespace fem

struct fem_ctx_s {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;   // collectives and halo exchanges that run beside the compute stream (KSP overlap)
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;   // cross-stream hand-overs of the same
    hipEvent_t t0 = nullptr, t1 = nullptr;
    int num_cu = 256;
    void bind() const { FEM_HIP(hipSetDevice(device)); }
};
