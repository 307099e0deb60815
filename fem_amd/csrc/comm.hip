// fem_amd/csrc/comm.hip -- communicator back ends and the halo-sum kernels.
#include "comm.h"

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>

#include "objects.h"

using namespace fem;

namespace fem {

// ---------------------------------------------------------------- Self
struct SelfComm : Comm {
    void allreduce_sum(double*, int, hipStream_t) override {}
    void exchange(int, const int*, const size_t*, const double*, double*, hipStream_t) override {}
};

// ---------------------------------------------------------------- RCCL (xGMI)
// librccl is loaded lazily so that single-GPU use never touches it.
struct Uid {
    char internal[128];
};
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Uid /*ncclUniqueId by value*/, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // prefer a copy the process already holds (PyTorch-ROCm links its own librccl), so that
        // a launcher using torch.distributed and this library share one RCCL instance
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (api.lib) break;
        }
        for (const char* n : names) {
            if (api.lib) break;
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.lib) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
        api.Send = (decltype(api.Send))dlsym(api.lib, "ncclSend");
        api.Recv = (decltype(api.Recv))dlsym(api.lib, "ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    });
    FEM_CHECK(api.lib && api.GetUniqueId && api.CommInitRank && api.AllReduce && api.Send && api.Recv &&
                  api.GroupStart && api.GroupEnd, 60, "RCCL (librccl.so) is not available");
    return api;
}
#define FEM_NCCL(expr)                                                                               \
    do {                                                                                             \
        int r__ = (expr);                                                                            \
        if (r__ != 0)                                                                                \
            throw ::fem::Error(61, std::string("RCCL: ") +                                           \
                                       (rccl().GetErrorString ? rccl().GetErrorString(r__) : "?") + \
                                       " in " #expr);                                                \
    } while (0)

constexpr int NCCL_FLOAT64 = 8;  // ncclDouble
constexpr int NCCL_SUM = 0;

struct RcclComm : Comm {
    void* nc = nullptr;
    ~RcclComm() override {
        if (nc && rccl().CommDestroy) rccl().CommDestroy(nc);
    }
    void allreduce_sum(double* dbuf, int n, hipStream_t s) override {
        note_allreduce(n);
        FEM_NCCL(rccl().AllReduce(dbuf, dbuf, (size_t)n, NCCL_FLOAT64, NCCL_SUM, nc, s));
    }
    // every GPU of an MI355X node has a direct xGMI link to every other one, so
    // the <=7 neighbours of a brick are one hop each: plain grouped send/recv.
    void exchange(int nneigh, const int* neigh_rank, const size_t* off, const double* sendbuf, double* recvbuf,
                  hipStream_t s) override {
        if (!nneigh) return;
        note_exchange((long long)off[nneigh]);
        FEM_NCCL(rccl().GroupStart());
        for (int i = 0; i < nneigh; i++) {
            const size_t cnt = off[i + 1] - off[i];
            if (!cnt) continue;
            FEM_NCCL(rccl().Send(sendbuf + off[i], cnt, NCCL_FLOAT64, neigh_rank[i], nc, s));
            FEM_NCCL(rccl().Recv(recvbuf + off[i], cnt, NCCL_FLOAT64, neigh_rank[i], nc, s));
        }
        FEM_NCCL(rccl().GroupEnd());
    }
    // the CG's p.Kp all-reduce inside the group of the halo exchange that follows it: one RCCL launch per iteration less
    void exchange_allreduce(int nneigh, const int* neigh_rank, const size_t* off, const double* sendbuf, double* recvbuf,
                            double* red, int nred, hipStream_t s) override {
        note_allreduce(nred);
        note_exchange((long long)off[nneigh]);
        FEM_NCCL(rccl().GroupStart());
        FEM_NCCL(rccl().AllReduce(red, red, (size_t)nred, NCCL_FLOAT64, NCCL_SUM, nc, s));
        for (int i = 0; i < nneigh; i++) {
            const size_t cnt = off[i + 1] - off[i];
            if (!cnt) continue;
            FEM_NCCL(rccl().Send(sendbuf + off[i], cnt, NCCL_FLOAT64, neigh_rank[i], nc, s));
            FEM_NCCL(rccl().Recv(recvbuf + off[i], cnt, NCCL_FLOAT64, neigh_rank[i], nc, s));
        }
        FEM_NCCL(rccl().GroupEnd());
    }
};

// ---------------------------------------------------------------- LocalGroup
struct LocalShared {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long gen = 0;
    std::vector<const double*> sendbuf;
    std::vector<const int*> neigh_rank;
    std::vector<const size_t*> off;
    std::vector<int> nneigh;
    std::vector<std::vector<double>> slots;
    int refs = 0;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long g = gen;
        if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct LocalComm : Comm {
    LocalShared* sh = nullptr;
    ~LocalComm() override {
        bool last;
        { std::lock_guard<std::mutex> lk(sh->mu); last = (--sh->refs == 0); }
        if (last) delete sh;
    }
    void allreduce_sum(double* dbuf, int n, hipStream_t s) override {
        // (as in exchange below: a failing domain still arrives at both barriers and throws afterwards)
        note_allreduce(n);
        std::string err;
        std::vector<double>& mine = sh->slots[rank];
        mine.assign((size_t)n, 0.0);
        if (hipMemcpyAsync(mine.data(), dbuf, sizeof(double) * n, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            err = "local all-reduce: copy to the host failed";
        sh->barrier();
        std::vector<double> sum(n, 0.0);
        for (int r = 0; r < size; r++)
            for (int k = 0; k < n && k < (int)sh->slots[r].size(); k++) sum[k] += sh->slots[r][k];
        sh->barrier();  // everybody has read before anybody overwrites its slot
        FEM_CHECK(err.empty(), 62, err);
        FEM_HIP(hipMemcpyAsync(dbuf, sum.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
        FEM_HIP(hipStreamSynchronize(s));
    }
    void exchange(int nneigh, const int* neigh_rank, const size_t* off, const double* sendbuf, double* recvbuf,
                  hipStream_t s) override {
        // A failure on one domain must not strand its peers in a barrier: the error is recorded, every
        // domain still arrives at both barriers, and the failing one throws afterwards.
        note_exchange(nneigh ? (long long)off[nneigh] : 0);
        std::string err;
        if (hipStreamSynchronize(s) != hipSuccess) err = "local exchange: stream synchronisation failed";
        sh->sendbuf[rank] = sendbuf; sh->neigh_rank[rank] = neigh_rank; sh->off[rank] = off; sh->nneigh[rank] = nneigh;
        sh->barrier();
        for (int i = 0; i < nneigh && err.empty(); i++) {
            const int q = neigh_rank[i];
            const size_t cnt = off[i + 1] - off[i];
            int idx = -1;
            for (int k = 0; k < sh->nneigh[q]; k++)
                if (sh->neigh_rank[q][k] == rank) idx = k;
            if (idx < 0) { err = "local exchange: neighbour lists are not symmetric"; break; }
            if (sh->off[q][idx + 1] - sh->off[q][idx] != cnt) { err = "local exchange: interface sizes differ"; break; }
            if (cnt && hipMemcpyAsync(recvbuf + off[i], sh->sendbuf[q] + sh->off[q][idx], sizeof(double) * cnt,
                                      hipMemcpyDeviceToDevice, s) != hipSuccess)
                err = "local exchange: device-to-device copy failed";
        }
        if (hipStreamSynchronize(s) != hipSuccess && err.empty()) err = "local exchange: stream synchronisation failed";
        sh->barrier();
        FEM_CHECK(err.empty(), 62, err);
    }
};

// ---------------------------------------------------------------- Callbacks (host-staged)
// The two operations are handed to the host program: the data goes through pinned host buffers, the caller's functions run
// on the calling thread between two stream synchronisations.  This is the back end for a host that already owns a message
// passing layer -- the reference is an MPI program (Makefile:6-10; MPI_AllReduce src/main.F90:274, Send/Recv src/m_io.F90:150-237):
// its all-reduce callback is one MPI_Allreduce(MPI_IN_PLACE, ...), its exchange callback a loop of MPI_Sendrecv -- and what
// bench.py --rehearse drives with torch.distributed's gloo, so that the N > 1 code paths run as real processes on one GPU.
struct PinnedBuf {
    double* p = nullptr;
    size_t n = 0;
    ~PinnedBuf() { if (p) (void)hipHostFree(p); }
    double* need(size_t cnt) {
        if (cnt > n) {
            if (p) (void)hipHostFree(p);
            p = nullptr; n = 0;
            const size_t want = std::max<size_t>(cnt, 64);
            FEM_HIP(hipHostMalloc((void**)&p, want * sizeof(double), hipHostMallocDefault));
            n = want;
        }
        return p;
    }
};
struct CallbackComm : Comm {
    void (*ar)(void*, double*, int) = nullptr;
    void (*ex)(void*, int, const int*, const long long*, const double*, double*) = nullptr;
    void* user = nullptr;
    PinnedBuf hred, hsend, hrecv;
    void allreduce_sum(double* dbuf, int n, hipStream_t s) override {
        note_allreduce(n);
        double* h = hred.need((size_t)n);
        FEM_HIP(hipMemcpyAsync(h, dbuf, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        FEM_HIP(hipStreamSynchronize(s));
        ar(user, h, n);
        FEM_HIP(hipMemcpyAsync(dbuf, h, sizeof(double) * n, hipMemcpyHostToDevice, s));
        FEM_HIP(hipStreamSynchronize(s));   // (the pinned buffer is reused by the next call)
    }
    void exchange(int nneigh, const int* neigh_rank, const size_t* off, const double* sendbuf, double* recvbuf,
                  hipStream_t s) override {
        if (!nneigh) return;
        const size_t tot = off[nneigh];
        note_exchange((long long)tot);
        double* hs = hsend.need(tot);
        double* hr = hrecv.need(tot);
        if (tot) FEM_HIP(hipMemcpyAsync(hs, sendbuf, sizeof(double) * tot, hipMemcpyDeviceToHost, s));
        FEM_HIP(hipStreamSynchronize(s));
        std::vector<long long> o(off, off + nneigh + 1);
        ex(user, nneigh, neigh_rank, o.data(), hs, hr);
        if (tot) FEM_HIP(hipMemcpyAsync(recvbuf, hr, sizeof(double) * tot, hipMemcpyHostToDevice, s));
        FEM_HIP(hipStreamSynchronize(s));
    }
};

// ---------------------------------------------------------------- halo kernels
__global__ void __launch_bounds__(256) k_halo_pack(int nslots, const int* __restrict__ nodes, int ncomp, size_t sn,
                                                   size_t sc, const double* __restrict__ vec,
                                                   double* __restrict__ sendbuf)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots * ncomp) return;
    const int k = i / ncomp, c = i - k * ncomp;
    sendbuf[i] = vec[(size_t)nodes[k] * sn + (size_t)c * sc];
}

// canonical order: every sharer adds the same contributions in ascending rank
// order, so all copies of an interface value stay bitwise identical.
__global__ void __launch_bounds__(256) k_halo_unpack(int ninodes, const int* __restrict__ inodes,
                                                     const int* __restrict__ cptr, const int* __restrict__ csrc,
                                                     int ncomp, size_t sn, size_t sc,
                                                     const double* __restrict__ recvbuf, double* __restrict__ vec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ninodes * ncomp) return;
    const int k = i / ncomp, c = i - k * ncomp;
    double* dst = vec + (size_t)inodes[k] * sn + (size_t)c * sc;
    const double own = *dst;
    double acc = 0.0;
    for (int q = cptr[k]; q < cptr[k + 1]; q++) {
        const int src = csrc[q];
        acc += (src < 0) ? own : recvbuf[(size_t)src * ncomp + c];
    }
    *dst = acc;
}

}  // namespace fem

void halo_sum(fem_mesh mesh, double* vec, int ncomp, size_t sn, size_t sc) { halo_sum_on(mesh, vec, ncomp, sn, sc, mesh->ctx->stream); }

void halo_sum_on(fem_mesh mesh, double* vec, int ncomp, size_t sn, size_t sc, hipStream_t s) { halo_sum_h(mesh->halo.get(), vec, ncomp, sn, sc, s, false); }

// the same over any list of shared items (the box levels of the partitioned multigrid keep one Halo per level, mg.hip)
void halo_sum_h(Halo* h, double* vec, int ncomp, size_t sn, size_t sc, hipStream_t s, bool call_when_empty)
{
    if (!h || h->comm->size == 1) return;
    const int nslots = h->neigh_ptr[h->nneigh];
    if (nslots == 0) {
        // the in-process communicator meets at a barrier in every exchange: a domain that shares nothing on this list still has to arrive
        if (call_when_empty) { const size_t z = 0; h->comm->exchange(0, nullptr, &z, nullptr, nullptr, s); }
        return;
    }
    FEM_CHECK((size_t)ncomp <= 6, 63, "halo_sum: too many components");
    hipLaunchKernelGGL(k_halo_pack, dim3(cdiv((size_t)nslots * ncomp, 256)), dim3(256), 0, s, nslots, h->neigh_nodes.p,
                       ncomp, sn, sc, vec, h->sendbuf.p);
    std::vector<size_t> off(h->nneigh + 1);
    for (int i = 0; i <= h->nneigh; i++) off[i] = (size_t)h->neigh_ptr[i] * ncomp;
    h->comm->exchange(h->nneigh, h->neigh_rank.data(), off.data(), h->sendbuf.p, h->recvbuf.p, s);
    hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv((size_t)h->ninodes * ncomp, 256)), dim3(256), 0, s, h->ninodes,
                       h->inodes.p, h->contrib_ptr.p, h->contrib_src.p, ncomp, sn, sc, h->recvbuf.p, vec);
    FEM_HIP(hipGetLastError());
}

void halo_sum_allreduce(fem_mesh mesh, double* vec, int ncomp, size_t sn, size_t sc, double* red, int nred)
{
    Halo* h = mesh->halo.get();
    hipStream_t s = mesh->ctx->stream;
    if (!h) return;
    const int nslots = h->neigh_ptr[h->nneigh];
    if (h->comm->size == 1 || nslots == 0) { h->comm->allreduce_sum(red, nred, s); return; }
    FEM_CHECK((size_t)ncomp <= 6, 63, "halo_sum: too many components");
    hipLaunchKernelGGL(k_halo_pack, dim3(cdiv((size_t)nslots * ncomp, 256)), dim3(256), 0, s, nslots, h->neigh_nodes.p,
                       ncomp, sn, sc, vec, h->sendbuf.p);
    std::vector<size_t> off(h->nneigh + 1);
    for (int i = 0; i <= h->nneigh; i++) off[i] = (size_t)h->neigh_ptr[i] * ncomp;
    h->comm->exchange_allreduce(h->nneigh, h->neigh_rank.data(), off.data(), h->sendbuf.p, h->recvbuf.p, red, nred, s);
    hipLaunchKernelGGL(k_halo_unpack, dim3(cdiv((size_t)h->ninodes * ncomp, 256)), dim3(256), 0, s, h->ninodes,
                       h->inodes.p, h->contrib_ptr.p, h->contrib_src.p, ncomp, sn, sc, h->recvbuf.p, vec);
    FEM_HIP(hipGetLastError());
}

// interface lists -> Halo: pack list, canonical-order unpack lists, owner flags (lowest rank among the sharers)
std::unique_ptr<Halo> halo_build(Comm* c, int nitems, int nneigh, const int* neigh_rank, const int* neigh_ptr, const int* neigh_nodes,
                                 hipStream_t s, const char* who)
{
    std::unique_ptr<Halo> h(new Halo);
    h->comm = c; h->nneigh = nneigh;
    h->neigh_rank.assign(neigh_rank, neigh_rank + nneigh);
    h->neigh_ptr.assign(neigh_ptr, neigh_ptr + nneigh + 1);
    const int nslots = neigh_ptr[nneigh];
    for (int i = 0; i < nneigh; i++) {
        FEM_CHECK(neigh_rank[i] >= 0 && neigh_rank[i] < c->size && neigh_rank[i] != c->rank, 3, std::string(who) + ": bad neighbour rank");
        FEM_CHECK(i == 0 || neigh_rank[i] > neigh_rank[i - 1], 3, std::string(who) + ": neighbour ranks must ascend");
    }
    std::vector<unsigned char> owned((size_t)std::max(nitems, 1), 1);
    std::map<int, std::vector<std::pair<int, int>>> contrib;  // item -> (rank, recv slot)
    for (int i = 0; i < nneigh; i++)
        for (int k = neigh_ptr[i]; k < neigh_ptr[i + 1]; k++) {
            const int v = neigh_nodes[k];
            FEM_CHECK(v >= 0 && v < nitems, 3, std::string(who) + ": id out of range");
            if (neigh_rank[i] < c->rank) owned[v] = 0;
            contrib[v].push_back({neigh_rank[i], k});
        }
    std::vector<int> inodes, cptr{0}, csrc;
    for (auto& kv : contrib) {
        auto lst = kv.second;
        lst.push_back({c->rank, -1});
        std::sort(lst.begin(), lst.end());
        inodes.push_back(kv.first);
        for (auto& pr : lst) csrc.push_back(pr.second);
        cptr.push_back((int)csrc.size());
    }
    h->ninodes = (int)inodes.size();
    h->neigh_nodes.alloc(std::max(nslots, 1));
    h->neigh_nodes.upload(neigh_nodes, nslots, s);
    h->inodes.alloc(std::max<size_t>(inodes.size(), 1)); h->inodes.upload(inodes.data(), inodes.size(), s);
    h->contrib_ptr.alloc(cptr.size()); h->contrib_ptr.upload(cptr.data(), cptr.size(), s);
    h->contrib_src.alloc(std::max<size_t>(csrc.size(), 1)); h->contrib_src.upload(csrc.data(), csrc.size(), s);
    h->owned.alloc(owned.size()); h->owned.upload(owned.data(), owned.size(), s);
    h->sendbuf.alloc(std::max<size_t>((size_t)nslots * 6, 1));
    h->recvbuf.alloc(std::max<size_t>((size_t)nslots * 6, 1));
    FEM_HIP(hipStreamSynchronize(s));
    return h;
}

void mesh_set_halo_impl(fem_mesh mesh, fem_comm comm, int nneigh, const int* neigh_rank, const int* neigh_ptr,
                        const int* neigh_nodes)
{
    FEM_CHECK(mesh && comm && comm->impl, 1, "fem_mesh_set_halo: null handle");
    Comm* c = comm->impl;
    mesh->ctx->bind();
    hipStream_t s = mesh->ctx->stream;
    mesh->halo = halo_build(c, mesh->nnds, nneigh, neigh_rank, neigh_ptr, neigh_nodes, s, "fem_mesh_set_halo");
    // global valence (replaces the whole-mesh MPI_AllReduce of src/main.F90:270-275 by one halo sum)
    halo_sum(mesh, mesh->valence_d.p, 1, 1, 0);
    FEM_HIP(hipStreamSynchronize(s));
}

// ---------------------------------------------------------------- C-level constructors
fem_comm comm_create_self_impl(fem_ctx ctx)
{
    auto* c = new SelfComm; c->ctx = ctx;
    auto* h = new fem_comm_s; h->impl = c;
    return h;
}

void comm_get_unique_id_impl(char* out)
{
    Uid id;
    FEM_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(out, id.internal, 128);
}

fem_comm comm_create_rccl_impl(fem_ctx ctx, int nranks, int rank, const char* uid)
{
    FEM_CHECK(ctx, 1, "fem_comm_create_rccl: null context");
    FEM_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, 2, "fem_comm_create_rccl: bad rank/size");
    ctx->bind();
    std::unique_ptr<RcclComm> c(new RcclComm);
    c->ctx = ctx; c->rank = rank; c->size = nranks;
    Uid id;
    std::memcpy(id.internal, uid, 128);
    FEM_NCCL(rccl().CommInitRank(&c->nc, nranks, id, rank));
    auto* h = new fem_comm_s; h->impl = c.release();
    return h;
}

void comm_create_local_group_impl(int n, const fem_ctx* ctxs, fem_comm* out)
{
    FEM_CHECK(n >= 1, 2, "fem_comm_create_local_group: n must be >= 1");
    auto* sh = new LocalShared;
    sh->n = n; sh->refs = n;
    sh->sendbuf.assign(n, nullptr); sh->neigh_rank.assign(n, nullptr); sh->off.assign(n, nullptr);
    sh->nneigh.assign(n, 0); sh->slots.resize(n);
    for (int r = 0; r < n; r++) {
        auto* c = new LocalComm; c->ctx = ctxs[r]; c->rank = r; c->size = n; c->sh = sh;
        out[r] = new fem_comm_s; out[r]->impl = c;
    }
}

fem_comm comm_create_callbacks_impl(fem_ctx ctx, int nranks, int rank, void (*ar)(void*, double*, int),
                                    void (*ex)(void*, int, const int*, const long long*, const double*, double*), void* user)
{
    FEM_CHECK(ctx, 1, "fem_comm_create_callbacks: null context");
    FEM_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, 2, "fem_comm_create_callbacks: bad rank/size");
    FEM_CHECK(ar && ex, 2, "fem_comm_create_callbacks: both callbacks are needed");
    ctx->bind();
    auto* c = new CallbackComm;
    c->ctx = ctx; c->rank = rank; c->size = nranks; c->ar = ar; c->ex = ex; c->user = user;
    auto* h = new fem_comm_s; h->impl = c;
    return h;
}
