// fem_amd/csrc/comm.h -- domain communicators.
//
// The reference moves data with MPI inside PETSc (MatAssembly stash, VecScatter,
// KSP reductions) plus explicit MPI_AllReduce / Bcast / Send / Recv
// (SURVEY.md section 2a).  Here a domain only ever needs two operations:
//   * allreduce_sum of a few doubles (CG scalars), and
//   * a neighbour exchange of packed interface values (halo sum).
// Back ends: Self (one domain), Rccl (one process per GPU, ncclSend/ncclRecv
// groups over xGMI + ncclAllReduce, all stream-ordered, no host sync), and
// LocalGroup (n domains driven by n host threads in one process, used to
// exercise the halo path on a single GPU), and Callbacks (host-staged: the two
// operations are handed to the host program's own message passing -- the
// reference is an MPI program, Makefile:6-10 -- through pinned host buffers).
#pragma once
#include <memory>

#include "common.h"

namespace fem {

struct Comm {
    int rank = 0, size = 1;
    fem_ctx ctx = nullptr;
    virtual ~Comm() {}
    // in-place sum over all domains of n doubles in device memory
    virtual void allreduce_sum(double* dbuf, int n, hipStream_t s) = 0;
    // neighbour i sends sendbuf[off[i]..off[i+1]) and receives the same range of recvbuf
    virtual void exchange(int nneigh, const int* neigh_rank, const size_t* off, const double* sendbuf,
                          double* recvbuf, hipStream_t s) = 0;
    // both in one go: the all-reduce of `nred` doubles and the neighbour exchange (RCCL: one group, one launch instead of two;
    // the other back ends: one after the other)
    virtual void exchange_allreduce(int nneigh, const int* neigh_rank, const size_t* off, const double* sendbuf, double* recvbuf,
                                    double* red, int nred, hipStream_t s) {
        allreduce_sum(red, nred, s);
        exchange(nneigh, neigh_rank, off, sendbuf, recvbuf, s);
    }
    virtual bool needs_reduce() const { return size > 1; }
    // what went through this communicator since it was created (fem_comm_get_stats): the tests of the partitioned paths assert
    // on it that no collective grows with the mesh (the reference's whole-mesh MPI_AllReduce, src/main.F90:274)
    long long n_allreduce = 0, max_allreduce = 0, sum_allreduce = 0;   // calls, largest count in doubles, total doubles
    long long n_exchange = 0, max_exchange = 0, sum_exchange = 0;      // calls, largest / total doubles sent by this domain
    void note_allreduce(long long n) { n_allreduce++; sum_allreduce += n; if (n > max_allreduce) max_allreduce = n; }
    void note_exchange(long long n) { n_exchange++; sum_exchange += n; if (n > max_exchange) max_exchange = n; }
};

}  // namespace fem

struct fem_comm_s {
    fem::Comm* impl = nullptr;
};

// halo operations on a mesh with an attached Halo (no-ops without one)
// element (node v, component c) of `vec` lives at vec[v*sn + c*sc]
void halo_sum(fem_mesh mesh, double* vec, int ncomp, size_t sn, size_t sc);
// the same on a given stream (the Krylov solver runs it beside the interior part of the product)
void halo_sum_on(fem_mesh mesh, double* vec, int ncomp, size_t sn, size_t sc, hipStream_t s);
// the same over any Halo (shared items of any kind: the box levels of the partitioned multigrid); call_when_empty: a domain whose
// list is empty still calls the (empty) exchange -- the in-process communicator meets at a barrier there
namespace fem { struct Halo; }
void halo_sum_h(fem::Halo* h, double* vec, int ncomp, size_t sn, size_t sc, hipStream_t s, bool call_when_empty);
std::unique_ptr<fem::Halo> halo_build(fem::Comm* c, int nitems, int nneigh, const int* neigh_rank, const int* neigh_ptr,
                                      const int* neigh_items, hipStream_t s, const char* who);
// the halo sum together with the all-reduce of red[0 .. nred) (Comm::exchange_allreduce), on the mesh's stream
void halo_sum_allreduce(fem_mesh mesh, double* vec, int ncomp, size_t sn, size_t sc, double* red, int nred);
