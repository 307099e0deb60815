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
// exercise the halo path on a single GPU).
#pragma once
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
// the halo sum together with the all-reduce of red[0 .. nred) (Comm::exchange_allreduce), on the mesh's stream
void halo_sum_allreduce(fem_mesh mesh, double* vec, int ncomp, size_t sn, size_t sc, double* red, int nred);
