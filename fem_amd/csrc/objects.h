// fem_amd/csrc/objects.h -- the objects behind the opaque C handles.
#pragma once
#include <memory>

#include "common.h"
#include "elements.h"

namespace fem {

// Elements of one type, contiguous: conn[nel][npe] (local node ids).
struct TypeGroup {
    int type = 0, npe = 0, nip = 0, nel = 0;
    DBuf<int> conn;          // [nel*npe]
    DBuf<int> elem_id;       // [nel] position in the caller's element list
    DBuf<double> E, nu;      // [nel] material per element (mat(el%mat,1:2), src/m_global.F90:254-255)
    DBuf<uint16_t> slot;     // [nel*npe*npe] block-column slot of (node a, node b) in row a; filled by the Mat
};

struct Comm;  // comm.h

// Interface lists of one domain (multi-domain runs).
struct Halo {
    Comm* comm = nullptr;
    int nneigh = 0;
    std::vector<int> neigh_rank, neigh_ptr;   // host copies
    DBuf<int> neigh_nodes;                    // [neigh_ptr[nneigh]] local node ids, per neighbour
    // canonical-order unpack: for interface node k (unique list `inodes`), the
    // contributions sorted by contributing rank: src index into recv buffer, or
    // -1 for "own value".
    int ninodes = 0;
    DBuf<int> inodes;        // [ninodes] unique interface nodes
    DBuf<int> contrib_ptr;   // [ninodes+1]
    DBuf<int> contrib_src;   // [contrib_ptr[ninodes]]  recv-buffer node slot or -1
    DBuf<unsigned char> owned;  // [nnds] 1 if this domain owns the node (lowest rank among sharers)
    DBuf<double> sendbuf, recvbuf;  // sized for max components (6) * interface nodes
};

}  // namespace fem

struct fem_mesh_s {
    fem_ctx ctx = nullptr;
    int pdim = 0, nels = 0, nnds = 0, cpdim = 0;
    std::vector<fem::TypeGroup> groups;
    fem::DBuf<double> coords;     // [nnds*pdim] node-major
    fem::DBuf<int> valence;       // [nnds] number of (non-cohesive) elements touching the node (count_node, src/main.F90:263)
    fem::DBuf<double> valence_d;  // [nnds] global valence as double (halo-summed when multi-domain)
    // host copies kept for the symbolic phase and for traction sides
    std::vector<int> h_eltype, h_eptr, h_conn;
    std::vector<int> elem_group, elem_pos;  // element -> (group index, position in group)
    std::unique_ptr<fem::Halo> halo;
};

struct fem_vec_s {
    fem_ctx ctx = nullptr;
    long long n = 0;
    fem::DBuf<double> d;
};

// Mat_K in sliced block-ELL: block rows = nodes, bs = pdim, slices of 64 rows.
//   bcol[(boff[s]+j)*64 + lane]                      block column (node) ids
//   val [((boff[s]+j)*bs*bs + r*bs + c)*64 + lane]   fp64 entries
struct fem_mat_s {
    fem_mesh mesh = nullptr;
    fem_ctx ctx = nullptr;
    int bs = 0, nbrows = 0, nslices = 0;
    long long nblocks = 0;        // true (unpadded) number of bs x bs blocks
    long long padded_blocks = 0;  // boff[nslices]*64
    fem::DBuf<int> boff;          // [nslices+1]
    fem::DBuf<int> bcol;
    fem::DBuf<double> val;
    fem::DBuf<int> rowlen;        // [nbrows] blocks per block row
    std::vector<int> h_boff, h_rowlen;
    fem::DBuf<double> spmv_partial;  // per-workgroup partials of the fused p.Kp
    int spmv_grid = 0;
    // node -> incident (group, element, local node), code = group<<30 | element<<3 | local node;
    // built by the symbolic phase, sorted (and kept) for the element-by-element product
    fem::DBuf<int> n2e_ptr;
    fem::DBuf<unsigned> n2e;
    bool n2e_sorted = false;
    // MatMult by element-by-element recomputation (`-mat_mult_type ebe`, elem_kernels.hip)
    bool mult_ebe = false;
    std::vector<fem::DBuf<double>> ebe_f;   // per element group: nodal forces [npe*pdim][nel]
    // hex8 / quad4: an element whose +x face is the -x face of the next element of the same wavefront
    // hands those forces over by a cross-lane move instead of storing them (halves the buffer traffic)
    std::vector<fem::DBuf<unsigned char>> ebe_donate;   // per group [nel]
    fem::DBuf<int> ebe_ptr;                 // gather lists without the handed-over entries
    fem::DBuf<unsigned> ebe_lst;
    fem::DBuf<double> ebe_partial;          // internal p.Kp partials when the caller wants none
    int ebe_grid = 0;                       // workgroups of all element groups + those of the penalty rows
    // penalty rows as INSERTed by ApplyKBC: dof index (or -1) and (value held here) - (assembled diagonal)
    fem::DBuf<int> bc_dof;
    fem::DBuf<double> bc_shift;
    fem::DBuf<double> bc_diag;              // assembled diagonal under the rows of bc_dof (undo of a changed list)
    std::vector<int> h_bc;                  // the merged constraint list (one record per node) now applied
    fem::DBuf<int> bc_node;                 // [nbrows] offset of a node's rows in bc_dof/bc_shift, or -1
    int nbc_dof = 0;
    bool bc_saved = false;
    // HIP-event profile of the element-by-element product (fem_mat_set_profiling): four events per product
    int profiling = 0;                      // 0 off, p > 0: every p-th product is bracketed by events
    size_t prod_count = 0;                  // products since profiling was switched on
    std::vector<hipEvent_t> pev;
    size_t pev_used = 0;
    ~fem_mat_s() {
        for (auto e : pev) (void)hipEventDestroy(e);
    }
};

struct fem_ksp_s;
