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
    // assembly of a mesh whose elements all have the blocks of the first one, bit for bit (elem_kernels.hip, AsmUniform):
    // 0 no assembly yet, 4 one assembly done, 1 the second assembly is comparing, 2 uniform, 3 not
    int ke_state = 0;
    DBuf<double> ke_ref;     // [npe][npe][pd][pd]
    DBuf<int> ke_diff;       // [1]
};

struct Comm;  // comm.h
struct MatPattern;

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
    // cohesive elements (type 5, el%nlMat >= 1): kept apart from the solid groups -- they add nothing to Mat_K, only
    // the tangent / traction of the Newton callbacks (src/m_global.F90:127-147,186-205) and their node pairs to the
    // sparsity.  Set by fem_mesh_set_cohesive before the first Mat is created.
    int ncoh = 0;
    fem::DBuf<int> coh_conn;        // [ncoh*4]
    fem::DBuf<double> coh_props;    // [ncoh*6] the element's cohMats(nlMat) row
    std::vector<int> h_coh_conn;
    std::vector<int> coh_colour_ptr;   // the cohesive elements are kept colour by colour (no shared node inside a colour): [ncolours + 1]
    std::shared_ptr<fem::MatPattern> pattern;   // symbolic phase result, built by the first fem_mat_create_aij
    // the merged boundary-condition records of the last fem_enforce_bc_force on the device (a Newton loop passes the same list at
    // every residual evaluation: merging, two uploads and the wait for them were 50 us of each, round 6)
    struct BcForceCache {
        std::vector<int> bcnode; std::vector<double> bcval;   // the host's lists as they were passed
        fem::DBuf<int> d_bc; fem::DBuf<double> d_bcval;
        int nm = 0;
    } bc_force;
};

struct fem_vec_s {
    fem_ctx ctx = nullptr;
    long long n = 0;
    fem::DBuf<double> d;
};

namespace fem {

struct EbeTilePlan;   // ebe_tile.hip
struct MgBoxes;       // mg.h
// what the direct solve of an unsymmetric Mat (BandLU::setup, direct.hip) derives from the sparsity alone, on the host: the connected
// components of the node graph (small independent blocks are factorised one wavefront each) or the reverse Cuthill-McKee numbering and
// its bandwidth.  Kept on the pattern: a Newton loop sets the solver up again at every iteration, on the same sparsity.
struct LuSymbolic {
    bool comp_env = true;                     // FEM_AMD_LU_COMPONENTS at the time it was built
    bool batched = false;
    int nc = 0;
    std::vector<int> cptr, cnodes, loc;
    std::vector<long long> foff;
    std::vector<int> perm;
    long long bwn = 0;
    bool pcr_failed = false;                  // a solve by parallel cyclic reduction has failed its backward-error test on this sparsity
    DBuf<int> d_cptr, d_cnodes, d_loc;        // the components on the device (batched factorisation)
    DBuf<long long> d_foff;
    DBuf<int> d_perm, d_perm_dof;             // the numbering on the device (node, dof): uploaded by the first set-up on this sparsity
};

// Everything about Mat_K that depends on the connectivity alone (the product of MatCreateAIJ's
// preallocation, src/main.F90:153-156, and of the implicit pattern growth of MatSetValues): built once
// per mesh by the symbolic phase and shared by every Mat created from that mesh -- the reference, too,
// creates Mat_K once (src/main.F90:154) and re-runs only the element loop.
//   bcol[(boff[s]+j)*64 + lane]   block column (node) ids of the sliced block-ELL layout
struct MatPattern {
    int bs = 0, nbrows = 0, nslices = 0;
    long long nblocks = 0;        // true (unpadded) number of bs x bs blocks
    long long padded_blocks = 0;  // boff[nslices]*64
    int maxw = 0;                 // widest slice (blocks per row)
    DBuf<int> boff;               // [nslices+1]
    DBuf<int> bcol;
    DBuf<int> rowlen;             // [nbrows] blocks per block row
    std::vector<int> h_boff, h_rowlen;
    // node -> incident (group, element, local node), code = group<<30 | element<<3 | local node,
    // ascending inside every node's list
    DBuf<int> n2e_ptr;
    DBuf<unsigned> n2e;
    // row-owner assembly (elem_kernels.hip): slices in which two lanes of one row would add the same
    // local column in the same pass (same node at the same local index of two incident elements)
    DBuf<unsigned char> slice_serial;   // [nslices]
    int max_valence = 0;
    // MatMult by element-by-element recomputation: hand-over flags, workgroup classes, gather lists
    bool ebe_ready = false;
    struct EbePlan {
        DBuf<unsigned char> donate;          // [nel] hex8 / quad4: the element hands its +x face to the next lane
        DBuf<int> blk_affine, blk_general;   // hex8: workgroups whose elements are all parallelepipeds / the others
        int n_affine = 0, n_general = 0;
    };
    std::vector<EbePlan> ebe_plan;     // per group
    std::shared_ptr<EbeTilePlan> ebe_tile;   // one-pass product by tiles (ebe_tile.hip); null: not built or not applicable
    std::shared_ptr<LuSymbolic> lu_sym;      // BandLU's host-side symbolic work (direct.hip); null: not built yet
    std::shared_ptr<MgBoxes> mg_boxes;       // -pc_type mg: the level-1 aggregates of this mesh (mg.h); null: not built yet
    bool ebe_tile_tried = false;
    DBuf<int> ebe_ptr;
    DBuf<unsigned> ebe_lst;
    int ebe_grid = 0;             // workgroups of all element groups + those of the node gather
    double t_symbolic_ms = 0.0;   // wall time of the symbolic phase that built this pattern
};

}  // namespace fem

// Mat_K in sliced block-ELL: block rows = nodes, bs = pdim, slices of 64 rows.
//   val [((boff[s]+j)*bs*bs + r*bs + c)*64 + lane]   fp64 entries
struct fem_mat_s {
    std::shared_ptr<fem::MatPattern> pat;
    fem_mesh mesh = nullptr;
    fem_ctx ctx = nullptr;
    int bs = 0, nbrows = 0, nslices = 0;
    long long nblocks = 0, padded_blocks = 0;
    // views into the shared pattern
    fem::DBuf<int>&boff, &bcol, &rowlen;
    std::vector<int>&h_boff, &h_rowlen;
    fem::DBuf<int>& n2e_ptr;
    fem::DBuf<unsigned>& n2e;
    fem::DBuf<double> val;
    unsigned long long state = 0;   // bumped by whatever changes the values (as PETSc's object state): a KSP that kept the solution of
                                    // an identical right-hand side (-ksp_reuse_solution) gives it out only for an unchanged operator
    // MatZeroEntries is lazy: the row-owner assembly overwrites every entry (padding included), so a
    // zeroed matrix that is assembled next never pays for the memset; every other reader goes through
    // values(), which materialises the zeros first.
    bool lazy_zero = false;
    bool unsymmetric = false;     // cohesive tangents were added (src/m_global.F90:830-831 scales only one of two terms)
    double* values() {
        if (lazy_zero) { val.zero(ctx->stream); lazy_zero = false; }
        return val.p;
    }
    fem::DBuf<double> spmv_partial;  // per-workgroup partials of the fused p.Kp
    int spmv_grid = 0;
    // MatMult by element-by-element recomputation (`-mat_mult_type ebe`, elem_kernels.hip)
    bool mult_ebe = false;
    std::vector<fem::DBuf<double>> ebe_f;   // per element group: nodal forces [npe*pdim][nel]
    fem::DBuf<double> ebe_partial;          // internal p.Kp partials when the caller wants none
    int ebe_grid = 0;
    bool sampling = false;                  // the product in flight is one the profile samples
    bool ebe_use_tile = false;              // `ebe`: one-pass tile product when the mesh allows it; `ebe_gather`: never
    // the one-pass tile product can deliver the sum of its p.Kp partials itself (its last workgroup adds them): the Krylov
    // solver points red_out at its scalar for the length of a solve
    double* red_out = nullptr;
    fem::DBuf<unsigned> red_counter;
    // penalty rows as INSERTed by ApplyKBC: dof index (or -1) and (value held here) - (assembled diagonal)
    fem::DBuf<int> bc_dof;
    fem::DBuf<double> bc_shift;
    fem::DBuf<double> bc_diag;              // assembled diagonal under the rows of bc_dof (undo of a changed list)
    std::vector<int> h_bc;                  // the merged constraint list (one record per node) now applied
    std::vector<int> h_bc_dev;              // the merged list that d_bc_dev holds (a Newton loop applies the same list to every Jacobian)
    fem::DBuf<int> d_bc_dev;
    fem::DBuf<int> bc_node;                 // [nbrows] offset of a node's rows in bc_dof/bc_shift, or -1
    int nbc_dof = 0;
    bool bc_saved = false;
    // HIP-event profile of the element-by-element product (fem_mat_set_profiling): four events per product
    int profiling = 0;                      // 0 off, p > 0: every p-th product is bracketed by events
    size_t prod_count = 0;                  // products since profiling was switched on
    std::vector<hipEvent_t> pev;
    size_t pev_used = 0;
    explicit fem_mat_s(std::shared_ptr<fem::MatPattern> p)
        : pat(std::move(p)), bs(pat->bs), nbrows(pat->nbrows), nslices(pat->nslices), nblocks(pat->nblocks),
          padded_blocks(pat->padded_blocks), boff(pat->boff), bcol(pat->bcol), rowlen(pat->rowlen), h_boff(pat->h_boff),
          h_rowlen(pat->h_rowlen), n2e_ptr(pat->n2e_ptr), n2e(pat->n2e) {}
    ~fem_mat_s() {
        for (auto e : pev) (void)hipEventDestroy(e);
    }
};

struct fem_ksp_s;
