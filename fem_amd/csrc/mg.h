// fem_amd/csrc/mg.h -- box-aggregation multigrid preconditioner (`-pc_type mg`).
//
// "better preconditioner behind the same KSP surface ... to cut
// CG iterations once SpMV is at roofline".  Unsmoothed-aggregation multigrid whose
// aggregates are the cells of a uniform box grid laid over the node coordinates
// (level 1: about 4 nodes per box edge; every further level merges 2x2x2 boxes, down to <= 5 boxes per direction).
// Because boxes are a structured index space even on unstructured meshes, every
// coarse operator is a dense 27-point block stencil: no coarse symbolic phase,
// trivially parallel Galerkin products, coalesced box-major storage.
//   P     rigid-body modes of each box (translations + rotations about the box centre: 6 coarse
//         dofs per box in 3-D, 3 in 2-D), zero on penalty dofs
//   A_c   = P^T A P (Galerkin), exact dense inverse on the coarsest grid
// Partitioned meshes: the boxes are cells of one grid over the global bounding box.  Every domain keeps, per
// level, the sub-grid its nodes reach and its own part P_r^T K_r P_r of the operator; a box holding nodes of
// several domains is shared like an interface node (one neighbour exchange sums over the sharers, in ascending
// rank order: all copies stay bitwise equal).  Only the coarsest level (<= 125 boxes) is replicated, through an
// all-reduce of at most 750 numbers per application.  No collective grows with the mesh.
//   cycle V(1,1), symmetric -> valid CG preconditioner; damped point Jacobi on the fine level, damped
//         block Jacobi (one box = one block) on the box levels with omega = 4/(3*1.1*lambda_max estimate)
#pragma once
#include <memory>
#include <vector>

#include "comm.h"
#include "objects.h"

namespace fem {

struct MgLevel {
    int nx = 0, ny = 0, nz = 0, nb = 0;
    double lo[3] = {0, 0, 0}, H = 1.0;   // box b = (ix,iy,iz) has its centre at lo + (idx + 0.5) * H
    DBuf<double> st;    // [27][cs*cs][nb]  stencil entries, box index fastest
    DBuf<float> stf;    // the cycle's single-precision copy of st (large levels; see setup_t)
    DBuf<double> dinv;  // [cs*cs][nb]  inverse of the diagonal block of every box (block-Jacobi smoother)
    DBuf<double> x, r, t;   // [nb*cs] iterate, right-hand side, residual
    // partitioned mesh: nx x ny x nz is the sub-grid of the global box grid that this domain's nodes reach, `off` its origin in
    // the global grid, `sh` what relates it to the next level's sub-grid (parent = (index + sh) >> 1); the boxes that also hold
    // nodes of other domains are shared like interface nodes (one Halo per level), `st` is this domain's part of the operator and
    // `diag` the diagonal blocks summed over the sharers.  Undivided mesh: off = sh = 0, no halo.
    int off[3] = {0, 0, 0}, sh[3] = {0, 0, 0};
    std::unique_ptr<Halo> halo;
    DBuf<double> diag;
};

// What the level-1 aggregation takes from the mesh alone -- bounding box, the box of every node, the node lists of the boxes: kept on
// the mesh's MatPattern (objects.h) and shared by every set-up on that mesh with the same box size and the same partition, as the
// sparsity and the tile plan are (the reference assembles and sets its solver up once per run, src/main.F90:151-171,339-403; a host
// that re-assembles on one mesh pays the geometry once).
struct MgBoxes {
    double box = 0.0;
    const void* halo_seen = nullptr;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, nn_global = 0.0;
    std::vector<double> tab;          // every domain's bounding box and count of owned nodes
    int grid[3] = {0, 0, 0}, sub[3] = {0, 0, 0}, off[3] = {0, 0, 0};   // global level-1 grid, this domain's sub-grid and its origin
    DBuf<int> agg;                  // [nn] level-1 box of every node
    DBuf<int> box_ptr, box_nodes;   // level-1 box -> its nodes (ascending)
    DBuf<int> cnt;                  // [nb] nodes per box
    // level-1 Galerkin product with full-line reads (mg.hip, k_galerkin_fine_quad): the groups of four consecutive boxes that
    // qualify -- first box, slot in gnodes, their 256 nodes sorted by id -- and the boxes that keep the wavefront-per-box kernel
    DBuf<int> gnodes, group_box0, group_slot, rest_boxes;
    DBuf<float> drel_v, drel_p;     // [nn][3] offsets of the nodes from their box centres, by node and in box-list order (the cycle's transfer kernels)
    int ngroups = 0, nrest = 0;
};

struct Multigrid {
    fem_mat A = nullptr;
    int bs = 0, cs = 0, nn = 0;      // fine block size; coarse dofs per box (rigid-body modes: 3 in 2-D, 6 in 3-D)
    double omega = 0.7, alpha = 1.5;   // upper bound of the fine-level Jacobi damping (V-cycle form); scaling of the coarse correction
    double omega_fine = 0.7, lambda_fine = 0.0;   // damping actually used: min(omega, 1.6 / lambda_max estimate)
    double box = 4.0;               // level-1 box edge in mean node spacings (-pc_mg_box)
    bool fine_additive = false;     // -pc_mg_type additive: z = D^-1 r + P V(P^T r), no fine-level product
    bool float_offsets = true;      // the cycle's transfer kernels read MgBoxes::drel_* (FEM_AMD_MG_FLOAT_OFFSETS=0: the coordinates)
    bool stencil_f32 = true;        // -pc_mg_stencil_f32: the cycle streams a single-precision copy of the large levels' stencils
    std::vector<MgLevel> lev;       // lev[0] = level 1 (first coarse grid)
    std::shared_ptr<MgBoxes> bx;    // level-1 aggregates (shared with the mesh's pattern)
    DBuf<unsigned char> w;          // [nn*bs] 1 on free dofs, 0 on penalty dofs
    DBuf<unsigned char> wr;         // partitioned mesh: w on the nodes this domain owns, 0 on the other copies
    DBuf<double> dinv0, t0;         // fine-level Jacobi inverse diagonal, work vector
    DBuf<double> cinv;              // dense inverse of the coarsest operator
    DBuf<double> omega_l;           // [levels] smoother damping per box level (from an eigenvalue estimate) + scratch
    int ncoarse = 0;
    void setup(fem_mat A);
    // z = M^-1 r  (one V-cycle, zero initial guess); `done` (device flag, may be null) lets the
    // two fine-level products return immediately once the enclosing solve has finished
    // With pzz/pzr (npart partials each) the additive form also leaves the partial sums of z.z
    // and z.r there (fuses_dots() tells the caller whether it did).
    void apply(const double* r, double* z, const int* done = nullptr, double* pzz = nullptr, double* pzr = nullptr,
               int npart = 0);
    bool fuses_dots() const;
    size_t device_bytes() const;
};

}  // namespace fem
