// fem_amd/csrc/direct.h -- direct solve behind the KSP surface (`-ksp_type preonly -pc_type lu|cholesky`).
//
// The reference's production path is a direct factorisation: KSPPREONLY + PCLU through MUMPS
// (src/main.F90:354-372).  SURVEY.md 8(f4) asks for that option for small and medium decks.
// Mat_K is symmetric (the penalty method only INSERTs on the diagonal, src/m_global.F90:400-414)
// and positive definite once rigid-body motion is constrained, so the factorisation is a banded
// Cholesky  P K P^T = L L^T  after a reverse Cuthill-McKee renumbering of the nodes:
//   storage   lower band, element (i, j) at ab[i*bw + j + bw]  (j <= i <= j + bw): any block inside
//             the band is an ordinary row-major matrix with row stride bw
//   factor    right-looking, 64-column blocks: potrf of the diagonal block plus the inverse of its
//             factor (one workgroup, LDS) -> panel L21 = A21 L11^-T and the symmetric rank-64 update
//             of the trailing window, both 64x64 tiles on the fp64 matrix cores (v_mfma_f64_16x16x4_f64)
//   solve     blocked forward / backward substitution, one launch per 64-row block; the diagonal
//             blocks act through their stored inverses (no dependent chains)
// Memory is n*(bw+1) doubles; set-up refuses decks whose band does not fit (use -ksp_type cg).
#pragma once
#include <vector>

#include "objects.h"

namespace fem {

struct BandCholesky {
    fem_mat A = nullptr;
    int bs = 0;
    long long n = 0;
    int bw = 0;                 // half bandwidth in dofs after renumbering
    bool failed = false;        // a non-positive pivot was met (matrix not positive definite)
    DBuf<int> perm_dof;         // [n] new position of every dof
    DBuf<double> ab;            // [n*(bw+1)] the factor L, band storage
    DBuf<double> linv;          // [n/64][64][64] inverse of every diagonal block of L
    DBuf<double> v0, v1, v2;    // work vectors (permuted rhs, y, x)
    DBuf<int> flag;
    void setup(fem_mat A);      // renumber, fill the band, factor
    void solve(const double* b, double* x);
    size_t device_bytes() const { return perm_dof.bytes() + ab.bytes() + linv.bytes() + v0.bytes() + v1.bytes() + v2.bytes(); }
};

// The same options on a Mat that is not symmetric (the Newton Jacobian: Mat_K + cohesive tangents, whose reference
// formula scales only one of its two terms, src/m_global.F90:830-831): general band LU with partial pivoting, as
// LAPACK's dgbtf2 -- row k is exchanged with the largest entry of column k among the kl rows below, the fill stays
// inside kl + ku super-diagonals.  One workgroup walks the columns (the dependency chain is the matrix order), its
// threads share the rank-1 update of the (kl x (kl+ku)) window: meant for the reference's cohesive decks and their
// tilings (config 5: 16,000 dofs, half bandwidth 15), where it finishes in milliseconds; large 3-D decks are not its job.
// A pivot below 1e-6 of the largest finite-size entry is treated the way the reference asks MUMPS to treat it
// (ICNTL(24) = 1, CNTL(3) = 1e-6, src/main.F90:368-371): the unknown is fixed to zero.
struct BandLU {
    fem_mat A = nullptr;
    int bs = 0;
    long long n = 0;
    int kl = 0;                 // half bandwidth in dofs after renumbering (kl = ku)
    int wb = 0;                 // stored row width: 3 kl + 1
    bool failed = false;        // a NaN reached a pivot column
    bool wave = false;          // factorised (and solved) by one wavefront with the window of a step in registers (3 kl + 1 <= 64)
    // narrow bands with many columns: block cyclic reduction (direct.hip), used where set-up met no small pivot; every solve checks its
    // backward error and falls back to the band LU (factorised then, `band_ready`)
    bool bcr = false, band_ready = false;
    int nbk = 0;                // blocks of kl rows
    DBuf<double> cL0, cD0, cU0, cL, cD, cU, cr;   // the block tridiagonal form (pristine / work copies), right-hand side and solution by blocks
    bool bcr_resident = false;  // the reduction as one resident launch per solve (k_bcr_resident)
    bool bcr_pcr = false;       // ... in its parallel form (k_pcr_resident: few blocks)
    DBuf<double> pcr_extra;     // the second set of arrays its levels alternate with: [D | r | r]
    DBuf<double> cEL, cEU, cz;  // D^-1 [L U r] of the eliminated blocks (resident launch)
    DBuf<unsigned> bcr_words;   // the grid barrier's words, status words behind them
    void factor_band();
    bool bcr_reduce(bool with_rhs);   // the levels down (and, with a right-hand side, up); false: a small pivot was met
    int null_pivots = 0;        // pivots below 1e-6 of the largest entry, fixed to zero
    const int* perm_dof = nullptr;   // [n] new position of every dof (kept on the sparsity's LuSymbolic)
    DBuf<double> ab;            // [n*wb] row i holds columns i-kl .. i+2kl
    DBuf<int> ipiv;             // [n]
    DBuf<double> v0;            // work vector
    DBuf<int> flag;
    // a matrix made of small independent blocks (connected components of at most 64 dofs): dense LU per block
    bool batched = false;
    int ncomp = 0;
    const int *comp_ptr = nullptr, *comp_nodes = nullptr, *node_loc = nullptr;   // component -> its nodes (ascending); node -> position in its component (kept on the sparsity's LuSymbolic)
    const long long* fac_off = nullptr;         // component -> offset of its m x m factor in fac
    DBuf<double> fac;
    void setup(fem_mat A);
    void solve(const double* b, double* x);
};

// reverse Cuthill-McKee ordering of an undirected graph given as CSR adjacency (host)
std::vector<int> rcm_order(int n, const std::vector<int>& ptr, const std::vector<int>& adj);

}  // namespace fem
