/* include/fem_amd.h -- C ABI of libfem_amd.so (MI355X / gfx950).
 *
 * The drop-in boundary for the assemble-and-solve hot path of the Fortran/PETSc
 * code srinath-chakravarthy/FEM ("defmod").  The reference has no FFI seam: its
 * "API" is (i) the Fortran module procedures of `elems`/`local`
 * (src/m_elems.F90, src/m_local.F90) and (ii) the PETSc Mat/Vec/KSP calls made
 * from src/main.F90 and src/m_global.F90.  Each entry point below names the
 * reference call site (file:line, relative to the reference root) it replaces.
 * fortran/fem_amd_capi.F90 holds the ISO_C_BINDING interface module a
 * maintainer of the reference would `use`; INTEGRATION.md shows the edit.
 *
 * Conventions
 *   - plain C types only; all functions return 0 on success, non-zero on error
 *     (PETSc style `ierr`; fem_last_error() gives the message).
 *   - node / element / dof ids are 0-based int32 (as at the reference's PETSc
 *     boundary, src/m_global.F90:294,314); dof = pdim*node + component.
 *   - element type numbers as elTypes in src/m_elems.F90:26:
 *     1 tri, 2 qua, 3 tet, 4 hex (8 nodes), 5 coh.
 *   - small dense element arrays are column-major exactly as the reference
 *     passes them: ecoords(nodecount,pdim), k(n*pdim,n*pdim), estress(nip,cpdim),
 *     stress_node(nnds,cpdim).
 *   - "host" pointers are caller-owned host memory; handles own device memory
 *     and must be destroyed by the caller (PETSc object discipline,
 *     src/main.F90:300-312).
 *   - all arithmetic is IEEE fp64.  There is no CPU fallback: every call that
 *     computes needs the GPU and fails loudly without it.
 */
#ifndef FEM_AMD_H
#define FEM_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fem_ctx_s *fem_ctx;   /* device + stream                          */
typedef struct fem_comm_s *fem_comm; /* domain communicator (self / local / RCCL / host callbacks) */
typedef struct fem_mesh_s *fem_mesh; /* device-resident SoA mesh of one domain   */
typedef struct fem_mat_s *fem_mat;   /* Mat_K       (src/m_global.F90:26)        */
typedef struct fem_vec_s *fem_vec;   /* Vec_F/Vec_U (src/m_global.F90:24)        */
typedef struct fem_ksp_s *fem_ksp;   /* Krylov      (src/m_global.F90:27)        */
typedef struct fem_snes_s *fem_snes; /* Solver      (src/m_global.F90:28)        */

#define FEM_TRI 1
#define FEM_QUA 2
#define FEM_TET 3
#define FEM_HEX 4
#define FEM_COH 5

/* KSPConvergedReason values (PETSc). */
#define FEM_KSP_CONVERGED_RTOL 2
#define FEM_KSP_CONVERGED_ATOL 3
#define FEM_KSP_CONVERGED_ITERATING 0
#define FEM_KSP_DIVERGED_ITS (-3)
#define FEM_KSP_DIVERGED_DTOL (-4)
#define FEM_KSP_DIVERGED_INDEFINITE_PC (-8)
#define FEM_KSP_DIVERGED_NANORINF (-9)
#define FEM_KSP_DIVERGED_INDEFINITE_MAT (-10)

#define FEM_KSP_CONVERGED_ITS 4
#define FEM_KSP_DIVERGED_BREAKDOWN (-5)
#define FEM_KSP_DIVERGED_PC_FAILED (-11)

/* SNESConvergedReason values (PETSc). */
#define FEM_SNES_CONVERGED_FNORM_ABS 2
#define FEM_SNES_CONVERGED_FNORM_RELATIVE 3
#define FEM_SNES_CONVERGED_SNORM_RELATIVE 4
#define FEM_SNES_DIVERGED_FUNCTION_COUNT (-2)
#define FEM_SNES_DIVERGED_LINEAR_SOLVE (-3)
#define FEM_SNES_DIVERGED_FNORM_NAN (-4)
#define FEM_SNES_DIVERGED_MAX_IT (-5)
#define FEM_SNES_DIVERGED_LINE_SEARCH (-6)

/* PETSC_DEFAULT for fem_ksp_set_tolerances arguments. */
#define FEM_DEFAULT_REAL (-2.0)
#define FEM_DEFAULT_INT (-2)

int fem_version(void);
const char *fem_last_error(void);
/* Number of visible GPUs; 0 means the library cannot compute. */
int fem_device_count(int *count);

/* ---- context: replaces PetscInitialize/PetscFinalize (src/main.F90:26,333) ---- */
int fem_ctx_create(int device, fem_ctx *ctx);
int fem_ctx_destroy(fem_ctx ctx);
int fem_ctx_synchronize(fem_ctx ctx);
/* HIP-event stopwatch on the context's stream (bench.py uses it for the
 * roofline leg: events are recorded on the stream the kernels run on). */
int fem_ctx_timer_start(fem_ctx ctx);
int fem_ctx_timer_stop(fem_ctx ctx, double *elapsed_ms);

/* ---- element library: module `elems` (src/m_elems.F90) -------------------- */
/* getNodeCount :83, getNip :92, getDim :49, getNps :40, getVtkid :107 */
int fem_el_counts(int eltype, int *nodecount, int *nip, int *dim, int *nps, int *vtkid);
/* Initialize :124 -> SamPts*, ShapeFuncPrecomp*, NodalStressInv :725.  Returns
 * the tables the kernels use: ipoints(nip,pdim), weights(nip), shape(nip,n)
 * [N_node at each Gauss point], n2inv(nip,n) [inverse of shape]; has_inv = 0
 * for types the reference skips (nip != nodecount). */
int fem_el_tables(int eltype, double *ipoints, double *weights, double *shape, double *n2inv, int *has_inv);

/* ---- element integration: module `local` (src/m_local.F90), batched ------- */
/* FormElKE :21-67 for nel elements of one type.  ecoords[nel][n*pdim] (each
 * element's block column-major (n,pdim)), E[nel], nu[nel] -> k[nel][nd*nd]
 * (each column-major, nd = n*pdim).  Host pointers. */
int fem_form_el_ke(fem_ctx ctx, int eltype, int nel, const double *ecoords, const double *E, const double *nu, double *k);
/* CalcElStress :93-123.  edisp[nel][nd] -> estress[nel][nip*cpdim] (col-major (nip,cpdim)). */
int fem_calc_el_stress(fem_ctx ctx, int eltype, int nel, const double *ecoords, const double *edisp, const double *E,
                       const double *nu, double *estress);
/* FormElIndx :70-78 (0-based). */
int fem_form_el_indx(int pdim, int n, const int *enodes, int *indx);
/* ShapeFunc src/m_elems.F90:136-141: N(1:nodecount) at integration point gp (0-based) of the type's rule. */
int fem_shape_func(int eltype, int gp, double *N);
/* ShapeFuncd src/m_elems.F90:177-185 at npts points: coord[npts][pdim] -> dN[npts][pdim*n], each dN(pdim,n)
 * column-major (derivatives with respect to the reference coordinates). */
int fem_shape_funcd(fem_ctx ctx, int eltype, int npts, const double *coord, double *dN);
/* FormdNdetJ src/m_local.F90:175-189 for nel elements, each at its own point: ipcoord[nel][pdim],
 * ecoords[nel][n*pdim] (column-major (n,pdim)) -> dN[nel][pdim*n] (dN/dx, column-major (pdim,n)), detj[nel]
 * (signed, MatDet src/m_utils.F90:60-66; J^-1 in closed form where the reference calls LAPACK). */
int fem_form_dn_detj(fem_ctx ctx, int eltype, int nel, const double *ipcoord, const double *ecoords, double *dN, double *detj);
/* BMat src/m_local.F90:148-171: dN[nel][pdim*n] -> B[nel][cpdim*(n*pdim)], each B(cpdim, n*pdim) column-major. */
int fem_bmat(fem_ctx ctx, int pdim, int nodecount, int nel, const double *dN, double *B);
/* DMat src/m_local.F90:204-228: D[nmat][cpdim*cpdim] (plane strain in 2-D). */
int fem_dmat(fem_ctx ctx, int pdim, int nmat, const double *E, const double *nu, double *D);

/* ---- cohesive element, element level ----------------------------------------------
 * Batched getCohValues / getCohRels / getCohGaps (src/m_elems.F90:658-720), Seplaw_1_Tract / Seplaw_1_Stiff
 * (src/m_seplaw.F90:15-94) and the element vector / matrix of applyTract_1 (src/m_global.F90:622-682) and
 * applyStiff_1 (:762-845) for ncoh 4-node cohesive elements.  ecoords[ncoh][8] = ecoords(4,2) column-major;
 * u8[ncoh][8] = the eight numbers getCohRels reads (local_u(2(k-1)+i), k = 1..4: the reference hands it
 * the rank's first four nodes, a host that wants the element's own nodes passes those); props[ncoh][6];
 * tract[ncoh][8], stiff[ncoh][64] = result(8,8) column-major, gp[ncoh][2][14] = urel, vrel, gap, vgap,
 * tract, stiff(2,2 column-major) per integration point; any output may be NULL.  Reference behaviour kept
 * as written: applyTract_1 assigns per integration point (the last one wins), applyStiff_1 scales only its
 * second term.  Host pointers. */
int fem_coh_apply_1(fem_ctx ctx, int ncoh, const double *ecoords, const double *u8, const double *props, double dt,
                    double *tract, double *stiff, double *gp);

/* ---- mesh: local_elements/coords/mat (src/m_global.F90:40,22,21) ---------- */
/* Uploads one domain's mesh as SoA.  eltype[nels], eptr[nels+1], conn[],
 * mat_id[nels] (0-based row of mats), coords[nnds*pdim] node-major,
 * mats[nmats*5] rows (E nu visc expn density; only E, nu are used,
 * src/m_global.F90:254-255).  Elements of type coh are ignored here. */
int fem_mesh_create(fem_ctx ctx, int pdim, int nels, int nnds, const int *eltype, const int *eptr, const int *conn,
                    const int *mat_id, const double *coords, int nmats, const double *mats, fem_mesh *mesh);
int fem_mesh_destroy(fem_mesh mesh);
/* Multi-domain runs: attach the communicator and the interface lists.
 * neigh_rank[nneigh] ascending; neigh_ptr[nneigh+1]; neigh_nodes[] = local node
 * ids shared with each neighbour, in the SAME order on both sides (e.g. sorted
 * by global node id).  Replaces the MatAssembly stash / VecScatter / AllReduce
 * traffic of src/main.F90:167-171,185-190,274,286-289 by neighbour halo sums. */
int fem_mesh_set_halo(fem_mesh mesh, fem_comm comm, int nneigh, const int *neigh_rank, const int *neigh_ptr,
                      const int *neigh_nodes);

/* ---- Mat: Mat_K ------------------------------------------------------------ */
/* MatCreateAIJ src/main.F90:154 (+ the implicit pattern growth of
 * MatSetValues): builds the sparsity from the connectivity on the device. */
int fem_mat_create_aij(fem_mesh mesh, fem_mat *mat);
int fem_mat_destroy(fem_mat mat);
int fem_mat_zero_entries(fem_mat mat);
/* The assembly loop src/main.F90:157-168: FormLocalK (src/m_global.F90:239-261)
 * -> FormElKE -> MatSetValues(ADD_VALUES) -> MatAssemblyBegin/End, for every
 * element of the mesh, in one pass of HIP kernels. */
int fem_mat_assemble_elements(fem_mat mat);
/* ApplyKBC src/main.F90:169, src/m_global.F90:264-299: INSERT the penalty on the
 * diagonal of every constrained dof.  bcnode[nbcs*(1+pdim)] = node, flag_1..
 * flag_pdim (0 = constrained). */
int fem_mat_apply_kbc(fem_mat mat, int nbcs, const int *bcnode);
/* Sizes: rows (dofs), scalar nonzeros of the AIJ pattern, bytes held on device. */
int fem_mat_get_info(fem_mat mat, long long *nrows, long long *nnz, long long *device_bytes);
/* AIJ export to host (MatView / MatGetRow use): rowptr[nrows+1], colind[nnz],
 * val[nnz]; columns ascending.  Any pointer may be NULL. */
int fem_mat_get_csr(fem_mat mat, int *rowptr, int *colind, double *val);
/* How MatMult is carried out.  "assembled" (default): sliced block-ELL SpMV over the stored blocks.
 * "ebe": element-by-element recomputation from the mesh (PETSc's MATSHELL idea: same operator, no
 * matrix read; needs the usual assemble + ApplyKBC first, which also serve the preconditioners);
 * deterministic, about four times as fast as the SpMV on hex8 meshes.  On a mesh of hex8 elements alone (or of quad4
 * elements alone) the product is one pass over tiles of elements with the nodal sums in LDS; every other mesh -- and "ebe_gather" on any mesh -- writes
 * element forces to a buffer and sums them node by node.  Also `-mat_mult_type` of the KSP. */
int fem_mat_set_mult_type(fem_mat mat, const char *type);
/* HIP-event profile of the element-by-element product on the product's own stream.  on = 0: off;
 * on = p > 0: every p-th product is bracketed by events (p = 1: all of them).  get_profile sums over
 * the n sampled products among the first `max_products` products since profiling was switched on
 * (element kernels / node gather / penalty rows, in ms). */
int fem_mat_set_profiling(fem_mat mat, int on);
int fem_mat_get_profile(fem_mat mat, int max_products, int *n, double *element_ms, double *gather_ms, double *penalty_ms);
/* Which product MatMult runs and what its plan looks like: kind 0 assembled SpMV, 1 element buffer + node gather, 2 one-pass
 * tiles (then: number of tiles, those of them whose elements are all parallelepipeds, element visits per product -- the
 * elements of a tile's rim are computed by every tile that owns one of their nodes --, lane slots, entries of the tiles'
 * node lists, host set-up time). */
int fem_mat_get_mult_info(fem_mat mat, int *kind, long long *tiles, long long *affine_tiles, long long *element_visits,
                          long long *lane_slots, long long *tile_nodes, double *setup_ms);
/* One-pass tiles, further figures of the plan: the tiles (of parallelepipeds) whose elements all carry one stored geometry record,
 * whether the product reads that record once per tile (per_tile_records 1: it does when every such tile of the plan has one; 2: that,
 * and every list of a tile sits at a fixed stride per list position with the slot records of all its -- at most four -- steps, so
 * that the kernel asks for all of them with the tile's record: k_ebe_tile_x4) and whether the plan is
 * large enough for its lists to be loaded with the streaming hint (lists + five vectors beyond the 256 MB Infinity Cache).  Zeros for
 * the other products.  (No reference counterpart: the reference's MatMult is PETSc's, src/main.F90:247.) */
int fem_mat_get_tile_geometry_info(fem_mat mat, long long *uniform_tiles, int *per_tile_records, int *streaming_lists);
/* Cross-check of the tile plan behind `-mat_mult_type ebe` (hex8 / quad4 meshes): builds the plan afresh with the named
 * builder -- "device" (the one the product uses: integer kernels, nothing leaves HBM) or "host" (worker threads over a host
 * copy of the mesh, kept as the check) -- without caching it, and returns a digest of every array: [0] tiles, [1] lane
 * slots, [2] node-list entries, [3] element visits, [4..11] FNV-1a of hdr, steps, nodes, elems, conn16, flags, the three
 * tile lists, the stored geometry.  A Mat must have been created from the mesh (its symbolic phase supplies the node ->
 * element lists).  digest[0] = 0 when the mesh gets no tile plan (other element types, more than 16 colours). */
int fem_mesh_tile_plan_digest(fem_mesh mesh, const char *builder, unsigned long long digest[12], double *setup_ms);
/* MatSetValues (src/m_global.F90:257-260 ADD_VALUES, :296 INSERT_VALUES): a logically dense m x n block, row-major, by
 * 0-based scalar dof indices; negative indices are skipped (as PETSc does); mode 1 = INSERT_VALUES, 2 = ADD_VALUES.  An
 * entry outside the sparsity of fem_mat_create_aij is an error.  Acts on the assembled values (a host patching a few rows);
 * the element loop is fem_mat_assemble_elements. */
int fem_mat_set_values(fem_mat mat, int m, const int *rows, int n, const int *cols, const double *v, int mode);
/* MatMult: y = K x (sub-assembled product + halo sum when a halo is attached). */
int fem_mat_mult(fem_mat mat, fem_vec x, fem_vec y);
int fem_mat_get_diagonal(fem_mat mat, fem_vec d);
/* Runs `reps` back-to-back MatMult launches between two HIP events recorded on
 * the context's stream; returns the average launch time. */
int fem_mat_mult_timed(fem_mat mat, fem_vec x, fem_vec y, int reps, double *avg_ms);

/* ---- Vec: Vec_F / Vec_U ------------------------------------------------------ */
int fem_vec_create(fem_ctx ctx, long long n, fem_vec *v); /* VecCreateMPI src/main.F90:138 */
int fem_vec_destroy(fem_vec v);
int fem_vec_get_size(fem_vec v, long long *n);
int fem_vec_set(fem_vec v, double alpha);                /* VecSet / VecZeroEntries */
int fem_vec_set_array(fem_vec v, const double *host);    /* VecSetValues(INSERT) of all entries */
int fem_vec_get_array(fem_vec v, double *host);          /* GetVec_U src/m_global.F90:549-564 */
int fem_vec_device_ptr(fem_vec v, void **dptr);
int fem_vec_copy(fem_vec src, fem_vec dst);
int fem_vec_axpy(fem_vec y, double a, fem_vec x);        /* y += a x */
int fem_vec_dot(fem_vec x, fem_vec y, double *dot);
int fem_vec_norm2(fem_vec x, double *nrm);

/* ---- RHS: FormRHS src/m_global.F90:373-436 + EnforceBCForce :439-455 -------- */
/* fval[nfrcs*(pdim+2)] rows f_1..f_pdim t1 t2 (t1,t2 already clamped to t as
 * ReadDistForces does, src/m_io.F90:357-358); telsd[ntrcs*2] = element
 * (0-based), side (1-based); tval like fval.  t_total is the deck's `t`. */
int fem_form_rhs(fem_mesh mesh, int nbcs, const int *bcnode, const double *bcval, int nfrcs, const int *fnode,
                 const double *fval, int ntrcs, const int *telsd, const double *tval, double t_total, double t_init,
                 double dt, fem_vec F);

/* ---- KSP: SetupKSPSolver src/main.F90:339-403, KSPSolve :247 ----------------- */
int fem_ksp_create(fem_ctx ctx, fem_ksp *ksp);
int fem_ksp_destroy(fem_ksp ksp);
int fem_ksp_set_operators(fem_ksp ksp, fem_mat A);          /* KSPSetOperators :348 */
/* KSPSetType / PCSetType.  "cg" pairs with "jacobi" | "none" | "mg" ("gamg"); "bcgs" (BiCGStab, for the unsymmetric
 * Newton Jacobian) with "jacobi" | "none"; "preonly" pairs with
 * "lu" | "cholesky" -- the reference's own production pairing KSPPREONLY + PCLU (MUMPS),
 * src/main.F90:354-358 -- served here by a banded Cholesky factorisation on the device (single
 * domain, decks whose band fits the HBM; its = 1, reason KSP_CONVERGED_ITS = 4, or
 * KSP_DIVERGED_PC_FAILED = -11 when a pivot is not positive); on a Mat that carries cohesive tangents (not symmetric)
 * by a band LU with partial pivoting and MUMPS-style null-pivot fixing (src/main.F90:368-371). */
int fem_ksp_set_type(fem_ksp ksp, const char *type);
int fem_ksp_set_pc_type(fem_ksp ksp, const char *type);
/* KSPSetTolerances :350; FEM_DEFAULT_* keeps the PETSc default
 * (rtol 1e-5, abstol 1e-50, dtol 1e5, maxits 10000). */
int fem_ksp_set_tolerances(fem_ksp ksp, double rtol, double abstol, double dtol, int maxits);
/* KSPSetFromOptions :377: parses "-ksp_type cg|bcgs|preonly -pc_type jacobi|none|mg|gamg|lu|cholesky
 * -ksp_rtol 1e-12 -ksp_atol .. -ksp_divtol .. -ksp_max_it .. -pc_mg_type multiplicative|additive
 * -pc_mg_alpha .. -pc_mg_omega .. -pc_mg_box .. -pc_mg_slice_min .. -mat_mult_type assembled|ebe|ebe_gather" */
int fem_ksp_set_from_options(fem_ksp ksp, const char *options);
int fem_ksp_set_up(fem_ksp ksp);                            /* KSPSetUp :378 */
int fem_ksp_solve(fem_ksp ksp, fem_vec b, fem_vec x);       /* KSPSolve :247 (zero initial guess) */
int fem_ksp_get_iteration_number(fem_ksp ksp, int *its);
int fem_ksp_get_residual_norm(fem_ksp ksp, double *rnorm);
int fem_ksp_get_converged_reason(fem_ksp ksp, int *reason);
/* KSPSetResidualHistory / KSPGetResidualHistory: copies min(len, its+1) norms. */
int fem_ksp_get_residual_history(fem_ksp ksp, double *hist, int len, int *n);
/* Per-solve HIP-event profile on the solve's stream.  on = 0: off; on = p > 0: the MatMult of every
 * p-th iteration is bracketed by events.  get_profile: number of sampled MatMult launches that ran,
 * their summed duration, total solve duration. */
int fem_ksp_set_profiling(fem_ksp ksp, int on);
int fem_ksp_get_profile(fem_ksp ksp, int *n_spmv, double *spmv_ms, double *solve_ms);
/* -ksp_reuse_solution 0|1 (default 1; iterative solves): a solve whose operator state, options and right-hand side are those of
 * the previous solve, bit for bit, returns the kept solution instead of iterating again -- increments 2..n of the reference's
 * loop on a linear deck with ramped loads (src/main.F90:216-250: one KSPSolve per increment, the same Vec_F every time).
 * n = how many solves of this KSP were answered that way. */
int fem_ksp_get_reused_solves(fem_ksp ksp, int *n);
/* -ksp_cg_resident 0|1 (default 1): KSPCG with PCJACOBI / PCNONE as ONE resident launch -- a workgroup per tile of the element-by-
 * element product stays for the whole solve, two grid-wide barriers per iteration -- where the operator's tile plan has every tile
 * resident at once (quad4 meshes of about a million dofs: BASELINE config 2) and the mesh is undivided; every other solve, and a
 * launch whose barrier gives up, takes the kernels of the launch-per-iteration path.  Same KSPSolve semantics (src/main.F90:247);
 * the iteration path agrees with the other one to round-off.  n = how many solves of this KSP ran that way. */
int fem_ksp_get_resident_solves(fem_ksp ksp, int *n);

/* ---- cohesive elements and the Newton increment: SURVEY 8 f1, BASELINE config 5 ------------------------------
 * The reference solves a deck with cohesive elements through SNES (src/main.F90:200-206,240-245) with the two
 * callbacks CalcJacobian (src/m_global.F90:98-158) and CalcResidual (:161-235).  The callbacks stay host code here
 * too (fem_amd/driver.py, fortran/): they are written with the device primitives below, one per PETSc call or loop
 * of the reference's callbacks.
 *
 * fem_mesh_set_cohesive: nlmat[nels] = el%nlMat of every element of the list the mesh was created from (0 = linear,
 * m >= 1 = cohesive with material cohMats(m), src/m_io.F90:94,308-314); props[ncohmats*6] rows sigma_max delta_n
 * delta_t q r zeta (src/m_seplaw.F90:19-24).  Call it before the first fem_mat_create_aij: the sparsity then holds
 * the node pairs of the cohesive elements.  Cohesive elements add nothing to Mat_K (the reference sends them through
 * FormElKE with an unhandled type, i.e. integrates an uninitialised dN: undefined behaviour that is not reproduced)
 * and are left out of the stress recovery. */
int fem_mesh_set_cohesive(fem_mesh mesh, const int *nlmat, int ncohmats, const double *props);
/* MatDuplicate(Mat_K, MAT_COPY_VALUES, Jacobian) src/main.F90:197; MatCopy(Mat_K, Jacobian, SAME_NONZERO_PATTERN)
 * src/m_global.F90:123 (both matrices from the same mesh). */
int fem_mat_duplicate(fem_mat A, fem_mat *B);
int fem_mat_copy(fem_mat src, fem_mat dst);
/* The element loop of CalcJacobian (src/m_global.F90:127-147): applyStiff_1 for every cohesive element at the total
 * displacement u_total and 64 MatSetValue(ADD_VALUES) each.  getCohRels reads four fixed local nodes of the rank
 * instead of the element's (src/m_elems.F90:697-704): rels_mode 0 repeats that with rels_nodes[4] (NULL = the first
 * four nodes, as on one rank), rels_mode 1 uses the element's own nodes.  ApplyKBC must follow, as in the reference. */
int fem_mat_add_cohesive(fem_mat J, fem_vec u_total, double dt, int rels_mode, const int *rels_nodes);
/* The element loop of CalcResidual (:186-205): R(node) += applyTract_1 for every cohesive element. */
int fem_vec_add_cohesive_traction(fem_mesh mesh, fem_vec u_total, double dt, int rels_mode, const int *rels_nodes, fem_vec R);
/* EnforceBCForce (src/m_global.F90:439-455) on any vector: INSERT PENALTY*val*dt/t_total on the constrained dofs. */
int fem_enforce_bc_force(fem_mesh mesh, int nbcs, const int *bcnode, const double *bcval, double dt, double t_total, fem_vec R);
/* MatMultAdd (src/m_global.F90:224): z = y + A x (z may be y). */
int fem_mat_mult_add(fem_mat A, fem_vec x, fem_vec y, fem_vec z);
int fem_vec_scale(fem_vec v, double alpha);              /* VecScale :223 */

/* SNES: Newton with PETSc's default `bt` line search (SNESNEWTONLS), PETSc's defaults and convergence tests. */
typedef int (*fem_snes_function_cb)(fem_snes snes, fem_vec x, fem_vec f, void *ctx);            /* CalcResidual */
typedef int (*fem_snes_jacobian_cb)(fem_snes snes, fem_vec x, fem_mat J, fem_mat P, void *ctx); /* CalcJacobian */
int fem_snes_create(fem_ctx ctx, fem_snes *snes);                                   /* SNESCreate :200 */
int fem_snes_destroy(fem_snes snes);
int fem_snes_get_ksp(fem_snes snes, fem_ksp *ksp);                                  /* SNESGetKSP :201 (owned by snes) */
int fem_snes_set_function(fem_snes snes, fem_vec r, fem_snes_function_cb f, void *ctx);             /* :204 */
int fem_snes_set_jacobian(fem_snes snes, fem_mat J, fem_mat P, fem_snes_jacobian_cb j, void *ctx);  /* :205 */
/* The reference's own two callbacks (CalcJacobian src/m_global.F90:98-158, CalcResidual :161-235) installed as the library's:
 * the sequence of primitives above, issued without a round trip through the host's language.  K = Mat_K with ApplyKBC, J = its
 * duplicate (Jacobian), R = Residual, F = Vec_F of the increment (the host refreshes it with fem_form_rhs), aggregate_u = the
 * running total the host keeps, work = a scratch Vec; bcnode / bcval as for fem_enforce_bc_force; rels_mode / rels_nodes as for
 * fem_mat_add_cohesive.  *state is freed with fem_snes_free_defmod_callbacks after fem_snes_destroy;
 * fem_snes_calc_defmod_jacobian(state, Vec_U) is the CalcJacobian call of src/main.F90:241 in front of SNESSolve. */
int fem_snes_set_defmod_callbacks(fem_snes snes, fem_mesh mesh, fem_mat K, fem_mat J, fem_vec R, fem_vec F, fem_vec aggregate_u,
                                  fem_vec work, int nbcs, const int *bcnode, const double *bcval, double dt, double t_total,
                                  int rels_mode, const int *rels_nodes, void **state);
int fem_snes_calc_defmod_jacobian(void *state, fem_vec du);
int fem_snes_free_defmod_callbacks(void *state);
/* SNESSetFromOptions :206: -snes_rtol -snes_atol -snes_stol -snes_max_it -snes_max_funcs -snes_linesearch_type bt|basic
 * -snes_linesearch_alpha|maxstep|minlambda|max_it; any other option goes to the KSP (KSPSetFromOptions). */
int fem_snes_set_from_options(fem_snes snes, const char *options);
int fem_snes_solve(fem_snes snes, fem_vec x);                                       /* SNESSolve :242 (x = initial guess) */
int fem_snes_get_iteration_number(fem_snes snes, int *its);                         /* :243 */
int fem_snes_get_converged_reason(fem_snes snes, int *reason);
int fem_snes_get_function_norm(fem_snes snes, double *fnorm);
int fem_snes_get_linear_solve_iterations(fem_snes snes, int *lits);
int fem_snes_get_number_function_evals(fem_snes snes, int *nfuncs);

/* Reverse Cuthill-McKee numbering used by the direct solve (host only): CSR adjacency of an
 * undirected graph with n nodes -> perm[old] = new. */
int fem_rcm_order(int n, const int *ptr, const int *adj, int *perm);

/* ---- nodal stress: src/main.F90:253-292 -------------------------------------- */
/* RecoverStress (src/m_global.F90:466) + RecoverNodalStress (:488) + valence
 * averaging (+ halo sums instead of MPI_AllReduce/Mat_Stress) for the
 * increment displacement U.  stress_node(nnds,cpdim) column-major, host. */
int fem_recover_nodal_stress(fem_mesh mesh, fem_vec U, double *stress_node);
/* Same, result left on the device in S (length nnds*cpdim, same layout). */
int fem_recover_nodal_stress_vec(fem_mesh mesh, fem_vec U, fem_vec S);

/* ---- communicators ------------------------------------------------------------ */
int fem_comm_create_self(fem_ctx ctx, fem_comm *comm);
/* RCCL over xGMI: one process per GPU; `unique_id` (128 bytes) comes from
 * fem_comm_get_unique_id on rank 0 and is broadcast by the launcher. */
int fem_comm_get_unique_id(char unique_id[128]);
int fem_comm_create_rccl(fem_ctx ctx, int nranks, int rank, const char unique_id[128], fem_comm *comm);
/* In-process group of `n` domains driven by `n` host threads (one fem_ctx
 * each, any mix of GPUs): used to exercise the halo path on a single GPU. */
int fem_comm_create_local_group(int n, const fem_ctx *ctxs, fem_comm *comms);
/* Host-staged communicator: the library hands its two operations to the host program's own message passing (the reference
 * is an MPI program: Makefile:6-10, MPI_AllReduce src/main.F90:274, MPI_Send/Recv src/m_io.F90:150-237) through pinned host
 * buffers.  `allreduce(user, buf, n)` sums buf[0..n) over all ranks in place (MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE,
 * MPI_SUM)); `exchange(user, nneigh, neigh_rank, off, send, recv)` sends send[off[i]..off[i+1]) to rank neigh_rank[i] and fills
 * the same range of recv from it, for every i (MPI_Sendrecv per neighbour; offsets in doubles).  Both run on the calling thread;
 * every rank calls the library in the same order, as with PETSc.  Slower than RCCL (two PCIe hops per operation): meant for
 * hosts without RCCL peers and for rehearsing the multi-rank paths with several processes on one GPU (bench.py --rehearse). */
typedef void (*fem_comm_allreduce_fn)(void *user, double *buf, int n);
typedef void (*fem_comm_exchange_fn)(void *user, int nneigh, const int *neigh_rank, const long long *off, const double *send,
                                     double *recv);
int fem_comm_create_callbacks(fem_ctx ctx, int nranks, int rank, fem_comm_allreduce_fn allreduce, fem_comm_exchange_fn exchange,
                              void *user, fem_comm *comm);
/* What went through the communicator since its creation: stats[0..2] = all-reduce calls, largest count, total count (doubles);
 * stats[3..5] = neighbour exchanges, largest and total number of doubles this domain sent.  The partitioned paths are tested on
 * it: no collective may grow with the mesh (the reference's whole-mesh MPI_AllReduce, src/main.F90:274, is what they replace). */
int fem_comm_get_stats(fem_comm comm, long long stats[6]);
int fem_comm_destroy(fem_comm comm);
int fem_comm_rank(fem_comm comm, int *rank, int *size);
/* The neighbour exchange the halo sums are built on (replaces the MPI_Send/MPI_Recv pairs of
 * src/m_io.F90:150-237 and PETSc's VecScatter): neighbour i receives send[off[i]..off[i+1]) and the
 * same range of recv is filled from it; one grouped ncclSend/ncclRecv per call on the RCCL back end.
 * Blocking (returns when recv is complete). */
int fem_comm_exchange(fem_comm comm, int nneigh, const int *neigh_rank, const long long *off, fem_vec send, fem_vec recv);
/* The same together with the in-place sum of red[0 .. nred) over all domains (the CG's p.Kp all-reduce inside the RCCL group of the
 * halo exchange that follows it, `-ksp_fuse_allreduce 1`: one RCCL launch per iteration less). */
int fem_comm_exchange_allreduce(fem_comm comm, int nneigh, const int *neigh_rank, const long long *off, fem_vec send, fem_vec recv,
                                fem_vec red, int nred);

#ifdef __cplusplus
}
#endif
#endif /* FEM_AMD_H */
