! fortran/fem_amd_capi.F90 -- ISO_C_BINDING interface to libfem_amd.so
! (include/fem_amd.h).  This is the module a maintainer of the reference adds
! to src/ and `use`s from main.F90 / m_global.F90 in place of the PETSc
! Mat/Vec/KSP calls; INTEGRATION.md lists the call-site edits.
!
! Handles are type(c_ptr) (the reference's PETSc handles are opaque 8-byte
! integers, src/m_global.F90:24-29); every function returns the PETSc-style
! integer error code that the reference keeps in `ierr` (src/m_global.F90:42).
! Indices are 0-based at this boundary, as at the reference's PETSc boundary.
module fem_amd_capi
  use iso_c_binding
  implicit none

  integer(c_int), parameter :: FEM_TRI = 1, FEM_QUA = 2, FEM_TET = 3, FEM_HEX = 4, FEM_COH = 5
  real(c_double), parameter :: FEM_DEFAULT_REAL = -2.0d0
  integer(c_int), parameter :: FEM_DEFAULT_INT = -2

  interface
    function fem_last_error() bind(C, name="fem_last_error") result(msg)
      import :: c_ptr
      type(c_ptr) :: msg
    end function
    function fem_ctx_create(device, ctx) bind(C, name="fem_ctx_create") result(ierr)
      import :: c_int, c_ptr
      integer(c_int), value :: device
      type(c_ptr), intent(out) :: ctx
      integer(c_int) :: ierr
    end function
    function fem_ctx_destroy(ctx) bind(C, name="fem_ctx_destroy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int) :: ierr
    end function
    function fem_ctx_synchronize(ctx) bind(C, name="fem_ctx_synchronize") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int) :: ierr
    end function
    function fem_el_counts(eltype, nodecount, nip, dim, nps, vtkid) bind(C, name="fem_el_counts") result(ierr)
      import :: c_int
      integer(c_int), value :: eltype
      integer(c_int), intent(out) :: nodecount, nip, dim, nps, vtkid
      integer(c_int) :: ierr
    end function
    function fem_el_tables(eltype, ipoints, weights, shape, n2inv, has_inv) bind(C, name="fem_el_tables") result(ierr)
      import :: c_int, c_double
      integer(c_int), value :: eltype
      real(c_double), intent(out) :: ipoints(*), weights(*), shape(*), n2inv(*)
      integer(c_int), intent(out) :: has_inv
      integer(c_int) :: ierr
    end function
    function fem_form_el_ke(ctx, eltype, nel, ecoords, E, nu, k) bind(C, name="fem_form_el_ke") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: eltype, nel
      real(c_double), intent(in) :: ecoords(*), E(*), nu(*)
      real(c_double), intent(out) :: k(*)
      integer(c_int) :: ierr
    end function
    function fem_calc_el_stress(ctx, eltype, nel, ecoords, edisp, E, nu, estress) &
        bind(C, name="fem_calc_el_stress") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: eltype, nel
      real(c_double), intent(in) :: ecoords(*), edisp(*), E(*), nu(*)
      real(c_double), intent(out) :: estress(*)
      integer(c_int) :: ierr
    end function
    function fem_form_el_indx(pdim, n, enodes, indx) bind(C, name="fem_form_el_indx") result(ierr)
      import :: c_int
      integer(c_int), value :: pdim, n
      integer(c_int), intent(in) :: enodes(*)
      integer(c_int), intent(out) :: indx(*)
      integer(c_int) :: ierr
    end function
    function fem_mesh_create(ctx, pdim, nels, nnds, eltype, eptr, conn, mat_id, coords, nmats, mats, mesh) &
        bind(C, name="fem_mesh_create") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: pdim, nels, nnds, nmats
      integer(c_int), intent(in) :: eltype(*), eptr(*), conn(*), mat_id(*)
      real(c_double), intent(in) :: coords(*), mats(*)
      type(c_ptr), intent(out) :: mesh
      integer(c_int) :: ierr
    end function
    function fem_mesh_destroy(mesh) bind(C, name="fem_mesh_destroy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mesh
      integer(c_int) :: ierr
    end function
    function fem_mat_create_aij(mesh, mat) bind(C, name="fem_mat_create_aij") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mesh
      type(c_ptr), intent(out) :: mat
      integer(c_int) :: ierr
    end function
    function fem_mat_destroy(mat) bind(C, name="fem_mat_destroy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mat
      integer(c_int) :: ierr
    end function
    function fem_mat_zero_entries(mat) bind(C, name="fem_mat_zero_entries") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mat
      integer(c_int) :: ierr
    end function
    function fem_mat_assemble_elements(mat) bind(C, name="fem_mat_assemble_elements") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mat
      integer(c_int) :: ierr
    end function
    function fem_mat_apply_kbc(mat, nbcs, bcnode) bind(C, name="fem_mat_apply_kbc") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mat
      integer(c_int), value :: nbcs
      integer(c_int), intent(in) :: bcnode(*)
      integer(c_int) :: ierr
    end function
    function fem_mat_get_info(mat, nrows, nnz, device_bytes) bind(C, name="fem_mat_get_info") result(ierr)
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: mat
      integer(c_long_long), intent(out) :: nrows, nnz, device_bytes
      integer(c_int) :: ierr
    end function
    function fem_mat_mult(mat, x, y) bind(C, name="fem_mat_mult") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mat, x, y
      integer(c_int) :: ierr
    end function
    function fem_vec_create(ctx, n, v) bind(C, name="fem_vec_create") result(ierr)
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: ctx
      integer(c_long_long), value :: n
      type(c_ptr), intent(out) :: v
      integer(c_int) :: ierr
    end function
    function fem_vec_destroy(v) bind(C, name="fem_vec_destroy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: v
      integer(c_int) :: ierr
    end function
    function fem_vec_set(v, alpha) bind(C, name="fem_vec_set") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: v
      real(c_double), value :: alpha
      integer(c_int) :: ierr
    end function
    function fem_vec_set_array(v, host) bind(C, name="fem_vec_set_array") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: v
      real(c_double), intent(in) :: host(*)
      integer(c_int) :: ierr
    end function
    function fem_vec_get_array(v, host) bind(C, name="fem_vec_get_array") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: v
      real(c_double), intent(out) :: host(*)
      integer(c_int) :: ierr
    end function
    function fem_form_rhs(mesh, nbcs, bcnode, bcval, nfrcs, fnode, fval, ntrcs, telsd, tval, t_total, t_init, dt, F) &
        bind(C, name="fem_form_rhs") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mesh, F
      integer(c_int), value :: nbcs, nfrcs, ntrcs
      integer(c_int), intent(in) :: bcnode(*), fnode(*), telsd(*)
      real(c_double), intent(in) :: bcval(*), fval(*), tval(*)
      real(c_double), value :: t_total, t_init, dt
      integer(c_int) :: ierr
    end function
    function fem_ksp_create(ctx, ksp) bind(C, name="fem_ksp_create") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      type(c_ptr), intent(out) :: ksp
      integer(c_int) :: ierr
    end function
    function fem_ksp_destroy(ksp) bind(C, name="fem_ksp_destroy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp
      integer(c_int) :: ierr
    end function
    function fem_ksp_set_operators(ksp, A) bind(C, name="fem_ksp_set_operators") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp, A
      integer(c_int) :: ierr
    end function
    function fem_ksp_set_type(ksp, t) bind(C, name="fem_ksp_set_type") result(ierr)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: ksp
      character(kind=c_char), intent(in) :: t(*)
      integer(c_int) :: ierr
    end function
    function fem_ksp_set_pc_type(ksp, t) bind(C, name="fem_ksp_set_pc_type") result(ierr)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: ksp
      character(kind=c_char), intent(in) :: t(*)
      integer(c_int) :: ierr
    end function
    function fem_ksp_set_tolerances(ksp, rtol, abstol, dtol, maxits) bind(C, name="fem_ksp_set_tolerances") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ksp
      real(c_double), value :: rtol, abstol, dtol
      integer(c_int), value :: maxits
      integer(c_int) :: ierr
    end function
    function fem_ksp_set_from_options(ksp, options) bind(C, name="fem_ksp_set_from_options") result(ierr)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: ksp
      character(kind=c_char), intent(in) :: options(*)
      integer(c_int) :: ierr
    end function
    function fem_ksp_set_up(ksp) bind(C, name="fem_ksp_set_up") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp
      integer(c_int) :: ierr
    end function
    function fem_ksp_solve(ksp, b, x) bind(C, name="fem_ksp_solve") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp, b, x
      integer(c_int) :: ierr
    end function
    function fem_ksp_get_iteration_number(ksp, its) bind(C, name="fem_ksp_get_iteration_number") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp
      integer(c_int), intent(out) :: its
      integer(c_int) :: ierr
    end function
    function fem_ksp_get_residual_norm(ksp, rnorm) bind(C, name="fem_ksp_get_residual_norm") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ksp
      real(c_double), intent(out) :: rnorm
      integer(c_int) :: ierr
    end function
    function fem_ksp_get_converged_reason(ksp, reason) bind(C, name="fem_ksp_get_converged_reason") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp
      integer(c_int), intent(out) :: reason
      integer(c_int) :: ierr
    end function
    function fem_recover_nodal_stress(mesh, U, stress_node) bind(C, name="fem_recover_nodal_stress") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mesh, U
      real(c_double), intent(out) :: stress_node(*)
      integer(c_int) :: ierr
    end function
    ! ---- the rest of include/fem_amd.h: multi-GPU, MatView-style access, Vec algebra, profiling ----
    function fem_version() bind(C, name="fem_version") result(v)
      import :: c_int
      integer(c_int) :: v
    end function
    function fem_device_count(count) bind(C, name="fem_device_count") result(ierr)
      import :: c_int
      integer(c_int), intent(out) :: count
      integer(c_int) :: ierr
    end function
    function fem_ctx_timer_start(ctx) bind(C, name="fem_ctx_timer_start") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      integer(c_int) :: ierr
    end function
    function fem_ctx_timer_stop(ctx, elapsed_ms) bind(C, name="fem_ctx_timer_stop") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      real(c_double), intent(out) :: elapsed_ms
      integer(c_int) :: ierr
    end function
    ! one process per GPU (replaces MPI_Comm_rank/size + PETSc's communicators, src/main.F90:27-28)
    function fem_comm_create_self(ctx, comm) bind(C, name="fem_comm_create_self") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      type(c_ptr), intent(out) :: comm
      integer(c_int) :: ierr
    end function
    function fem_comm_get_unique_id(unique_id) bind(C, name="fem_comm_get_unique_id") result(ierr)
      import :: c_int, c_char
      character(kind=c_char), intent(out) :: unique_id(128)
      integer(c_int) :: ierr
    end function
    function fem_comm_create_rccl(ctx, nranks, rank, unique_id, comm) bind(C, name="fem_comm_create_rccl") result(ierr)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: ctx
      integer(c_int), value :: nranks, rank
      character(kind=c_char), intent(in) :: unique_id(128)
      type(c_ptr), intent(out) :: comm
      integer(c_int) :: ierr
    end function
    function fem_comm_create_local_group(n, ctxs, comms) bind(C, name="fem_comm_create_local_group") result(ierr)
      import :: c_int, c_ptr
      integer(c_int), value :: n
      type(c_ptr), intent(in) :: ctxs(*)
      type(c_ptr), intent(out) :: comms(*)
      integer(c_int) :: ierr
    end function
    function fem_comm_create_callbacks(ctx, nranks, rank, allreduce, exchange, user, comm) &
        bind(C, name="fem_comm_create_callbacks") result(ierr)
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: ctx, user
      integer(c_int), value :: nranks, rank
      type(c_funptr), value :: allreduce, exchange
      type(c_ptr), intent(out) :: comm
      integer(c_int) :: ierr
    end function
    function fem_comm_get_stats(comm, stats) bind(C, name="fem_comm_get_stats") result(ierr)
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: comm
      integer(c_long_long), intent(out) :: stats(6)
      integer(c_int) :: ierr
    end function
    function fem_comm_destroy(comm) bind(C, name="fem_comm_destroy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: comm
      integer(c_int) :: ierr
    end function
    function fem_comm_rank(comm, rank, nranks) bind(C, name="fem_comm_rank") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: comm
      integer(c_int), intent(out) :: rank, nranks
      integer(c_int) :: ierr
    end function
    function fem_comm_exchange_allreduce(comm, nneigh, neigh_rank, off, send, recv, red, nred) &
        bind(C, name="fem_comm_exchange_allreduce") result(ierr)
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: comm, send, recv, red
      integer(c_int), value :: nneigh, nred
      integer(c_int), intent(in) :: neigh_rank(*)
      integer(c_long_long), intent(in) :: off(*)
      integer(c_int) :: ierr
    end function
    function fem_comm_exchange(comm, nneigh, neigh_rank, off, send, recv) bind(C, name="fem_comm_exchange") result(ierr)
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: comm, send, recv
      integer(c_int), value :: nneigh
      integer(c_int), intent(in) :: neigh_rank(*)
      integer(c_long_long), intent(in) :: off(*)
      integer(c_int) :: ierr
    end function
    ! interface lists of this rank (what nl2g, src/main.F90:78-96, already holds); 0-based node ids
    function fem_mesh_set_halo(mesh, comm, nneigh, neigh_rank, neigh_ptr, neigh_nodes) &
        bind(C, name="fem_mesh_set_halo") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mesh, comm
      integer(c_int), value :: nneigh
      integer(c_int), intent(in) :: neigh_rank(*), neigh_ptr(*), neigh_nodes(*)
      integer(c_int) :: ierr
    end function
    function fem_coh_apply_1(ctx, ncoh, ecoords, u8, props, dt, tract, stiff, gp) bind(C, name="fem_coh_apply_1") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: ncoh
      real(c_double), intent(in) :: ecoords(*), u8(*), props(*)
      real(c_double), value :: dt
      real(c_double), intent(out) :: tract(*), stiff(*), gp(*)
      integer(c_int) :: ierr
    end function
    function fem_rcm_order(n, ptr, adj, perm) bind(C, name="fem_rcm_order") result(ierr)
      import :: c_int
      integer(c_int), value :: n
      integer(c_int), intent(in) :: ptr(*), adj(*)
      integer(c_int), intent(out) :: perm(*)
      integer(c_int) :: ierr
    end function
    ! MatView-style access (src/m_global.F90:702-741 GetStiffnessArray, without the dense copy)
    function fem_mat_get_csr(mat, rowptr, colind, val) bind(C, name="fem_mat_get_csr") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mat
      integer(c_int), intent(out) :: rowptr(*), colind(*)
      real(c_double), intent(out) :: val(*)
      integer(c_int) :: ierr
    end function
    function fem_mat_set_mult_type(mat, mtype) bind(C, name="fem_mat_set_mult_type") result(ierr)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: mat
      character(kind=c_char), intent(in) :: mtype(*)
      integer(c_int) :: ierr
    end function
    function fem_mat_set_profiling(mat, on) bind(C, name="fem_mat_set_profiling") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mat
      integer(c_int), value :: on
      integer(c_int) :: ierr
    end function
    function fem_mat_get_profile(mat, max_products, n, element_ms, gather_ms, penalty_ms) &
        bind(C, name="fem_mat_get_profile") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mat
      integer(c_int), value :: max_products
      integer(c_int), intent(out) :: n
      real(c_double), intent(out) :: element_ms, gather_ms, penalty_ms
      integer(c_int) :: ierr
    end function
    function fem_mat_get_mult_info(mat, kind, tiles, affine_tiles, element_visits, lane_slots, tile_nodes, setup_ms) &
        bind(C, name="fem_mat_get_mult_info") result(ierr)
      import :: c_int, c_ptr, c_double, c_long_long
      type(c_ptr), value :: mat
      integer(c_int), intent(out) :: kind
      integer(c_long_long), intent(out) :: tiles, affine_tiles, element_visits, lane_slots, tile_nodes
      real(c_double), intent(out) :: setup_ms
      integer(c_int) :: ierr
    end function
    function fem_mat_get_tile_geometry_info(mat, uniform_tiles, per_tile_records, streaming_lists) &
        bind(C, name="fem_mat_get_tile_geometry_info") result(ierr)
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: mat
      integer(c_long_long), intent(out) :: uniform_tiles
      integer(c_int), intent(out) :: per_tile_records, streaming_lists
      integer(c_int) :: ierr
    end function
    function fem_mesh_tile_plan_digest(mesh, builder, digest, setup_ms) bind(C, name="fem_mesh_tile_plan_digest") result(ierr)
      import :: c_int, c_ptr, c_double, c_long_long, c_char
      type(c_ptr), value :: mesh
      character(kind=c_char), intent(in) :: builder(*)
      integer(c_long_long), intent(out) :: digest(12)
      real(c_double), intent(out) :: setup_ms
      integer(c_int) :: ierr
    end function
    function fem_mat_set_values(mat, m, rows, n, cols, v, mode) bind(C, name="fem_mat_set_values") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mat
      integer(c_int), value :: m, n, mode
      integer(c_int), intent(in) :: rows(*), cols(*)
      real(c_double), intent(in) :: v(*)
      integer(c_int) :: ierr
    end function
    function fem_mat_get_diagonal(mat, d) bind(C, name="fem_mat_get_diagonal") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mat, d
      integer(c_int) :: ierr
    end function
    function fem_mat_mult_timed(mat, x, y, reps, avg_ms) bind(C, name="fem_mat_mult_timed") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mat, x, y
      integer(c_int), value :: reps
      real(c_double), intent(out) :: avg_ms
      integer(c_int) :: ierr
    end function
    function fem_vec_get_size(v, n) bind(C, name="fem_vec_get_size") result(ierr)
      import :: c_int, c_ptr, c_long_long
      type(c_ptr), value :: v
      integer(c_long_long), intent(out) :: n
      integer(c_int) :: ierr
    end function
    function fem_vec_device_ptr(v, dptr) bind(C, name="fem_vec_device_ptr") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: v
      type(c_ptr), intent(out) :: dptr
      integer(c_int) :: ierr
    end function
    function fem_vec_copy(src, dst) bind(C, name="fem_vec_copy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: src, dst
      integer(c_int) :: ierr
    end function
    function fem_vec_axpy(y, a, x) bind(C, name="fem_vec_axpy") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: y, x
      real(c_double), value :: a
      integer(c_int) :: ierr
    end function
    function fem_vec_dot(x, y, dot) bind(C, name="fem_vec_dot") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: x, y
      real(c_double), intent(out) :: dot
      integer(c_int) :: ierr
    end function
    function fem_vec_norm2(x, nrm) bind(C, name="fem_vec_norm2") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: x
      real(c_double), intent(out) :: nrm
      integer(c_int) :: ierr
    end function
    function fem_ksp_get_residual_history(ksp, hist, len, n) bind(C, name="fem_ksp_get_residual_history") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ksp
      real(c_double), intent(out) :: hist(*)
      integer(c_int), value :: len
      integer(c_int), intent(out) :: n
      integer(c_int) :: ierr
    end function
    function fem_ksp_set_profiling(ksp, on) bind(C, name="fem_ksp_set_profiling") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp
      integer(c_int), value :: on
      integer(c_int) :: ierr
    end function
    function fem_ksp_get_profile(ksp, n_spmv, spmv_ms, solve_ms) bind(C, name="fem_ksp_get_profile") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ksp
      integer(c_int), intent(out) :: n_spmv
      real(c_double), intent(out) :: spmv_ms, solve_ms
      integer(c_int) :: ierr
    end function
    function fem_ksp_get_reused_solves(ksp, n) bind(C, name="fem_ksp_get_reused_solves") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp
      integer(c_int), intent(out) :: n
      integer(c_int) :: ierr
    end function
    function fem_ksp_get_resident_solves(ksp, n) bind(C, name="fem_ksp_get_resident_solves") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ksp
      integer(c_int), intent(out) :: n
      integer(c_int) :: ierr
    end function
    function fem_recover_nodal_stress_vec(mesh, U, S) bind(C, name="fem_recover_nodal_stress_vec") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: mesh, U, S
      integer(c_int) :: ierr
    end function
    function fem_shape_func(eltype, gp, N) bind(C, name="fem_shape_func") result(ierr)
      import :: c_int, c_double
      integer(c_int), value :: eltype, gp
      real(c_double), intent(out) :: N(*)
      integer(c_int) :: ierr
    end function
    function fem_shape_funcd(ctx, eltype, npts, coord, dN) bind(C, name="fem_shape_funcd") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: eltype, npts
      real(c_double), intent(in) :: coord(*)
      real(c_double), intent(out) :: dN(*)
      integer(c_int) :: ierr
    end function
    function fem_form_dn_detj(ctx, eltype, nel, ipcoord, ecoords, dN, detj) bind(C, name="fem_form_dn_detj") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: eltype, nel
      real(c_double), intent(in) :: ipcoord(*), ecoords(*)
      real(c_double), intent(out) :: dN(*), detj(*)
      integer(c_int) :: ierr
    end function
    function fem_bmat(ctx, pdim, nodecount, nel, dN, B) bind(C, name="fem_bmat") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: pdim, nodecount, nel
      real(c_double), intent(in) :: dN(*)
      real(c_double), intent(out) :: B(*)
      integer(c_int) :: ierr
    end function
    function fem_dmat(ctx, pdim, nmat, E, nu, D) bind(C, name="fem_dmat") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx
      integer(c_int), value :: pdim, nmat
      real(c_double), intent(in) :: E(*), nu(*)
      real(c_double), intent(out) :: D(*)
      integer(c_int) :: ierr
    end function
    ! ---- cohesive elements + Newton (SNES): src/main.F90:197-206,240-245, src/m_global.F90:98-235 ----
    function fem_mesh_set_cohesive(mesh, nlmat, ncohmats, props) bind(C, name="fem_mesh_set_cohesive") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mesh
      integer(c_int), intent(in) :: nlmat(*)
      integer(c_int), value :: ncohmats
      real(c_double), intent(in) :: props(*)
      integer(c_int) :: ierr
    end function
    function fem_mat_duplicate(A, B) bind(C, name="fem_mat_duplicate") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: A
      type(c_ptr), intent(out) :: B
      integer(c_int) :: ierr
    end function
    function fem_mat_copy(src, dst) bind(C, name="fem_mat_copy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: src, dst
      integer(c_int) :: ierr
    end function
    function fem_mat_add_cohesive(J, u_total, dt, rels_mode, rels_nodes) bind(C, name="fem_mat_add_cohesive") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: J, u_total
      real(c_double), value :: dt
      integer(c_int), value :: rels_mode
      integer(c_int), intent(in) :: rels_nodes(*)
      integer(c_int) :: ierr
    end function
    function fem_vec_add_cohesive_traction(mesh, u_total, dt, rels_mode, rels_nodes, R) &
        bind(C, name="fem_vec_add_cohesive_traction") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mesh, u_total, R
      real(c_double), value :: dt
      integer(c_int), value :: rels_mode
      integer(c_int), intent(in) :: rels_nodes(*)
      integer(c_int) :: ierr
    end function
    function fem_enforce_bc_force(mesh, nbcs, bcnode, bcval, dt, t_total, R) bind(C, name="fem_enforce_bc_force") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: mesh, R
      integer(c_int), value :: nbcs
      integer(c_int), intent(in) :: bcnode(*)
      real(c_double), intent(in) :: bcval(*)
      real(c_double), value :: dt, t_total
      integer(c_int) :: ierr
    end function
    function fem_mat_mult_add(A, x, y, z) bind(C, name="fem_mat_mult_add") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: A, x, y, z
      integer(c_int) :: ierr
    end function
    function fem_vec_scale(v, alpha) bind(C, name="fem_vec_scale") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: v
      real(c_double), value :: alpha
      integer(c_int) :: ierr
    end function
    function fem_snes_create(ctx, snes) bind(C, name="fem_snes_create") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: ctx
      type(c_ptr), intent(out) :: snes
      integer(c_int) :: ierr
    end function
    function fem_snes_destroy(snes) bind(C, name="fem_snes_destroy") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: snes
      integer(c_int) :: ierr
    end function
    function fem_snes_get_ksp(snes, ksp) bind(C, name="fem_snes_get_ksp") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: snes
      type(c_ptr), intent(out) :: ksp
      integer(c_int) :: ierr
    end function
    ! the callbacks are bind(C) functions of the host: integer(c_int) function f(snes, x, f, ctx) / j(snes, x, J, P, ctx)
    function fem_snes_set_function(snes, r, f, ctx) bind(C, name="fem_snes_set_function") result(ierr)
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: snes, r, ctx
      type(c_funptr), value :: f
      integer(c_int) :: ierr
    end function
    function fem_snes_set_jacobian(snes, J, P, jf, ctx) bind(C, name="fem_snes_set_jacobian") result(ierr)
      import :: c_int, c_ptr, c_funptr
      type(c_ptr), value :: snes, J, P, ctx
      type(c_funptr), value :: jf
      integer(c_int) :: ierr
    end function
    function fem_snes_set_defmod_callbacks(snes, mesh, K, J, R, F, aggregate_u, work, nbcs, bcnode, bcval, dt, t_total, &
                                           rels_mode, rels_nodes, state) bind(C, name="fem_snes_set_defmod_callbacks") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: snes, mesh, K, J, R, F, aggregate_u, work
      integer(c_int), value :: nbcs, rels_mode
      integer(c_int), intent(in) :: bcnode(*)
      real(c_double), intent(in) :: bcval(*)
      real(c_double), value :: dt, t_total
      type(c_ptr), value :: rels_nodes          ! c_loc of four 0-based node ids, or c_null_ptr
      type(c_ptr), intent(out) :: state
      integer(c_int) :: ierr
    end function
    function fem_snes_calc_defmod_jacobian(state, du) bind(C, name="fem_snes_calc_defmod_jacobian") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: state, du
      integer(c_int) :: ierr
    end function
    function fem_snes_free_defmod_callbacks(state) bind(C, name="fem_snes_free_defmod_callbacks") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: state
      integer(c_int) :: ierr
    end function
    function fem_snes_set_from_options(snes, options) bind(C, name="fem_snes_set_from_options") result(ierr)
      import :: c_int, c_ptr, c_char
      type(c_ptr), value :: snes
      character(kind=c_char), intent(in) :: options(*)
      integer(c_int) :: ierr
    end function
    function fem_snes_solve(snes, x) bind(C, name="fem_snes_solve") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: snes, x
      integer(c_int) :: ierr
    end function
    function fem_snes_get_iteration_number(snes, its) bind(C, name="fem_snes_get_iteration_number") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: snes
      integer(c_int), intent(out) :: its
      integer(c_int) :: ierr
    end function
    function fem_snes_get_converged_reason(snes, reason) bind(C, name="fem_snes_get_converged_reason") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: snes
      integer(c_int), intent(out) :: reason
      integer(c_int) :: ierr
    end function
    function fem_snes_get_function_norm(snes, fnorm) bind(C, name="fem_snes_get_function_norm") result(ierr)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: snes
      real(c_double), intent(out) :: fnorm
      integer(c_int) :: ierr
    end function
    function fem_snes_get_linear_solve_iterations(snes, lits) bind(C, name="fem_snes_get_linear_solve_iterations") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: snes
      integer(c_int), intent(out) :: lits
      integer(c_int) :: ierr
    end function
    function fem_snes_get_number_function_evals(snes, nfuncs) bind(C, name="fem_snes_get_number_function_evals") result(ierr)
      import :: c_int, c_ptr
      type(c_ptr), value :: snes
      integer(c_int), intent(out) :: nfuncs
      integer(c_int) :: ierr
    end function
  end interface

contains

  ! The reference never checks ierr (src/m_global.F90:42); this wrapper does.
  subroutine fem_check(ierr, where)
    integer(c_int), intent(in) :: ierr
    character(*), intent(in) :: where
    character(kind=c_char), pointer :: cmsg(:)
    type(c_ptr) :: p
    integer :: i
    if (ierr == 0) return
    p = fem_last_error()
    write(*, '(A,A,A,I0)', advance='no') " libfem_amd error in ", where, ": code ", ierr
    if (c_associated(p)) then
      call c_f_pointer(p, cmsg, (/512/))
      write(*, '(A)', advance='no') " : "
      do i = 1, 512
        if (cmsg(i) == c_null_char) exit
        write(*, '(A)', advance='no') cmsg(i)
      end do
    end if
    write(*, *)
    error stop 1
  end subroutine fem_check

  integer(c_int) function fem_type_no(eltype) result(t)
    character(*), intent(in) :: eltype
    t = 0
    select case (trim(eltype))
    case ("tri"); t = FEM_TRI
    case ("qua"); t = FEM_QUA
    case ("tet"); t = FEM_TET
    case ("hex"); t = FEM_HEX
    case ("coh"); t = FEM_COH
    end select
  end function fem_type_no

end module fem_amd_capi
