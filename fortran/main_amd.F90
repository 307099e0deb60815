! fortran/main_amd.F90 -- a `main.F90`-shaped host driver over libfem_amd.so.
!
! Same phases, banners and increment loop as the reference's program main
! (src/main.F90:26-336), with every PETSc Mat/Vec/KSP call replaced by the
! batched C-ABI call that INTEGRATION.md pairs it with.  Single process / one
! GPU (the reference's METIS partition + element shipping, src/m_io.F90:107-237,
! has no counterpart here; multi-GPU runs use generated bricks, see bench.py).
!
!   defmod_amd -f deck.inp [-ksp_rtol 1e-12 ...] [-vec_view] [-o prefix]
!
! Deck grammar: the reference's current reader (src/m_io.F90:12-20,71-105,
! 282-475; SURVEY.md appendix B), read with the same list-directed READs.
program main_amd
  use iso_c_binding
  use fem_amd_capi
  implicit none

  character(256) :: input_file, arg, prefix
  character(1024) :: options
  character(12) :: stype
  character(4) :: et
  integer :: pdim, cpdim, nodal_bw, nels, nnds, nmts, ncohmats, nceqs, nfrcs, ntrcs, nbcs
  real(8) :: t, dt, t_init
  integer :: i, j, n, nargs, dtNo, seplaw, tno, its, reason
  logical :: vec_view
  integer(c_int), allocatable :: eltype(:), eptr(:), conn(:), mat_id(:), nlmat(:), bcnode(:), fnode(:), telsd(:)
  integer, allocatable :: nodes(:), bcrow(:)
  real(c_double), allocatable :: coords(:), mats(:), bcval(:), fval(:), tval(:), uu(:), aggregate_u(:)
  real(c_double), allocatable :: stress_node(:), aggregate_stress(:), row(:), props(:)
  type(c_ptr) :: ctx, mesh, Mat_K, Vec_F, Vec_U, Krylov
  integer(c_long_long) :: nrows, nnz, dbytes
  real(c_double) :: rnorm

  input_file = ""; options = ""; prefix = "0_output"; vec_view = .false.
  nargs = command_argument_count()
  i = 1
  do while (i <= nargs)
    call get_command_argument(i, arg)
    if (trim(arg) == "-f" .and. i < nargs) then
      call get_command_argument(i + 1, input_file); i = i + 2
    else if (trim(arg) == "-o" .and. i < nargs) then
      call get_command_argument(i + 1, prefix); i = i + 2
    else if (trim(arg) == "-vec_view") then
      vec_view = .true.; i = i + 1
    else
      options = trim(options)//" "//trim(arg); i = i + 1
    end if
  end do
  if (len_trim(input_file) == 0) then
    print *, "Usage: defmod_amd -f <filename> [-ksp_rtol <tol> ...]"   ! src/main.F90:32
    stop
  end if

  ! ---- input (src/main.F90:36-103) ----
  print *, "Reading input ..."
  open(10, file=trim(input_file), status='old')
  read(10, *) stype, pdim, nodal_bw
  read(10, *) nels, nnds, nmts, ncohmats, nceqs, nfrcs, ntrcs, nbcs
  read(10, *) t, dt
  cpdim = 3; if (pdim == 3) cpdim = 6
  if (nceqs /= 0) error stop "nceqs must be 0"
  allocate(eltype(nels), eptr(nels + 1), conn(8*nels), mat_id(nels), nlmat(nels))
  eptr(1) = 0
  do i = 1, nels
    read(10, *) et
    backspace(10)
    tno = fem_type_no(et)
    if (tno == 0) error stop "unknown element type"
    n = 4; if (tno == FEM_TRI) n = 3; if (tno == FEM_HEX) n = 8
    allocate(nodes(n))
    read(10, *) et, nodes, mat_id(i), nlmat(i)
    eltype(i) = tno
    conn(eptr(i) + 1:eptr(i) + n) = nodes - 1
    eptr(i + 1) = eptr(i) + n
    mat_id(i) = mat_id(i) - 1
    deallocate(nodes)
  end do
  allocate(coords(nnds*pdim), row(pdim))
  do i = 1, nnds
    read(10, *) row
    coords((i - 1)*pdim + 1:i*pdim) = row
  end do
  deallocate(row)
  allocate(mats(5*max(nmts, 1)), row(5))
  do i = 1, nmts
    read(10, *) row
    mats((i - 1)*5 + 1:i*5) = row
  end do
  deallocate(row)
  do i = 1, ncohmats                       ! read and ignored: cohesive path is out of scope here
    read(10, *) seplaw
    backspace(10)
    allocate(props(6))
    read(10, *) seplaw, props
    deallocate(props)
  end do
  allocate(bcnode(max(nbcs, 1)*(1 + pdim)), bcval(max(nbcs, 1)*pdim), bcrow(1 + pdim), row(pdim))
  do i = 1, nbcs
    read(10, *) bcrow, row
    bcrow(1) = bcrow(1) - 1
    bcnode((i - 1)*(1 + pdim) + 1:i*(1 + pdim)) = bcrow
    bcval((i - 1)*pdim + 1:i*pdim) = row
  end do
  deallocate(row)
  allocate(fnode(max(nfrcs, 1)), fval(max(nfrcs, 1)*(pdim + 2)), row(pdim + 2))
  do i = 1, nfrcs
    read(10, *) j, row
    row(pdim + 1) = min(row(pdim + 1), t)  ! src/m_io.F90:357-358
    row(pdim + 2) = min(row(pdim + 2), t)
    fnode(i) = j - 1
    fval((i - 1)*(pdim + 2) + 1:i*(pdim + 2)) = row
  end do
  allocate(telsd(max(ntrcs, 1)*2), tval(max(ntrcs, 1)*(pdim + 2)))
  do i = 1, ntrcs
    read(10, *) telsd(2*i - 1), telsd(2*i), row
    telsd(2*i - 1) = telsd(2*i - 1) - 1
    tval((i - 1)*(pdim + 2) + 1:i*(pdim + 2)) = row
  end do
  deallocate(row)
  close(10)

  ! ---- device objects (PetscInitialize, VecCreateMPI, MatCreateAIJ: src/main.F90:26,138,154) ----
  call fem_check(fem_ctx_create(0_c_int, ctx), "fem_ctx_create")
  call fem_check(fem_mesh_create(ctx, pdim, nels, nnds, eltype, eptr, conn, mat_id, coords, nmts, mats, mesh), &
                 "fem_mesh_create")
  n = pdim*nnds
  call fem_check(fem_vec_create(ctx, int(n, c_long_long), Vec_U), "fem_vec_create")
  call fem_check(fem_vec_create(ctx, int(n, c_long_long), Vec_F), "fem_vec_create")
  allocate(uu(n), aggregate_u(n), stress_node(nnds*cpdim), aggregate_stress(nnds*cpdim))
  aggregate_u = 0; aggregate_stress = 0

  print *, "Forming [K] ..."                                             ! src/main.F90:151-171
  call fem_check(fem_mat_create_aij(mesh, Mat_K), "fem_mat_create_aij")
  call fem_check(fem_mat_assemble_elements(Mat_K), "fem_mat_assemble_elements")
  call fem_check(fem_mat_apply_kbc(Mat_K, nbcs, bcnode), "fem_mat_apply_kbc")
  call fem_check(fem_mat_get_info(Mat_K, nrows, nnz, dbytes), "fem_mat_get_info")
  write(*, '(A,I0,A,I0,A)') "  [K]: ", nrows, " rows, ", nnz, " nonzeros"

  print *, "Setting up environment ..."                                  ! SetupKSPSolver, src/main.F90:339-403
  call fem_check(fem_ksp_create(ctx, Krylov), "fem_ksp_create")
  call fem_check(fem_ksp_set_operators(Krylov, Mat_K), "fem_ksp_set_operators")
  call fem_check(fem_ksp_set_type(Krylov, "cg"//c_null_char), "fem_ksp_set_type")
  call fem_check(fem_ksp_set_pc_type(Krylov, "jacobi"//c_null_char), "fem_ksp_set_pc_type")
  call fem_check(fem_ksp_set_tolerances(Krylov, real(1.e-9, c_double), FEM_DEFAULT_REAL, FEM_DEFAULT_REAL, &
                 FEM_DEFAULT_INT), "fem_ksp_set_tolerances")            ! tol = 1.e-9, src/main.F90:349
  call fem_check(fem_ksp_set_from_options(Krylov, trim(options)//c_null_char), "fem_ksp_set_from_options")
  call fem_check(fem_ksp_set_up(Krylov), "fem_ksp_set_up")

  ! ---- increment loop (src/main.F90:216-296) ----
  dtNo = 1
  do
    t_init = dt*(dtNo - 1)
    if (t_init >= t) exit
    write(*, '(A,I0)') " Interval: ", dtNo
    print *, "    Forming RHS ..."
    call fem_check(fem_form_rhs(mesh, nbcs, bcnode, bcval, nfrcs, fnode, fval, ntrcs, telsd, tval, t, t_init, dt, &
                   Vec_F), "fem_form_rhs")
    print *, "    Solving ..."
    call fem_check(fem_ksp_solve(Krylov, Vec_F, Vec_U), "fem_ksp_solve")
    call fem_check(fem_ksp_get_iteration_number(Krylov, its), "its")
    call fem_check(fem_ksp_get_converged_reason(Krylov, reason), "reason")
    call fem_check(fem_ksp_get_residual_norm(Krylov, rnorm), "rnorm")
    write(*, '(A,I0,A,I0,A,ES10.3)') "     KSP iterations: ", its, "  reason: ", reason, "  |M^-1 r|: ", rnorm
    if (reason < 0) error stop "KSP diverged"
    call fem_check(fem_vec_get_array(Vec_U, uu), "fem_vec_get_array")    ! GetVec_U, src/main.F90:250
    if (vec_view) then
      do i = 1, n
        write(*, '(ES23.15)') uu(i)
      end do
    end if
    aggregate_u = aggregate_u + uu
    print *, "    Getting stress ..."
    call fem_check(fem_recover_nodal_stress(mesh, Vec_U, stress_node), "fem_recover_nodal_stress")
    aggregate_stress = aggregate_stress + stress_node
    dtNo = dtNo + 1
  end do

  call WriteOutput()

  print *, "Cleaning up ..."
  call fem_check(fem_ksp_destroy(Krylov), "fem_ksp_destroy")
  call fem_check(fem_vec_destroy(Vec_F), "fem_vec_destroy")
  call fem_check(fem_vec_destroy(Vec_U), "fem_vec_destroy")
  call fem_check(fem_mat_destroy(Mat_K), "fem_mat_destroy")
  call fem_check(fem_mesh_destroy(mesh), "fem_mesh_destroy")
  call fem_check(fem_ctx_destroy(ctx), "fem_ctx_destroy")
  print *, "Finished"

contains

  ! Legacy ASCII VTK in the reference's layout (WriteOutput, src/m_io.F90:480-555):
  ! POINTS F0.3, CELLS "n id0..", CELL_TYPES vtk ids, SCALARS STRESS FLOAT cpdim
  ! F0.6, VECTORS displacements F0.6; plus a full-precision companion file.
  subroutine WriteOutput()
    character(300) :: name
    character(32) :: fmt
    integer :: e, k, ncol, vtk, dum1, dum2, dum3, dum4, nn
    real(8) :: p(3)
    name = trim(prefix)//"_000000.vtk"
    open(11, file=trim(name), status='replace')
    write(11, '(A)') "# vtk DataFile Version 2.0"
    write(11, '(A)') "File written by Defmod"
    write(11, '(A)') "ASCII"
    write(11, '(A)') "DATASET UNSTRUCTURED_GRID"
    write(11, '(A,I0,A)') "POINTS ", nnds, " double"
    do k = 1, nnds
      p = 0
      p(1:pdim) = coords((k - 1)*pdim + 1:k*pdim)
      write(11, '(3(F0.3,1X))') p
    end do
    ncol = 0
    do e = 1, nels
      ncol = ncol + (eptr(e + 1) - eptr(e)) + 1
    end do
    write(11, '(A,I0,1X,I0)') "CELLS ", nels, ncol
    do e = 1, nels
      nn = eptr(e + 1) - eptr(e)
      write(fmt, '(A,I0,A)') "(I0,", nn, "(1X,I0))"
      write(11, fmt) nn, conn(eptr(e) + 1:eptr(e + 1))
    end do
    write(11, '(A,I0)') "CELL_TYPES ", nels
    do e = 1, nels
      call fem_check(fem_el_counts(eltype(e), dum1, dum2, dum3, dum4, vtk), "fem_el_counts")
      write(11, '(I0)') vtk
    end do
    write(11, '(A,I0)') "POINT_DATA ", nnds
    write(11, '(A,I0)') "SCALARS STRESS FLOAT ", cpdim
    write(11, '(A)') "LOOKUP_TABLE DEFAULT"
    write(fmt, '(A,I0,A)') "(", cpdim, "(F0.6,1X))"
    do k = 1, nnds
      write(11, fmt) (aggregate_stress(k + (j - 1)*nnds), j=1, cpdim)
    end do
    write(11, '(A)') "VECTORS displacements double"
    do k = 1, nnds
      p = 0
      p(1:pdim) = aggregate_u((k - 1)*pdim + 1:k*pdim)
      write(11, '(3(F0.6,1X))') p
    end do
    close(11)
    ! full precision (the 6-digit VTK cannot pin 1e-10 parity)
    name = trim(prefix)//"_000000.full.txt"
    open(12, file=trim(name), status='replace')
    write(12, '(I0,1X,I0,1X,I0)') nnds, pdim, cpdim
    do k = 1, nnds
      write(12, '(9(ES24.16,1X))') aggregate_u((k - 1)*pdim + 1:k*pdim), &
                                    (aggregate_stress(k + (j - 1)*nnds), j=1, cpdim)
    end do
    close(12)
  end subroutine WriteOutput

end program main_amd
