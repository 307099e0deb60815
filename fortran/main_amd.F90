! fortran/main_amd.F90 -- a `main.F90`-shaped host driver over libfem_amd.so.
!
! Same phases, banners and increment loop as the reference's program main (src/main.F90:26-336), with every PETSc
! Mat/Vec/KSP call replaced by the batched C-ABI call that INTEGRATION.md pairs it with.
!
!   defmod_amd -f deck.inp [-ksp_rtol 1e-12 ...] [-vec_view] [-o prefix] [-np N]
!
! One domain (default) or several, as the reference under `mpiexec -n 2` (Makefile:6):
!   -np N        N domains in this process, one OpenMP host thread and one device context each, the in-process
!                communicator between them (any number of GPUs, including one);
!   one process per GPU over RCCL/xGMI: start N copies with FEM_AMD_RANK / FEM_AMD_NRANKS (or RANK / WORLD_SIZE)
!                and FEM_AMD_UID_FILE set in the environment; rank 0 writes the communicator id to that file, the
!                others read it.  Rank and size are read before any GPU call; device = FEM_AMD_DEVICE, LOCAL_RANK or 0.
! Every rank reads the deck (the reference reads on rank 0 and ships elements, src/m_io.F90:107-237), partitions the
! elements by recursive coordinate bisection of their centroids (METIS_PartMeshNodal, src/m_io.F90:137, is not
! available), keeps its elements with their nodes renumbered in ascending global id (src/main.F90:49-96) and hands
! the library, per neighbouring rank, the shared nodes in ascending global id (fem_mesh_set_halo).  Output: one
! legacy VTK file per rank, <rank>_output_000000.vtk (src/m_io.F90:497).
!
! Deck grammar: the reference's current reader (src/m_io.F90:12-20,71-105, 282-475; SURVEY.md appendix B), read
! with the same list-directed READs.
module defmod_deck
  use iso_c_binding
  use fem_amd_capi
  implicit none

  type deck_t
    character(12) :: stype
    integer :: pdim, cpdim, nodal_bw, nels, nnds, nmts, ncohmats, nceqs, nfrcs, ntrcs, nbcs
    real(8) :: t, dt
    integer(c_int), allocatable :: eltype(:), eptr(:), conn(:), mat_id(:), nlmat(:), bcnode(:), fnode(:), telsd(:)
    real(c_double), allocatable :: coords(:), mats(:), bcval(:), fval(:), tval(:)
  end type deck_t

  ! -resolve_rtol <tol>: after the last increment, KSPSetTolerances(tol) (src/main.F90:349-351) and KSPSolve of the same Vec_F
  ! again -- a parity check's "tighten and re-solve"; the line it prints carries the iteration count and how many solves of this
  ! KSP were answered from a kept solution (-ksp_reuse_solution)
  real(c_double), save :: resolve_rtol = -1.0_c_double

contains

  ! ReadParameters / ReadElementsCoords / ReadDist* (src/m_io.F90:12-20,71-105,282-475); ids become 0-based
  subroutine read_deck(path, d)
    character(*), intent(in) :: path
    type(deck_t), intent(out) :: d
    character(4) :: et
    integer :: i, j, n, tno, seplaw
    integer, allocatable :: nodes(:), bcrow(:)
    real(8), allocatable :: row(:), props(:)
    open(10, file=trim(path), status='old')
    read(10, *) d%stype, d%pdim, d%nodal_bw
    read(10, *) d%nels, d%nnds, d%nmts, d%ncohmats, d%nceqs, d%nfrcs, d%ntrcs, d%nbcs
    read(10, *) d%t, d%dt
    d%cpdim = 3; if (d%pdim == 3) d%cpdim = 6
    if (d%nceqs /= 0) error stop "nceqs must be 0"
    allocate(d%eltype(d%nels), d%eptr(d%nels + 1), d%conn(8*d%nels), d%mat_id(d%nels), d%nlmat(d%nels))
    d%eptr(1) = 0
    do i = 1, d%nels
      read(10, *) et
      backspace(10)
      tno = fem_type_no(et)
      if (tno == 0) error stop "unknown element type"
      n = 4; if (tno == FEM_TRI) n = 3; if (tno == FEM_HEX) n = 8
      allocate(nodes(n))
      read(10, *) et, nodes, d%mat_id(i), d%nlmat(i)
      d%eltype(i) = tno
      d%conn(d%eptr(i) + 1:d%eptr(i) + n) = nodes - 1
      d%eptr(i + 1) = d%eptr(i) + n
      d%mat_id(i) = max(d%mat_id(i) - 1, 0)
      deallocate(nodes)
    end do
    allocate(d%coords(d%nnds*d%pdim), row(d%pdim))
    do i = 1, d%nnds
      read(10, *) row
      d%coords((i - 1)*d%pdim + 1:i*d%pdim) = row
    end do
    deallocate(row)
    allocate(d%mats(5*max(d%nmts, 1)), row(5))
    do i = 1, d%nmts
      read(10, *) row
      d%mats((i - 1)*5 + 1:i*5) = row
    end do
    deallocate(row)
    do i = 1, d%ncohmats                   ! read and skipped: the Newton branch is driven from fem_amd/driver.py
      allocate(props(6))
      read(10, *) seplaw, props
      deallocate(props)
    end do
    allocate(d%bcnode(max(d%nbcs, 1)*(1 + d%pdim)), d%bcval(max(d%nbcs, 1)*d%pdim), bcrow(1 + d%pdim), row(d%pdim))
    do i = 1, d%nbcs
      read(10, *) bcrow, row
      bcrow(1) = bcrow(1) - 1
      d%bcnode((i - 1)*(1 + d%pdim) + 1:i*(1 + d%pdim)) = bcrow
      d%bcval((i - 1)*d%pdim + 1:i*d%pdim) = row
    end do
    deallocate(row)
    allocate(d%fnode(max(d%nfrcs, 1)), d%fval(max(d%nfrcs, 1)*(d%pdim + 2)), row(d%pdim + 2))
    do i = 1, d%nfrcs
      read(10, *) j, row
      row(d%pdim + 1) = min(row(d%pdim + 1), d%t)  ! src/m_io.F90:357-358
      row(d%pdim + 2) = min(row(d%pdim + 2), d%t)
      d%fnode(i) = j - 1
      d%fval((i - 1)*(d%pdim + 2) + 1:i*(d%pdim + 2)) = row
    end do
    allocate(d%telsd(max(d%ntrcs, 1)*2), d%tval(max(d%ntrcs, 1)*(d%pdim + 2)))
    do i = 1, d%ntrcs
      read(10, *) d%telsd(2*i - 1), d%telsd(2*i), row
      d%telsd(2*i - 1) = d%telsd(2*i - 1) - 1
      d%tval((i - 1)*(d%pdim + 2) + 1:i*(d%pdim + 2)) = row
    end do
    deallocate(row)
    close(10)
  end subroutine read_deck

  ! Element -> part by recursive coordinate bisection of the element centroids (stands in for METIS_PartMeshNodal,
  ! src/m_io.F90:137): the longest axis is cut, ties broken by element number; deterministic on every rank.
  recursive subroutine rcb(idx, cent, pdim, p0, np, epart)
    integer, intent(in) :: idx(:), pdim, p0, np
    real(8), intent(in) :: cent(:, :)
    integer, intent(inout) :: epart(:)
    integer :: n, ax, i, j, cut, nl, tmp
    integer, allocatable :: ord(:)
    real(8) :: ext, e1
    n = size(idx)
    if (np == 1 .or. n == 0) then
      epart(idx) = p0
      return
    end if
    ax = 1; ext = -1.0d0
    do i = 1, pdim
      e1 = maxval(cent(idx, i)) - minval(cent(idx, i))
      if (e1 > ext) then
        ext = e1; ax = i
      end if
    end do
    allocate(ord(n)); ord = idx
    do i = 2, n                            ! insertion sort by (coordinate, element number)
      tmp = ord(i); j = i - 1
      do while (j >= 1)
        if (cent(ord(j), ax) < cent(tmp, ax) .or. (cent(ord(j), ax) == cent(tmp, ax) .and. ord(j) < tmp)) exit
        ord(j + 1) = ord(j); j = j - 1
      end do
      ord(j + 1) = tmp
    end do
    nl = np/2
    cut = (n*nl)/np
    call rcb(ord(1:cut), cent, pdim, p0, nl, epart)
    call rcb(ord(cut + 1:n), cent, pdim, p0 + nl, np - nl, epart)
  end subroutine rcb

  subroutine partition_elements(d, nparts, epart)
    type(deck_t), intent(in) :: d
    integer, intent(in) :: nparts
    integer, allocatable, intent(out) :: epart(:)
    real(8), allocatable :: cent(:, :)
    integer, allocatable :: idx(:)
    integer :: e, a, n
    allocate(epart(d%nels), cent(d%nels, d%pdim), idx(d%nels))
    do e = 1, d%nels
      n = d%eptr(e + 1) - d%eptr(e)
      cent(e, :) = 0.0d0
      do a = 1, n
        cent(e, :) = cent(e, :) + d%coords(d%conn(d%eptr(e) + a)*d%pdim + 1:d%conn(d%eptr(e) + a)*d%pdim + d%pdim)
      end do
      cent(e, :) = cent(e, :)/n
      idx(e) = e
    end do
    epart = 0
    call rcb(idx, cent, d%pdim, 0, nparts, epart)
  end subroutine partition_elements

  ! One domain from setup to output: the body of the reference's program main for one rank.
  subroutine run_domain(d, epart, rank, nranks, ctx, comm, options, outdir, prefix, vec_view)
    type(deck_t), intent(in) :: d
    integer, intent(in) :: epart(:), rank, nranks
    type(c_ptr), intent(in) :: ctx, comm
    character(*), intent(in) :: options, outdir, prefix
    logical, intent(in) :: vec_view
    integer :: pdim, cpdim, nels, nnds, nbcs, nfrcs, ntrcs, i, j, k, e, a, n, q, dtNo, its, reason, nn, nneigh
    integer, allocatable :: g2l(:), nl2g(:), e2l(:), cnt(:)
    logical, allocatable :: mine(:), theirs(:)
    integer(c_int), allocatable :: eltype(:), eptr(:), conn(:), mat_id(:), bcnode(:), fnode(:), telsd(:)
    integer(c_int), allocatable :: neigh_rank(:), neigh_ptr(:), neigh_nodes(:)
    real(c_double), allocatable :: coords(:), bcval(:), fval(:), tval(:), uu(:), aggregate_u(:), stress_node(:), aggregate_stress(:)
    type(c_ptr) :: mesh, Mat_K, Vec_F, Vec_U, Krylov
    integer(c_long_long) :: nrows, nnz, dbytes
    real(c_double) :: rnorm
    real(8) :: t_init
    logical :: root
    root = rank == 0
    pdim = d%pdim; cpdim = d%cpdim

    ! ---- local elements and nodes (DistributeElements + renumbering, src/main.F90:44-96) ----
    allocate(g2l(d%nnds), mine(d%nnds)); mine = .false.
    nels = 0
    do e = 1, d%nels
      if (epart(e) /= rank) cycle
      nels = nels + 1
      do a = d%eptr(e) + 1, d%eptr(e + 1)
        mine(d%conn(a) + 1) = .true.
      end do
    end do
    nnds = count(mine)
    allocate(nl2g(nnds)); g2l = -1; k = 0
    do i = 1, d%nnds                        ! ascending global id
      if (mine(i)) then
        k = k + 1; nl2g(k) = i - 1; g2l(i) = k - 1
      end if
    end do
    allocate(eltype(max(nels, 1)), eptr(nels + 1), conn(8*max(nels, 1)), mat_id(max(nels, 1)), e2l(d%nels))
    eptr(1) = 0; k = 0; e2l = -1
    do e = 1, d%nels
      if (epart(e) /= rank) cycle
      k = k + 1; e2l(e) = k - 1
      n = d%eptr(e + 1) - d%eptr(e)
      eltype(k) = d%eltype(e); mat_id(k) = d%mat_id(e)
      do a = 1, n
        conn(eptr(k) + a) = g2l(d%conn(d%eptr(e) + a) + 1)
      end do
      eptr(k + 1) = eptr(k) + n
    end do
    allocate(coords(max(nnds, 1)*pdim))
    do i = 1, nnds
      coords((i - 1)*pdim + 1:i*pdim) = d%coords(nl2g(i)*pdim + 1:nl2g(i)*pdim + pdim)
    end do
    ! constraints and nodal forces on every rank that holds the node (the reference broadcasts them, src/m_io.F90:
    ! 362-363,402-403); tractions follow their element
    allocate(bcnode(max(d%nbcs, 1)*(1 + pdim)), bcval(max(d%nbcs, 1)*pdim)); nbcs = 0
    do i = 1, d%nbcs
      if (g2l(d%bcnode((i - 1)*(1 + pdim) + 1) + 1) < 0) cycle
      nbcs = nbcs + 1
      bcnode((nbcs - 1)*(1 + pdim) + 1:nbcs*(1 + pdim)) = d%bcnode((i - 1)*(1 + pdim) + 1:i*(1 + pdim))
      bcnode((nbcs - 1)*(1 + pdim) + 1) = g2l(d%bcnode((i - 1)*(1 + pdim) + 1) + 1)
      bcval((nbcs - 1)*pdim + 1:nbcs*pdim) = d%bcval((i - 1)*pdim + 1:i*pdim)
    end do
    allocate(fnode(max(d%nfrcs, 1)), fval(max(d%nfrcs, 1)*(pdim + 2))); nfrcs = 0
    do i = 1, d%nfrcs
      if (g2l(d%fnode(i) + 1) < 0) cycle
      nfrcs = nfrcs + 1
      fnode(nfrcs) = g2l(d%fnode(i) + 1)
      fval((nfrcs - 1)*(pdim + 2) + 1:nfrcs*(pdim + 2)) = d%fval((i - 1)*(pdim + 2) + 1:i*(pdim + 2))
    end do
    allocate(telsd(max(d%ntrcs, 1)*2), tval(max(d%ntrcs, 1)*(pdim + 2))); ntrcs = 0
    do i = 1, d%ntrcs
      if (e2l(d%telsd(2*i - 1) + 1) < 0) cycle
      ntrcs = ntrcs + 1
      telsd(2*ntrcs - 1) = e2l(d%telsd(2*i - 1) + 1); telsd(2*ntrcs) = d%telsd(2*i)
      tval((ntrcs - 1)*(pdim + 2) + 1:ntrcs*(pdim + 2)) = d%tval((i - 1)*(pdim + 2) + 1:i*(pdim + 2))
    end do

    ! ---- device objects (PetscInitialize, VecCreateMPI, MatCreateAIJ: src/main.F90:26,138,154) ----
    call fem_check(fem_mesh_create(ctx, pdim, nels, nnds, eltype, eptr, conn, mat_id, coords, d%nmts, d%mats, mesh), &
                   "fem_mesh_create")
    if (nranks > 1) then
      ! interface lists: for every other rank the nodes both hold, ascending global id on both sides
      allocate(theirs(d%nnds), neigh_rank(nranks), neigh_ptr(nranks + 1), neigh_nodes(nnds*max(nranks - 1, 1)), cnt(1))
      nneigh = 0; neigh_ptr(1) = 0
      do q = 0, nranks - 1
        if (q == rank) cycle
        theirs = .false.
        do e = 1, d%nels
          if (epart(e) /= q) cycle
          do a = d%eptr(e) + 1, d%eptr(e + 1)
            theirs(d%conn(a) + 1) = .true.
          end do
        end do
        k = neigh_ptr(nneigh + 1)
        do i = 1, d%nnds
          if (mine(i) .and. theirs(i)) then
            k = k + 1; neigh_nodes(k) = g2l(i)
          end if
        end do
        if (k > neigh_ptr(nneigh + 1)) then
          nneigh = nneigh + 1; neigh_rank(nneigh) = q; neigh_ptr(nneigh + 1) = k
        end if
      end do
      call fem_check(fem_mesh_set_halo(mesh, comm, nneigh, neigh_rank, neigh_ptr, neigh_nodes), "fem_mesh_set_halo")
    end if
    n = pdim*nnds
    call fem_check(fem_vec_create(ctx, int(n, c_long_long), Vec_U), "fem_vec_create")
    call fem_check(fem_vec_create(ctx, int(n, c_long_long), Vec_F), "fem_vec_create")
    allocate(uu(n), aggregate_u(n), stress_node(nnds*cpdim), aggregate_stress(nnds*cpdim))
    aggregate_u = 0; aggregate_stress = 0

    if (root) print *, "Forming [K] ..."                                   ! src/main.F90:151-171
    call fem_check(fem_mat_create_aij(mesh, Mat_K), "fem_mat_create_aij")
    call fem_check(fem_mat_assemble_elements(Mat_K), "fem_mat_assemble_elements")
    call fem_check(fem_mat_apply_kbc(Mat_K, nbcs, bcnode), "fem_mat_apply_kbc")
    call fem_check(fem_mat_get_info(Mat_K, nrows, nnz, dbytes), "fem_mat_get_info")
    if (root) write(*, '(A,I0,A,I0,A)') "  [K]: ", nrows, " rows, ", nnz, " nonzeros"

    if (root) print *, "Setting up environment ..."                        ! SetupKSPSolver, src/main.F90:339-403
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
      t_init = d%dt*(dtNo - 1)
      if (t_init >= d%t) exit
      if (root) write(*, '(A,I0)') " Interval: ", dtNo
      if (root) print *, "    Forming RHS ..."
      call fem_check(fem_form_rhs(mesh, nbcs, bcnode, bcval, nfrcs, fnode, fval, ntrcs, telsd, tval, d%t, t_init, d%dt, &
                     Vec_F), "fem_form_rhs")
      if (root) print *, "    Solving ..."
      call fem_check(fem_ksp_solve(Krylov, Vec_F, Vec_U), "fem_ksp_solve")
      call fem_check(fem_ksp_get_iteration_number(Krylov, its), "its")
      call fem_check(fem_ksp_get_converged_reason(Krylov, reason), "reason")
      call fem_check(fem_ksp_get_residual_norm(Krylov, rnorm), "rnorm")
      if (root) write(*, '(A,I0,A,I0,A,ES10.3)') "     KSP iterations: ", its, "  reason: ", reason, "  |M^-1 r|: ", rnorm
      if (reason < 0) error stop "KSP diverged"
      call fem_check(fem_vec_get_array(Vec_U, uu), "fem_vec_get_array")    ! GetVec_U, src/main.F90:250
      if (vec_view .and. root) then
        do i = 1, n
          write(*, '(ES23.15)') uu(i)
        end do
      end if
      aggregate_u = aggregate_u + uu
      if (root) print *, "    Getting stress ..."
      call fem_check(fem_recover_nodal_stress(mesh, Vec_U, stress_node), "fem_recover_nodal_stress")
      aggregate_stress = aggregate_stress + stress_node
      dtNo = dtNo + 1
    end do

    if (resolve_rtol > 0.0_c_double) then
      call fem_check(fem_ksp_get_reused_solves(Krylov, i), "fem_ksp_get_reused_solves")
      call fem_check(fem_ksp_set_tolerances(Krylov, resolve_rtol, FEM_DEFAULT_REAL, FEM_DEFAULT_REAL, FEM_DEFAULT_INT), &
                     "fem_ksp_set_tolerances")
      call fem_check(fem_ksp_solve(Krylov, Vec_F, Vec_U), "fem_ksp_solve")
      call fem_check(fem_ksp_get_iteration_number(Krylov, j), "its")
      call fem_check(fem_ksp_get_converged_reason(Krylov, reason), "reason")
      call fem_check(fem_ksp_get_reused_solves(Krylov, k), "fem_ksp_get_reused_solves")
      if (root) write(*, '(A,ES10.3,A,I0,A,I0,A,I0,A,I0,A,I0)') " Re-solve at rtol ", resolve_rtol, ": iterations ", j, &
        "  (last increment ", its, ")  reason ", reason, "  reused before ", i, " after ", k
      call fem_check(fem_vec_get_array(Vec_U, uu), "fem_vec_get_array")
      if (root) write(*, '(A,ES23.15)') " Re-solve max |du|: ", maxval(abs(uu))
    end if

    call write_output()

    if (root) print *, "Cleaning up ..."
    call fem_check(fem_ksp_destroy(Krylov), "fem_ksp_destroy")
    call fem_check(fem_vec_destroy(Vec_F), "fem_vec_destroy")
    call fem_check(fem_vec_destroy(Vec_U), "fem_vec_destroy")
    call fem_check(fem_mat_destroy(Mat_K), "fem_mat_destroy")
    call fem_check(fem_mesh_destroy(mesh), "fem_mesh_destroy")

  contains

    ! Legacy ASCII VTK in the reference's layout (WriteOutput, src/m_io.F90:480-555): POINTS F0.3, CELLS "n id0..",
    ! CELL_TYPES vtk ids, SCALARS STRESS FLOAT cpdim F0.6, VECTORS displacements F0.6 -- this rank's elements and
    ! nodes in local numbering; plus a full-precision companion file with the global node ids.
    subroutine write_output()
      character(300) :: name
      character(32) :: fmt
      integer :: ee, kk, ncol, vtk, dum1, dum2, dum3, dum4, nne, u1, u2, jj
      real(8) :: p(3)
      u1 = 20 + 2*rank; u2 = u1 + 1
      if (nranks == 1) then
        name = trim(prefix)//"_000000.vtk"
      else
        write(name, '(A,I0,A)') trim(outdir), rank, "_output_000000.vtk"   ! src/m_io.F90:497
      end if
      open(u1, file=trim(name), status='replace')
      write(u1, '(A)') "# vtk DataFile Version 2.0"
      write(u1, '(A)') "File written by Defmod"
      write(u1, '(A)') "ASCII"
      write(u1, '(A)') "DATASET UNSTRUCTURED_GRID"
      write(u1, '(A,I0,A)') "POINTS ", nnds, " double"
      do kk = 1, nnds
        p = 0
        p(1:pdim) = coords((kk - 1)*pdim + 1:kk*pdim)
        write(u1, '(3(F0.3,1X))') p
      end do
      ncol = 0
      do ee = 1, nels
        ncol = ncol + (eptr(ee + 1) - eptr(ee)) + 1
      end do
      write(u1, '(A,I0,1X,I0)') "CELLS ", nels, ncol
      do ee = 1, nels
        nne = eptr(ee + 1) - eptr(ee)
        write(fmt, '(A,I0,A)') "(I0,", nne, "(1X,I0))"
        write(u1, fmt) nne, conn(eptr(ee) + 1:eptr(ee + 1))
      end do
      write(u1, '(A,I0)') "CELL_TYPES ", nels
      do ee = 1, nels
        call fem_check(fem_el_counts(eltype(ee), dum1, dum2, dum3, dum4, vtk), "fem_el_counts")
        write(u1, '(I0)') vtk
      end do
      write(u1, '(A,I0)') "POINT_DATA ", nnds
      write(u1, '(A,I0)') "SCALARS STRESS FLOAT ", cpdim
      write(u1, '(A)') "LOOKUP_TABLE DEFAULT"
      write(fmt, '(A,I0,A)') "(", cpdim, "(F0.6,1X))"
      do kk = 1, nnds
        write(u1, fmt) (aggregate_stress(kk + (jj - 1)*nnds), jj=1, cpdim)
      end do
      write(u1, '(A)') "VECTORS displacements double"
      do kk = 1, nnds
        p = 0
        p(1:pdim) = aggregate_u((kk - 1)*pdim + 1:kk*pdim)
        write(u1, '(3(F0.6,1X))') p
      end do
      close(u1)
      ! full precision (the 6-digit VTK cannot pin 1e-10 parity)
      name = name(1:len_trim(name) - 4)//".full.txt"
      open(u2, file=trim(name), status='replace')
      write(u2, '(I0,1X,I0,1X,I0)') nnds, pdim, cpdim
      do kk = 1, nnds
        if (nranks == 1) then
          write(u2, '(9(ES24.16,1X))') aggregate_u((kk - 1)*pdim + 1:kk*pdim), &
                                        (aggregate_stress(kk + (jj - 1)*nnds), jj=1, cpdim)
        else
          write(u2, '(I0,1X,9(ES24.16,1X))') nl2g(kk), aggregate_u((kk - 1)*pdim + 1:kk*pdim), &
                                              (aggregate_stress(kk + (jj - 1)*nnds), jj=1, cpdim)
        end if
      end do
      close(u2)
    end subroutine write_output

  end subroutine run_domain

end module defmod_deck

program main_amd
  use iso_c_binding
  use fem_amd_capi
  use defmod_deck
  use omp_lib
  implicit none

  character(256) :: input_file, arg, prefix, outdir, uidfile
  character(1024) :: options
  character(32) :: ebuf
  type(deck_t) :: deck
  integer, allocatable :: epart(:)
  type(c_ptr), allocatable :: ctxs(:), comms(:)
  type(c_ptr) :: ctx, comm
  character(kind=c_char) :: uid(128)
  integer :: i, nargs, np_local, rank, nranks, device, stat, pos, tries
  logical :: vec_view, have

  input_file = ""; options = ""; prefix = "0_output"; vec_view = .false.; np_local = 1
  nargs = command_argument_count()
  i = 1
  do while (i <= nargs)
    call get_command_argument(i, arg)
    if (trim(arg) == "-f" .and. i < nargs) then
      call get_command_argument(i + 1, input_file); i = i + 2
    else if (trim(arg) == "-o" .and. i < nargs) then
      call get_command_argument(i + 1, prefix); i = i + 2
    else if (trim(arg) == "-np" .and. i < nargs) then
      call get_command_argument(i + 1, arg); read(arg, *) np_local; i = i + 2
    else if (trim(arg) == "-vec_view") then
      vec_view = .true.; i = i + 1
    else if (trim(arg) == "-resolve_rtol" .and. i < nargs) then
      call get_command_argument(i + 1, arg); read(arg, *) resolve_rtol; i = i + 2
    else
      options = trim(options)//" "//trim(arg); i = i + 1
    end if
  end do
  if (len_trim(input_file) == 0) then
    print *, "Usage: defmod_amd -f <filename> [-np <domains>] [-ksp_rtol <tol> ...]"   ! src/main.F90:32
    stop
  end if
  pos = index(prefix, "/", back=.true.)
  outdir = ""; if (pos > 0) outdir = prefix(1:pos)

  ! rank and size of a one-process-per-GPU launch: from the environment, before anything touches the GPU
  rank = 0; nranks = 1
  call env_int("FEM_AMD_RANK", rank, have); if (.not. have) call env_int("RANK", rank, have)
  call env_int("FEM_AMD_NRANKS", nranks, have); if (.not. have) call env_int("WORLD_SIZE", nranks, have)
  device = 0
  call env_int("FEM_AMD_DEVICE", device, have); if (.not. have) call env_int("LOCAL_RANK", device, have)
  if (np_local > 1 .and. nranks > 1) error stop "-np and a multi-process launch exclude each other"

  if (rank == 0) print *, "Reading input ..."
  call read_deck(input_file, deck)

  if (np_local > 1) then
    ! ---- N domains in this process: one host thread and one context each (MPI_Comm_Size = np_local) ----
    print *, "Partitioning mesh ..."
    call partition_elements(deck, np_local, epart)
    allocate(ctxs(np_local), comms(np_local))
    do i = 1, np_local
      call fem_check(fem_ctx_create(int(device, c_int), ctxs(i)), "fem_ctx_create")
    end do
    call fem_check(fem_comm_create_local_group(int(np_local, c_int), ctxs, comms), "fem_comm_create_local_group")
    !$omp parallel num_threads(np_local) default(shared)
    call run_domain(deck, epart, omp_get_thread_num(), np_local, ctxs(omp_get_thread_num() + 1), &
                    comms(omp_get_thread_num() + 1), options, outdir, prefix, vec_view)
    !$omp end parallel
    do i = 1, np_local
      call fem_check(fem_comm_destroy(comms(i)), "fem_comm_destroy")
      call fem_check(fem_ctx_destroy(ctxs(i)), "fem_ctx_destroy")
    end do
  else
    call fem_check(fem_ctx_create(int(device, c_int), ctx), "fem_ctx_create")
    comm = c_null_ptr
    if (nranks > 1) then
      ! ---- one process per GPU: RCCL communicator, its id handed over through a file ----
      if (rank == 0) print *, "Partitioning mesh ..."
      call partition_elements(deck, nranks, epart)
      call get_environment_variable("FEM_AMD_UID_FILE", uidfile, status=stat)
      if (stat /= 0) error stop "multi-process launch: FEM_AMD_UID_FILE is not set"
      if (rank == 0) then
        call fem_check(fem_comm_get_unique_id(uid), "fem_comm_get_unique_id")
        open(30, file=trim(uidfile)//".tmp", status='replace', access='stream', form='unformatted')
        write(30) uid
        close(30)
        call rename(trim(uidfile)//".tmp", trim(uidfile))
      else
        tries = 0
        do
          inquire(file=trim(uidfile), exist=have)
          if (have) exit
          call sleep(1); tries = tries + 1
          if (tries > 120) error stop "multi-process launch: no communicator id after 120 s"
        end do
        open(30, file=trim(uidfile), status='old', access='stream', form='unformatted')
        read(30) uid
        close(30)
      end if
      call fem_check(fem_comm_create_rccl(ctx, int(nranks, c_int), int(rank, c_int), uid, comm), "fem_comm_create_rccl")
    else
      allocate(epart(deck%nels)); epart = 0
    end if
    call run_domain(deck, epart, rank, nranks, ctx, comm, options, outdir, prefix, vec_view)
    if (nranks > 1) call fem_check(fem_comm_destroy(comm), "fem_comm_destroy")
    call fem_check(fem_ctx_destroy(ctx), "fem_ctx_destroy")
  end if
  if (rank == 0) print *, "Finished"

contains

  subroutine env_int(name, val, found)
    character(*), intent(in) :: name
    integer, intent(inout) :: val
    logical, intent(out) :: found
    integer :: st, v
    call get_environment_variable(name, ebuf, status=st)
    found = .false.
    if (st /= 0) return
    read(ebuf, *, iostat=st) v
    if (st == 0) then
      val = v; found = .true.
    end if
  end subroutine env_int

end program main_amd
