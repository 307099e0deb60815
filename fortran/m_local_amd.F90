! fortran/m_local_amd.F90 -- the element-level Fortran API of the reference's
! modules `elems` / `local` (src/m_elems.F90, src/m_local.F90), served by the
! HIP library.  Procedure names, argument order, array shapes (column-major,
! ecoords(nodecount,pdim), k(n*pdim,n*pdim), estress(nip,cpdim)) and the
! 1-based node numbering of FormElIndx are the reference's, so host code and
! per-element parity tests can switch with
!     use local_amd      ! instead of: use local
! A GPU context must exist first: call ElemsAmdInit(device) once (it stands
! where `call Initialize(ipoints, weights)` stands in src/main.F90:41).
module local_amd
  use iso_c_binding
  use fem_amd_capi
  implicit none
  private
  public :: pdim, cpdim, ElemsAmdInit, ElemsAmdFinalize, elems_ctx
  public :: getNodeCount, getNip, getDim, getNps, getVtkid, getElTypeNo
  public :: ElTables, FormElKE, FormElIndx, CalcElStress

  integer :: pdim = 2, cpdim = 3          ! module state of `elems` (src/m_elems.F90:20)
  type(c_ptr) :: elems_ctx = c_null_ptr

contains

  subroutine ElemsAmdInit(device)
    integer, intent(in) :: device
    if (pdim == 2) cpdim = 3              ! Initialize, src/m_elems.F90:128-129
    if (pdim == 3) cpdim = 6
    if (.not. c_associated(elems_ctx)) call fem_check(fem_ctx_create(int(device, c_int), elems_ctx), "fem_ctx_create")
  end subroutine ElemsAmdInit

  subroutine ElemsAmdFinalize()
    if (c_associated(elems_ctx)) call fem_check(fem_ctx_destroy(elems_ctx), "fem_ctx_destroy")
    elems_ctx = c_null_ptr
  end subroutine ElemsAmdFinalize

  integer function getElTypeNo(eltype)    ! src/m_elems.F90:63-73
    character(*), intent(in) :: eltype
    getElTypeNo = fem_type_no(eltype)
  end function getElTypeNo

  subroutine counts(eltype, n, np, d, s, v)
    character(*), intent(in) :: eltype
    integer(c_int), intent(out) :: n, np, d, s, v
    call fem_check(fem_el_counts(fem_type_no(eltype), n, np, d, s, v), "fem_el_counts("//trim(eltype)//")")
  end subroutine counts

  integer function getNodeCount(eltype)   ! src/m_elems.F90:83
    character(*), intent(in) :: eltype
    integer(c_int) :: n, np, d, s, v
    call counts(eltype, n, np, d, s, v); getNodeCount = n
  end function getNodeCount

  integer function getNip(eltype)         ! src/m_elems.F90:92
    character(*), intent(in) :: eltype
    integer(c_int) :: n, np, d, s, v
    call counts(eltype, n, np, d, s, v); getNip = np
  end function getNip

  integer function getDim(eltype)         ! src/m_elems.F90:49
    character(*), intent(in) :: eltype
    integer(c_int) :: n, np, d, s, v
    call counts(eltype, n, np, d, s, v); getDim = d
  end function getDim

  integer function getNps(eltype)         ! src/m_elems.F90:40
    character(*), intent(in) :: eltype
    integer(c_int) :: n, np, d, s, v
    call counts(eltype, n, np, d, s, v); getNps = s
  end function getNps

  integer function getVtkid(eltype)       ! src/m_elems.F90:107
    character(*), intent(in) :: eltype
    integer(c_int) :: n, np, d, s, v
    call counts(eltype, n, np, d, s, v); getVtkid = v
  end function getVtkid

  ! The tables Initialize builds (src/m_elems.F90:124-133): ipoints(nip,pdim),
  ! weights(nip), N(nip,nodecount), N2Inv(nip,nodecount); has_inv false where
  ! NodalStressInv skips the type.
  subroutine ElTables(eltype, ipoints, weights, N, N2Inv, has_inv)
    character(*), intent(in) :: eltype
    real(8), intent(out) :: ipoints(:, :), weights(:), N(:, :), N2Inv(:, :)
    logical, intent(out) :: has_inv
    integer(c_int) :: nn, np, d, s, v, hi
    real(c_double), allocatable :: a(:), w(:), sh(:), inv(:)
    call counts(eltype, nn, np, d, s, v)
    allocate(a(np*d), w(np), sh(np*nn), inv(np*nn))
    inv = 0.0d0
    call fem_check(fem_el_tables(fem_type_no(eltype), a, w, sh, inv, hi), "fem_el_tables")
    ipoints(1:np, 1:d) = reshape(a, (/np, d/))
    weights(1:np) = w
    N(1:np, 1:nn) = reshape(sh, (/np, nn/))
    N2Inv(1:np, 1:nn) = reshape(inv, (/np, nn/))
    has_inv = hi /= 0
  end subroutine ElTables

  ! FormElKE, src/m_local.F90:21-67 (dt is unused there as well)
  subroutine FormElKE(eltype, ecoords, E, nu, dt, k)
    character(*), intent(in) :: eltype
    real(8), intent(in) :: ecoords(:, :), E, nu, dt
    real(8), intent(out) :: k(:, :)
    real(c_double), allocatable :: ec(:), kk(:)
    real(c_double) :: Ea(1), na(1)
    integer :: n, nd
    n = size(ecoords, 1); nd = n*size(ecoords, 2)
    allocate(ec(nd), kk(nd*nd))
    ec = reshape(ecoords, (/nd/))
    Ea(1) = E; na(1) = nu
    call fem_check(fem_form_el_ke(elems_ctx, fem_type_no(eltype), 1_c_int, ec, Ea, na, kk), "fem_form_el_ke")
    k(1:nd, 1:nd) = reshape(kk, (/nd, nd/))
  end subroutine FormElKE

  ! FormElIndx, src/m_local.F90:70-78 (1-based node ids in, 1-based dof ids out)
  subroutine FormElIndx(enodes, indx)
    integer, intent(in) :: enodes(:)
    integer, intent(out) :: indx(:)
    integer(c_int), allocatable :: en(:), ix(:)
    integer :: n
    n = size(enodes)
    allocate(en(n), ix(n*pdim))
    en = enodes - 1
    call fem_check(fem_form_el_indx(int(pdim, c_int), int(n, c_int), en, ix), "fem_form_el_indx")
    indx(1:n*pdim) = ix + 1
  end subroutine FormElIndx

  ! CalcElStress, src/m_local.F90:93-123
  subroutine CalcElStress(eltype, ecoords, edisp, E, nu, estress)
    character(*), intent(in) :: eltype
    real(8), intent(in) :: ecoords(:, :), edisp(:), E, nu
    real(8), intent(out) :: estress(:, :)
    real(c_double), allocatable :: ec(:), ed(:), es(:)
    real(c_double) :: Ea(1), na(1)
    integer :: n, nd, np, cp
    n = size(ecoords, 1); nd = n*size(ecoords, 2)
    np = getNip(eltype); cp = 3; if (size(ecoords, 2) == 3) cp = 6
    allocate(ec(nd), ed(nd), es(np*cp))
    ec = reshape(ecoords, (/nd/)); ed = edisp(1:nd)
    Ea(1) = E; na(1) = nu
    call fem_check(fem_calc_el_stress(elems_ctx, fem_type_no(eltype), 1_c_int, ec, ed, Ea, na, es), "fem_calc_el_stress")
    estress(1:np, 1:cp) = reshape(es, (/np, cp/))
  end subroutine CalcElStress

end module local_amd
