! fortran/m_local_amd.F90 -- the element-level Fortran API of the reference's
! modules `elems` / `local` (src/m_elems.F90, src/m_local.F90), served by the
! HIP library.  Procedure names, argument order, array shapes (column-major,
! ecoords(nodecount,pdim), k(n*pdim,n*pdim), estress(nip,cpdim)) and the
! 1-based node numbering of FormElIndx are the reference's, so host code and
! per-element parity tests can switch with
!     use local_amd      ! instead of: use local
! or, compiled with -DFEM_AMD_AS_LOCAL, the module IS called `local` and a reference-side file that says
! `use local` builds against it unchanged.  As in the reference the caller sets `pdim` and then calls
! Initialize(ipoints, weights) (src/main.F90:40-41), which here also brings up the GPU context (device
! FEM_AMD_DEVICE, default 0); ElemsAmdInit(device) does the same without the tables.
#ifdef FEM_AMD_AS_LOCAL
#define LOCAL_MODULE_NAME local
#else
#define LOCAL_MODULE_NAME local_amd
#endif
module LOCAL_MODULE_NAME
  use iso_c_binding
  use fem_amd_capi
  implicit none
  private
  public :: pdim, cpdim, ElemsAmdInit, ElemsAmdFinalize, elems_ctx
  public :: getNodeCount, getNip, getDim, getNps, getVtkid, getElTypeNo
  public :: ElTables, FormElKE, FormElIndx, CalcElStress
  public :: array1d, array2d, array1dWrapper, elTypeCount, ipoints, weights, shapeFuncMem, nodalStressInvMem
  public :: Initialize, ShapeFunc, ShapeFuncd, FormdNdetJ, BMat, DMat

  ! the ragged-array types of module `utils` (src/m_utils.F90:6-20) that Initialize's arguments use
  type array1d
    real(8), allocatable :: array(:)
  end type
  type array2d
    real(8), allocatable :: array(:, :)
  end type array2d
  type array1dWrapper
    type(array1d), allocatable :: array(:)
  end type array1dWrapper

  integer, parameter :: elTypeCount = 5   ! src/m_elems.F90:25
  integer :: pdim = 2, cpdim = 3          ! module state of `elems` (src/m_elems.F90:20)
  type(c_ptr) :: elems_ctx = c_null_ptr
  ! module storage of `local` / `elems` (src/m_local.F90:14-15, src/m_elems.F90:32-35)
  type(array2d), target :: ipoints(elTypeCount)
  type(array1d), target :: weights(elTypeCount)
  type(array1dWrapper), target :: shapeFuncMem(elTypeCount)
  type(array2d) :: nodalStressInvMem(elTypeCount)
  character(4), parameter :: elTypeNames(elTypeCount) = (/"tri ", "qua ", "tet ", "hex ", "coh "/)

contains

  ! Initialize (src/m_elems.F90:124-133): cpdim, then for every element type of the current pdim the Gauss points
  ! and weights into the caller's arrays, the shape functions at the Gauss points and the extrapolation inverse
  ! into the module's memo arrays -- all read from the library's element tables (the ones its kernels use).
  subroutine Initialize(ipts, wts)
    type(array2d) :: ipts(:)
    type(array1d) :: wts(:)
    integer :: t, g, dev, stat
    integer(c_int) :: nn, np, d, s, v, hi
    real(c_double), allocatable :: a(:), w(:), sh(:), inv(:)
    character(16) :: buf
    dev = 0
    call get_environment_variable("FEM_AMD_DEVICE", buf, status=stat)
    if (stat == 0) read(buf, *, iostat=stat) dev
    call ElemsAmdInit(dev)
    do t = 1, elTypeCount
      call fem_check(fem_el_counts(int(t, c_int), nn, np, d, s, v), "fem_el_counts")
      if (d /= pdim) cycle                 ! SamPts: tri, qua, coh in 2-D; tet, hex in 3-D (src/m_elems.F90:148-158)
      allocate(a(np*d), w(np), sh(np*nn), inv(np*nn))
      inv = 0.0d0
      call fem_check(fem_el_tables(int(t, c_int), a, w, sh, inv, hi), "fem_el_tables")
      if (allocated(ipts(t)%array)) deallocate(ipts(t)%array)
      if (allocated(wts(t)%array)) deallocate(wts(t)%array)
      allocate(ipts(t)%array(np, d), wts(t)%array(np))
      ipts(t)%array = reshape(a, (/np, d/)); wts(t)%array = w
      if (allocated(shapeFuncMem(t)%array)) deallocate(shapeFuncMem(t)%array)
      allocate(shapeFuncMem(t)%array(np))
      do g = 1, np
        allocate(shapeFuncMem(t)%array(g)%array(nn))
        shapeFuncMem(t)%array(g)%array = sh(g:np*nn:np)          ! shape(nip, n) column-major: row g
      end do
      if (allocated(nodalStressInvMem(t)%array)) deallocate(nodalStressInvMem(t)%array)
      if (hi /= 0) then                    ! NodalStressInv fills the types with nip == nodecount (src/m_elems.F90:735)
        allocate(nodalStressInvMem(t)%array(np, nn))
        nodalStressInvMem(t)%array = reshape(inv, (/np, nn/))
      end if
      deallocate(a, w, sh, inv)
    end do
  end subroutine Initialize

  ! ShapeFunc (src/m_elems.F90:136-141): a pointer into the module's memo storage
  subroutine ShapeFunc(eltype, nip, N)
    character(*), intent(in) :: eltype
    integer, intent(in) :: nip
    real(8), pointer, intent(out) :: N(:)
    N => shapeFuncMem(getElTypeNo(eltype))%array(nip)%array
  end subroutine ShapeFunc

  ! ShapeFuncd (src/m_elems.F90:177-185): dN(pdim, nodecount) at the point coord
  subroutine ShapeFuncd(eltype, dN, coord)
    character(*) :: eltype
    real(8) :: dN(:, :), coord(:)
    real(c_double), allocatable :: d(:)
    real(c_double) :: c(3)
    integer :: n, pd
    n = getNodeCount(eltype); pd = getDim(eltype)
    allocate(d(pd*n))
    c = 0.0d0; c(1:pd) = coord(1:pd)
    call fem_check(fem_shape_funcd(elems_ctx, fem_type_no(eltype), 1_c_int, c, d), "fem_shape_funcd")
    dN(1:pd, 1:n) = reshape(d, (/pd, n/))
  end subroutine ShapeFuncd

  ! FormdNdetJ (src/m_local.F90:175-189)
  subroutine FormdNdetJ(eltype, ipcoord, ecoords, dN, detj)
    character(*) :: eltype
    real(8) :: ipcoord(pdim), detj, ecoords(:, :), dN(:, :)
    real(c_double), allocatable :: ec(:), d(:)
    real(c_double) :: dj(1), c(3)
    integer :: n, pd
    n = size(ecoords, 1); pd = size(ecoords, 2)
    allocate(ec(n*pd), d(pd*n))
    ec = reshape(ecoords, (/n*pd/))
    c = 0.0d0; c(1:pd) = ipcoord(1:pd)
    call fem_check(fem_form_dn_detj(elems_ctx, fem_type_no(eltype), 1_c_int, c, ec, d, dj), "fem_form_dn_detj")
    dN(1:pd, 1:n) = reshape(d, (/pd, n/))
    detj = dj(1)
  end subroutine FormdNdetJ

  ! BMat (src/m_local.F90:148-171)
  subroutine BMat(nodecount, dN, B)
    integer :: nodecount
    real(8) :: dN(pdim, nodecount), B(cpdim, nodecount*pdim)
    real(c_double), allocatable :: d(:), bb(:)
    allocate(d(pdim*nodecount), bb(cpdim*nodecount*pdim))
    d = reshape(dN, (/pdim*nodecount/))
    call fem_check(fem_bmat(elems_ctx, int(pdim, c_int), int(nodecount, c_int), 1_c_int, d, bb), "fem_bmat")
    B = reshape(bb, (/cpdim, nodecount*pdim/))
  end subroutine BMat

  ! DMat (src/m_local.F90:204-208): plane strain for a 3x3 D, 3-D for a 6x6 D
  subroutine DMat(D, E, nu)
    real(8) :: D(:, :), E, nu
    real(c_double), allocatable :: dd(:)
    real(c_double) :: Ea(1), na(1)
    integer :: cp, pd
    cp = size(D, 1); pd = 2; if (cp == 6) pd = 3
    allocate(dd(cp*cp))
    Ea(1) = E; na(1) = nu
    call fem_check(fem_dmat(elems_ctx, int(pd, c_int), 1_c_int, Ea, na, dd), "fem_dmat")
    D(1:cp, 1:cp) = reshape(dd, (/cp, cp/))
  end subroutine DMat

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

end module LOCAL_MODULE_NAME
