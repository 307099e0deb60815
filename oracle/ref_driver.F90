! oracle/ref_driver.F90 -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
!
! Thin bind(C) entry points around the *reference's own* element modules
! (`utils`, `elems`, `local`, `seplaw`; /root/reference/src/m_utils.F90,
! m_elems.F90, m_local.F90, m_seplaw.F90), which are compiled from where they
! lie by oracle/Makefile into oracle/_ref/libdefmod_ref.so.  Nothing in this
! file restates reference arithmetic: every number returned is produced by the
! reference routines named in each wrapper.  It exists so that
!   * the C restatement in oracle/defmod_oracle.c can be validated against the
!     real reference element maths, and
!   * tests/golden/gen_golden.py can dump golden vectors.
! Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
! load the resulting library.
!
! Element type numbers follow elTypes in m_elems.F90:26 (1 tri, 2 qua, 3 tet,
! 4 hex, 5 coh).  All arrays are column-major exactly as the reference holds
! them (ecoords(nodecount,pdim), k(n*pdim,n*pdim), estress(nip,cpdim)).

module ref_driver
  use iso_c_binding
  use local          ! brings elems + utils, ipoints(:), weights(:)
  use seplaw
  implicit none
contains

  function tname(t) result(s)
    integer(c_int), value :: t
    character(3) :: s
    s = elTypes(t)(1:3)
  end function tname

  ! pdim is a module global that the caller must set before Initialize
  ! (main.F90:40-41); Initialize = m_elems.F90:124.
  subroutine ref_init(pd) bind(C, name="ref_init")
    integer(c_int), value :: pd
    pdim = pd
    call Initialize(ipoints, weights)
  end subroutine ref_init

  subroutine ref_counts(t, nodecount, nipo, dimo, npso, vtk) bind(C, name="ref_counts")
    integer(c_int), value :: t
    integer(c_int), intent(out) :: nodecount, nipo, dimo, npso, vtk
    nodecount = getNodeCount(tname(t))
    nipo = getNip(tname(t))
    dimo = getDim(tname(t))
    npso = getNps(tname(t))
    vtk = getVtkid(tname(t))
  end subroutine ref_counts

  ! ipoints/weights (m_local.F90:14-15), shapeFuncMem, nodalStressInvMem
  ! (m_elems.F90:32-35).  has_inv = 0 when NodalStressInv skipped the type.
  subroutine ref_tables(t, ip, w, shp, n2inv, has_inv) bind(C, name="ref_tables")
    integer(c_int), value :: t
    real(c_double), intent(out) :: ip(*), w(*), shp(*), n2inv(*)
    integer(c_int), intent(out) :: has_inv
    integer :: n, np, i, j
    n = getNodeCount(tname(t)); np = getNip(tname(t))
    do j = 1, pdim
      do i = 1, np
        ip(i + (j-1)*np) = ipoints(t)%array(i, j)
      end do
    end do
    do i = 1, np
      w(i) = weights(t)%array(i)
      do j = 1, n
        shp(i + (j-1)*np) = shapeFuncMem(t)%array(i)%array(j)   ! shp(gp,node)
      end do
    end do
    has_inv = 0
    if (allocated(nodalStressInvMem(t)%array)) then
      if (getDim(tname(t)) == pdim .and. np == n) then
        has_inv = 1
        do j = 1, n
          do i = 1, np
            n2inv(i + (j-1)*np) = nodalStressInvMem(t)%array(i, j)
          end do
        end do
      end if
    end if
  end subroutine ref_tables

  ! FormElKE, m_local.F90:21-67
  subroutine ref_form_el_ke(t, ecoords, E, nu, k) bind(C, name="ref_form_el_ke")
    integer(c_int), value :: t
    real(c_double), value :: E, nu
    real(c_double), intent(in) :: ecoords(*)
    real(c_double), intent(out) :: k(*)
    integer :: n, nd, i, j
    real(8), allocatable :: ec(:,:), kk(:,:)
    real(8) :: dtl
    n = getNodeCount(tname(t)); nd = n*pdim
    allocate(ec(n, pdim), kk(nd, nd))
    ec = reshape(ecoords(1:n*pdim), (/n, pdim/))
    dtl = 1.0d0
    call FormElKE(tname(t), ec, E, nu, dtl, kk)
    do j = 1, nd
      do i = 1, nd
        k(i + (j-1)*nd) = kk(i, j)
      end do
    end do
    deallocate(ec, kk)
  end subroutine ref_form_el_ke

  ! CalcElStress, m_local.F90:93-123
  subroutine ref_calc_el_stress(t, ecoords, edisp, E, nu, estress) bind(C, name="ref_calc_el_stress")
    integer(c_int), value :: t
    real(c_double), value :: E, nu
    real(c_double), intent(in) :: ecoords(*), edisp(*)
    real(c_double), intent(out) :: estress(*)
    integer :: n, np, i, j
    real(8), allocatable :: ec(:,:), ed(:), es(:,:)
    n = getNodeCount(tname(t)); np = getNip(tname(t))
    allocate(ec(n, pdim), ed(n*pdim), es(np, cpdim))
    ec = reshape(ecoords(1:n*pdim), (/n, pdim/))
    ed = edisp(1:n*pdim)
    call CalcElStress(tname(t), ec, ed, E, nu, es)
    do j = 1, cpdim
      do i = 1, np
        estress(i + (j-1)*np) = es(i, j)
      end do
    end do
    deallocate(ec, ed, es)
  end subroutine ref_calc_el_stress

  ! FormdNdetJ, m_local.F90:175-189
  subroutine ref_form_dn_detj(t, ipcoord, ecoords, dN, detj) bind(C, name="ref_form_dn_detj")
    integer(c_int), value :: t
    real(c_double), intent(in) :: ipcoord(*), ecoords(*)
    real(c_double), intent(out) :: dN(*), detj
    integer :: n, i, j
    real(8), allocatable :: ec(:,:), d(:,:)
    real(8) :: ipc(pdim)
    n = getNodeCount(tname(t))
    allocate(ec(n, pdim), d(pdim, n))
    ec = reshape(ecoords(1:n*pdim), (/n, pdim/))
    ipc = ipcoord(1:pdim)
    call FormdNdetJ(tname(t), ipc, ec, d, detj)
    do j = 1, n
      do i = 1, pdim
        dN(i + (j-1)*pdim) = d(i, j)
      end do
    end do
    deallocate(ec, d)
  end subroutine ref_form_dn_detj

  ! DMat, m_local.F90:204-228
  subroutine ref_dmat(E, nu, D) bind(C, name="ref_dmat")
    real(c_double), value :: E, nu
    real(c_double), intent(out) :: D(*)
    real(8) :: DD(cpdim, cpdim)
    integer :: i, j
    call DMat(DD, E, nu)
    do j = 1, cpdim
      do i = 1, cpdim
        D(i + (j-1)*cpdim) = DD(i, j)
      end do
    end do
  end subroutine ref_dmat

  ! FormElIndx, m_local.F90:70-78
  subroutine ref_form_el_indx(n, enodes, indx) bind(C, name="ref_form_el_indx")
    integer(c_int), value :: n
    integer(c_int), intent(in) :: enodes(n)
    integer(c_int), intent(out) :: indx(n*pdim)
    integer :: en(n), ix(n*pdim)
    en = enodes
    call FormElIndx(en, ix)
    indx = ix
  end subroutine ref_form_el_indx

  ! EdgeAreaNodes, m_elems.F90:188-196 (+ per-type routines)
  subroutine ref_edge_area_nodes(t, enodes, ecoords, side, area, snodes) bind(C, name="ref_edge_area_nodes")
    integer(c_int), value :: t, side
    integer(c_int), intent(in) :: enodes(*)
    real(c_double), intent(in) :: ecoords(*)
    real(c_double), intent(out) :: area
    integer(c_int), intent(out) :: snodes(*)
    integer :: n, nps_l
    integer, allocatable :: en(:), sn(:)
    real(8), allocatable :: ec(:,:)
    n = getNodeCount(tname(t)); nps_l = getNps(tname(t))
    allocate(en(n), sn(nps_l), ec(n, pdim))
    en = enodes(1:n)
    ec = reshape(ecoords(1:n*pdim), (/n, pdim/))
    call EdgeAreaNodes(tname(t), en, ec, side, area, sn)
    snodes(1:nps_l) = sn
    deallocate(en, sn, ec)
  end subroutine ref_edge_area_nodes

  ! Seplaw_1_Tract / Seplaw_1_Stiff, m_seplaw.F90:15,57
  subroutine ref_seplaw1(props, gap, vgap, dtime, tract, stiff) bind(C, name="ref_seplaw1")
    real(c_double), intent(in) :: props(6), gap(2), vgap(2)
    real(c_double), value :: dtime
    real(c_double), intent(out) :: tract(2), stiff(4)
    real(8) :: p(6), st(2,2)
    p = props
    call Seplaw_1_Tract(p, gap, vgap, dtime, tract)
    call Seplaw_1_Stiff(p, gap, vgap, dtime, st)
    stiff(1) = st(1,1); stiff(2) = st(2,1); stiff(3) = st(1,2); stiff(4) = st(2,2)
  end subroutine ref_seplaw1

  ! getCohValues, m_elems.F90:658-675
  subroutine ref_coh_values(ecoords, tangent, normal, det) bind(C, name="ref_coh_values")
    real(c_double), intent(in) :: ecoords(8)
    real(c_double), intent(out) :: tangent(2), normal(2), det
    real(8) :: ec(4,2)
    ec = reshape(ecoords, (/4, 2/))
    call getCohValues(ec, tangent, normal, det)
  end subroutine ref_coh_values

  ! getCohRels, m_elems.F90:676-706.  The routine ignores `nodes` and reads local_u(2*(k-1)+i), k = 1..4:
  ! the wrapper hands it exactly those eight numbers.
  subroutine ref_coh_rels(local_u, nip, dt, urel, vrel) bind(C, name="ref_coh_rels")
    real(c_double), intent(in) :: local_u(8)
    integer(c_int), value :: nip
    real(c_double), value :: dt
    real(c_double), intent(out) :: urel(2), vrel(2)
    integer :: nodes(4)
    real(8) :: lu(8)
    nodes = (/1, 2, 3, 4/)
    lu = local_u
    call getCohRels(nodes, lu, int(nip), dt, urel, vrel)
  end subroutine ref_coh_rels

  ! getCohGaps, m_elems.F90:709-720
  subroutine ref_coh_gaps(tangent, normal, urel, vrel, gap, vgap) bind(C, name="ref_coh_gaps")
    real(c_double), intent(in) :: tangent(2), normal(2), urel(2), vrel(2)
    real(c_double), intent(out) :: gap(2), vgap(2)
    call getCohGaps(tangent, normal, urel, vrel, gap, vgap)
  end subroutine ref_coh_gaps

end module ref_driver
