! fortran/test_elems_amd.F90 -- drives the `local`-compatible element API the way a reference-side caller would
! (module state pdim, Initialize(ipoints, weights), character type tags, assumed-shape arrays) over the golden cases
! of tests/golden/element_cases.txt (the reference's own compiled modules produced the expected values, kept in
! tests/golden/element_golden.json) and prints every result with 17 significant digits; tests/test_gpu_fortran.py
! compares them at 1e-13.  Usage: test_elems_amd <cases file>
program test_elems_amd
  use local_amd
  implicit none
  character(256) :: path
  character(4) :: tname
  integer :: ncases, ic, n, pd, np, cp, nd, i, last_pd
  real(8) :: E, nu, detj
  real(8), allocatable :: ec(:, :), u(:), xi(:), k(:, :), es(:, :), dN(:, :), B(:, :), D(:, :), dNxi(:, :)
  real(8), pointer :: N1(:)
  integer, allocatable :: enodes(:), indx(:)
  call get_command_argument(1, path)
  open(11, file=trim(path), status='old')
  read(11, *) ncases
  last_pd = 0
  do ic = 1, ncases
    read(11, *) tname, n, pd, np
    read(11, *) E, nu
    allocate(ec(n, pd), u(n*pd), xi(pd))
    read(11, *) ec
    read(11, *) u
    read(11, *) xi
    if (pd /= last_pd) then              ! src/main.F90:40-41: pdim is set, then Initialize builds the tables
      pdim = pd
      call Initialize(ipoints, weights)
      last_pd = pd
    end if
    cp = cpdim; nd = n*pd
    allocate(k(nd, nd), es(np, cp), dN(pd, n), B(cp, nd), D(cp, cp), dNxi(pd, n), enodes(n), indx(nd))
    write(*, '(A,1X,A,4(1X,I0))') "CASE", trim(tname), n, pd, np, cp
    call FormElKE(trim(tname), ec, E, nu, 1.0d0, k)
    write(*, '(A)', advance='no') "K"; write(*, '(*(1X,ES24.16E3))') k
    call CalcElStress(trim(tname), ec, u, E, nu, es)
    write(*, '(A)', advance='no') "S"; write(*, '(*(1X,ES24.16E3))') es
    call FormdNdetJ(trim(tname), xi, ec, dN, detj)
    write(*, '(A)', advance='no') "DN"; write(*, '(*(1X,ES24.16E3))') dN
    write(*, '(A,1X,ES24.16E3)') "DETJ", detj
    call BMat(n, dN, B)
    write(*, '(A)', advance='no') "B"; write(*, '(*(1X,ES24.16E3))') B
    call DMat(D, E, nu)
    write(*, '(A)', advance='no') "D"; write(*, '(*(1X,ES24.16E3))') D
    call ShapeFunc(trim(tname), 1, N1)
    write(*, '(A)', advance='no') "N1"; write(*, '(*(1X,ES24.16E3))') N1
    call ShapeFuncd(trim(tname), dNxi, xi)
    write(*, '(A)', advance='no') "DNXI"; write(*, '(*(1X,ES24.16E3))') dNxi
    write(*, '(A)', advance='no') "IP"; write(*, '(*(1X,ES24.16E3))') ipoints(getElTypeNo(trim(tname)))%array
    write(*, '(A)', advance='no') "W"; write(*, '(*(1X,ES24.16E3))') weights(getElTypeNo(trim(tname)))%array
    enodes = (/(9 + i, i = 1, n)/)
    call FormElIndx(enodes, indx)
    write(*, '(A,*(1X,I0))') "INDX", indx
    write(*, '(A,5(1X,I0))') "CNT", getNodeCount(trim(tname)), getNip(trim(tname)), getDim(trim(tname)), &
      getNps(trim(tname)), getVtkid(trim(tname))
    deallocate(ec, u, xi, k, es, dN, B, D, dNxi, enodes, indx)
  end do
  close(11)
  call ElemsAmdFinalize()
end program test_elems_amd
