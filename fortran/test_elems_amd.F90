! fortran/test_elems_amd.F90 -- exercises the `local`-compatible element API the
! way a reference-side caller would (character type tags, assumed-shape arrays)
! and prints known answers (unit quad / unit hex, E=3640 nu=0.3) for the tests.
program test_elems_amd
  use local_amd
  implicit none
  real(8) :: ecq(4, 2), kq(8, 8), ech(8, 3), kh(24, 24), u(8), es(4, 3)
  integer :: indx(8), i
  pdim = 2
  call ElemsAmdInit(0)
  ecq = reshape((/0.d0, 1.d0, 1.d0, 0.d0, 0.d0, 0.d0, 1.d0, 1.d0/), (/4, 2/))
  call FormElKE("qua", ecq, 3640.d0, 0.3d0, 1.d0, kq)
  write(*, '(A,8(F12.4,1X))') "KQ1 ", kq(1, :)
  call FormElIndx((/4, 9, 2, 7/), indx)
  write(*, '(A,8(I0,1X))') "INDX ", indx
  u = (/0.d0, 0.d0, 0.d0, 0.d0, 0.d0, 0.05d0, 0.d0, 0.05d0/)     ! uniform eps_yy = 0.05
  call CalcElStress("qua", ecq, u, 3640.d0, 0.3d0, es)
  write(*, '(A,3(F12.6,1X))') "SQ1 ", es(1, :)
  write(*, '(A,5(I0,1X))') "CNT ", getNodeCount("hex"), getNip("hex"), getDim("hex"), getNps("hex"), getVtkid("hex")
  pdim = 3
  call ElemsAmdInit(0)
  ech = reshape((/0.d0, 1.d0, 1.d0, 0.d0, 0.d0, 1.d0, 1.d0, 0.d0, &
                  0.d0, 0.d0, 1.d0, 1.d0, 0.d0, 0.d0, 1.d0, 1.d0, &
                  0.d0, 0.d0, 0.d0, 0.d0, 1.d0, 1.d0, 1.d0, 1.d0/), (/8, 3/))
  call FormElKE("hex", ech, 3640.d0, 0.3d0, 1.d0, kh)
  write(*, '(A,6(F12.4,1X))') "KH1 ", kh(1, 1:6)
  call ElemsAmdFinalize()
end program test_elems_amd
