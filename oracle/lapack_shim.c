/* oracle/lapack_shim.c -- TEST INFRASTRUCTURE.
 *
 * The reference's MatInv (/root/reference/src/m_utils.F90:45-57) calls LAPACK
 * dgetrf/dgetri.  This image has no system LAPACK, but a real one ships inside
 * scipy's bundled OpenBLAS with `scipy_`-prefixed symbols.  These two
 * forwarders only rename the symbols; the factorisation and inverse are the
 * genuine LAPACK routines, not a stand-in.
 */
extern void scipy_dgetrf_(int *m, int *n, double *a, int *lda, int *ipiv, int *info);
extern void scipy_dgetri_(int *n, double *a, int *lda, int *ipiv, double *work, int *lwork, int *info);

void dgetrf_(int *m, int *n, double *a, int *lda, int *ipiv, int *info)
{
    scipy_dgetrf_(m, n, a, lda, ipiv, info);
}

void dgetri_(int *n, double *a, int *lda, int *ipiv, double *work, int *lwork, int *info)
{
    scipy_dgetri_(n, a, lda, ipiv, work, lwork, info);
}
