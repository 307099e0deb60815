/* tests/c/abi_smoke.c -- the boundary used from plain C (gcc, no C++/Python/torch):
 * the reference's elastic example (2 quad4, E=3640, nu=0.3, top edge u_y = 0.1 in 10
 * increments) through include/fem_amd.h only.  Prints the aggregate displacements and
 * nodal stresses; tests/test_gpu_fortran.py compiles and runs it. */
#include <stdio.h>
#include <stdlib.h>

#include "fem_amd.h"

#define CK(call)                                                              \
    do {                                                                      \
        int ierr_ = (call);                                                   \
        if (ierr_) { fprintf(stderr, "%s -> %d: %s\n", #call, ierr_, fem_last_error()); return 1; } \
    } while (0)

int main(void)
{
    const int eltype[2] = {FEM_QUA, FEM_QUA}, eptr[3] = {0, 4, 8}, conn[8] = {0, 1, 4, 3, 1, 2, 5, 4}, mat_id[2] = {0, 0};
    const double coords[12] = {0, 2, 0, 1, 0, 0, 1, 2, 1, 1, 1, 0};
    const double mats[5] = {3640, 0.3, 1e18, 1.0, 3000.0};
    const int bcnode[18] = {0, 0, 0, 1, 0, 1, 2, 0, 0, 3, 0, 0, 4, 0, 1, 5, 0, 0};
    const double bcval[12] = {0, 0.1, 0, 0, 0, 0, 0, 0.1, 0, 0, 0, 0};
    const double t = 1.0, dt = 0.1;
    double u[12] = {0}, s[18] = {0}, uu[12], sn[18];
    fem_ctx ctx; fem_mesh mesh; fem_mat K; fem_vec F, U; fem_ksp ksp;
    int step, i, its, reason;

    CK(fem_ctx_create(0, &ctx));
    CK(fem_mesh_create(ctx, 2, 2, 6, eltype, eptr, conn, mat_id, coords, 1, mats, &mesh));
    CK(fem_mat_create_aij(mesh, &K));
    CK(fem_mat_assemble_elements(K));
    CK(fem_mat_apply_kbc(K, 6, bcnode));
    CK(fem_vec_create(ctx, 12, &F));
    CK(fem_vec_create(ctx, 12, &U));
    CK(fem_ksp_create(ctx, &ksp));
    CK(fem_ksp_set_operators(ksp, K));
    CK(fem_ksp_set_from_options(ksp, "-ksp_type cg -pc_type jacobi -ksp_rtol 1e-14"));
    CK(fem_ksp_set_up(ksp));
    /* KSPSetTolerances between two solves of one F (src/main.F90:349-351): the second solve runs under the new tolerances and is
     * never the solution kept from the first (-ksp_reuse_solution, on by default); a third, unchanged, solve is the kept one */
    {
        int its_a, its_b, its_c, reused_b, reused_c;
        CK(fem_form_rhs(mesh, 6, bcnode, bcval, 0, NULL, NULL, 0, NULL, NULL, t, 0.0, dt, F));
        CK(fem_ksp_set_tolerances(ksp, 1e-2, FEM_DEFAULT_REAL, FEM_DEFAULT_REAL, FEM_DEFAULT_INT));
        CK(fem_ksp_solve(ksp, F, U));
        CK(fem_ksp_get_iteration_number(ksp, &its_a));
        CK(fem_ksp_set_tolerances(ksp, 1e-14, FEM_DEFAULT_REAL, FEM_DEFAULT_REAL, FEM_DEFAULT_INT));
        CK(fem_ksp_solve(ksp, F, U));
        CK(fem_ksp_get_iteration_number(ksp, &its_b));
        CK(fem_ksp_get_reused_solves(ksp, &reused_b));
        CK(fem_ksp_solve(ksp, F, U));
        CK(fem_ksp_get_iteration_number(ksp, &its_c));
        CK(fem_ksp_get_reused_solves(ksp, &reused_c));
        printf("RETOL %d %d %d %d %d\n", its_a, its_b, its_c, reused_b, reused_c);
    }
    for (step = 0; dt * step < t; step++) {
        CK(fem_form_rhs(mesh, 6, bcnode, bcval, 0, NULL, NULL, 0, NULL, NULL, t, dt * step, dt, F));
        CK(fem_ksp_solve(ksp, F, U));
        CK(fem_ksp_get_converged_reason(ksp, &reason));
        CK(fem_ksp_get_iteration_number(ksp, &its));
        if (reason <= 0) { fprintf(stderr, "KSP reason %d\n", reason); return 2; }
        CK(fem_vec_get_array(U, uu));
        CK(fem_recover_nodal_stress(mesh, U, sn));
        for (i = 0; i < 12; i++) u[i] += uu[i];
        for (i = 0; i < 18; i++) s[i] += sn[i];
    }
    printf("STEPS %d\n", step);
    for (i = 0; i < 6; i++) printf("NODE %d %.15e %.15e %.15e %.15e %.15e\n", i, u[2 * i], u[2 * i + 1], s[i], s[6 + i], s[12 + i]);
    CK(fem_ksp_destroy(ksp)); CK(fem_vec_destroy(F)); CK(fem_vec_destroy(U));
    CK(fem_mat_destroy(K)); CK(fem_mesh_destroy(mesh)); CK(fem_ctx_destroy(ctx));
    return 0;
}
