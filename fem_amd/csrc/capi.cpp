// fem_amd/csrc/capi.cpp -- extern "C" surface of libfem_amd.so (include/fem_amd.h).
// Every entry point converts C++ exceptions into PETSc-style integer error
// codes; fem_last_error() returns the message of the calling thread's last failure.
#include <cstring>

#include <algorithm>
#include <vector>

#include "comm.h"
#include "direct.h"
#include "objects.h"

using namespace fem;

// implemented in the .hip translation units
namespace fem { void upload_elem_tables(hipStream_t s); }
void device_copy_segments(hipStream_t s, int n, void* const* dst, const void* const* src, const size_t* bytes);   // mat.hip
fem_mesh mesh_create_impl(fem_ctx, int, int, int, const int*, const int*, const int*, const int*, const double*, int,
                          const double*);
void mesh_set_halo_impl(fem_mesh, fem_comm, int, const int*, const int*, const int*);
fem_mat mat_create_impl(fem_mesh);
void mat_assemble_elements_impl(fem_mat);
void coh_apply_1_impl(fem_ctx, int, const double*, const double*, const double*, double, double*, double*, double*);
void mat_set_mult_type_impl(fem_mat, const char*);
void mat_set_profiling_impl(fem_mat, int);
void mat_get_profile_impl(fem_mat, int, int*, double*, double*, double*);
void mat_get_mult_info_impl(fem_mat, int*, long long*, long long*, long long*, long long*, long long*, double*);
void mat_get_tile_geometry_info_impl(fem_mat, long long*, int*, int*);
void mesh_tile_plan_digest_impl(fem_mesh, const char*, unsigned long long*, double*);
void mat_apply_kbc_impl(fem_mat, int, const int*);
void mat_get_csr_impl(fem_mat, int*, int*, double*);
void mat_set_values_impl(fem_mat, int, const int*, int, const int*, const double*, int);
void mat_mult_raw(fem_mat, const double*, double*, double*, const int*);
void mat_get_diagonal_raw(fem_mat, double*);
void form_el_ke_impl(fem_ctx, int, int, const double*, const double*, const double*, double*);
void calc_el_stress_impl(fem_ctx, int, int, const double*, const double*, const double*, const double*, double*);
void recover_nodal_stress_dev(fem_mesh, const double*, double*);
void form_rhs_impl(fem_mesh, int, const int*, const double*, int, const int*, const double*, int, const int*,
                   const double*, double, double, double, double*);
fem_ksp ksp_new(fem_ctx);
void ksp_delete(fem_ksp);
void ksp_set_operators(fem_ksp, fem_mat);
void ksp_set_type(fem_ksp, const char*);
void ksp_set_pc_type(fem_ksp, const char*);
void ksp_set_tolerances(fem_ksp, double, double, double, int);
void ksp_set_from_options_impl(fem_ksp, const char*);
void ksp_set_up_impl(fem_ksp);
void ksp_solve_impl(fem_ksp, fem_vec, fem_vec);
int ksp_get_reused(fem_ksp);
int ksp_get_resident(fem_ksp);
void ksp_get_result(fem_ksp, int*, double*, int*);
int ksp_get_history(fem_ksp, double*, int);
void ksp_set_profiling(fem_ksp, int);
void ksp_get_profile(fem_ksp, int*, double*, double*);
void vec_fill(fem_vec, double);
void vec_axpy(fem_vec, double, fem_vec);
double vec_dot(fem_vec, fem_vec);
fem_comm comm_create_self_impl(fem_ctx);
void comm_get_unique_id_impl(char*);
fem_comm comm_create_rccl_impl(fem_ctx, int, int, const char*);
void comm_create_local_group_impl(int, const fem_ctx*, fem_comm*);
fem_comm comm_create_callbacks_impl(fem_ctx, int, int, void (*)(void*, double*, int),
                                    void (*)(void*, int, const int*, const long long*, const double*, double*), void*);

void shape_funcd_impl(fem_ctx, int, int, const double*, double*);
void form_dn_detj_impl(fem_ctx, int, int, const double*, const double*, double*, double*);
void bmat_impl(fem_ctx, int, int, int, const double*, double*);
void dmat_impl(fem_ctx, int, int, const double*, const double*, double*);
void mesh_set_cohesive_impl(fem_mesh, const int*, int, const double*);
void mat_add_cohesive_impl(fem_mat, const double*, double, int, const int*);
void vec_add_cohesive_traction_impl(fem_mesh, const double*, double, int, const int*, double*);
void enforce_bc_force_impl(fem_mesh, int, const int*, const double*, double, double, double*);
void vec_scale(fem_vec, double);
void vec_waxpy(fem_vec w, double a, fem_vec x, fem_vec y);
extern "C" {
typedef int (*fem_snes_function_cb_t)(fem_snes, fem_vec, fem_vec, void*);
typedef int (*fem_snes_jacobian_cb_t)(fem_snes, fem_vec, fem_mat, fem_mat, void*);
}
fem_snes snes_new(fem_ctx);
void snes_delete(fem_snes);
void snes_set_function(fem_snes, fem_vec, fem_snes_function_cb_t, void*);
void snes_set_jacobian(fem_snes, fem_mat, fem_mat, fem_snes_jacobian_cb_t, void*);
fem_ksp snes_get_ksp(fem_snes);
void snes_set_from_options_impl(fem_snes, const char*);
void snes_solve_impl(fem_snes, fem_vec);
void snes_get_result(fem_snes, int*, int*, double*, int*, int*);

namespace fem {
static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
}  // namespace fem

#define FEM_TRY try {
#define FEM_CATCH                                                              \
    }                                                                          \
    catch (const fem::Error& e) { fem::set_last_error(e.what()); return e.code ? e.code : 1; } \
    catch (const std::exception& e) { fem::set_last_error(e.what()); return 99; }              \
    return 0;

extern "C" {

int fem_version(void) { return 100; }
const char* fem_last_error(void) { return fem::g_last_error.c_str(); }

int fem_device_count(int* count)
{
    FEM_TRY
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    FEM_CATCH
}

int fem_ctx_create(int device, fem_ctx* out)
{
    FEM_TRY
    int n = 0;
    FEM_CHECK(hipGetDeviceCount(&n) == hipSuccess && n > 0, 50,
              "fem_ctx_create: no HIP device available (libfem_amd has no CPU path)");
    FEM_CHECK(device >= 0 && device < n, 50, "fem_ctx_create: device index out of range");
    std::unique_ptr<fem_ctx_s> c(new fem_ctx_s);
    c->device = device;
    c->bind();
    hipDeviceProp_t prop;
    FEM_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount;
    FEM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    FEM_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    fem::pool_register_stream(c->stream);        // (released device blocks are ordered behind the work of these two: pool.cpp)
    fem::pool_register_stream(c->comm_stream);
    FEM_HIP(hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
    FEM_HIP(hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming));
    FEM_HIP(hipEventCreateWithFlags(&c->ev_c, hipEventDisableTiming));
    FEM_HIP(hipEventCreate(&c->t0));
    FEM_HIP(hipEventCreate(&c->t1));
    fem::upload_elem_tables(c->stream);
    FEM_HIP(hipStreamSynchronize(c->stream));
    *out = c.release();
    FEM_CATCH
}

int fem_ctx_destroy(fem_ctx c)
{
    FEM_TRY
    if (!c) return 0;
    c->bind();
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->comm_stream);
    fem::pool_unregister_stream(c->stream);
    fem::pool_unregister_stream(c->comm_stream);
    (void)hipEventDestroy(c->t0); (void)hipEventDestroy(c->t1);
    (void)hipStreamDestroy(c->stream);
    (void)hipStreamDestroy(c->comm_stream);
    (void)hipEventDestroy(c->ev_a); (void)hipEventDestroy(c->ev_b); (void)hipEventDestroy(c->ev_c);
    delete c;
    pool_trim();   // cached device blocks go back to the driver with the context
    FEM_CATCH
}

int fem_ctx_synchronize(fem_ctx c)
{
    FEM_TRY
    FEM_CHECK(c, 1, "null context");
    c->bind();
    FEM_HIP(hipStreamSynchronize(c->stream));
    FEM_CATCH
}

int fem_ctx_timer_start(fem_ctx c)
{
    FEM_TRY
    FEM_CHECK(c, 1, "null context");
    c->bind();
    FEM_HIP(hipEventRecord(c->t0, c->stream));
    FEM_CATCH
}

int fem_ctx_timer_stop(fem_ctx c, double* ms)
{
    FEM_TRY
    FEM_CHECK(c, 1, "null context");
    c->bind();
    FEM_HIP(hipEventRecord(c->t1, c->stream));
    FEM_HIP(hipEventSynchronize(c->t1));
    float f = 0;
    FEM_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
    *ms = f;
    FEM_CATCH
}

// ---- element library ------------------------------------------------------------------
int fem_el_counts(int t, int* nodecount, int* nip, int* dim, int* nps, int* vtkid)
{
    FEM_TRY
    FEM_CHECK(t >= FEM_TRI && t <= FEM_COH, 3, "fem_el_counts: unknown element type");
    const ElemTable& tb = host_elem_tables()[t];
    if (nodecount) *nodecount = tb.npe;
    if (nip) *nip = tb.nip;
    if (dim) *dim = tb.dim;
    if (nps) *nps = tb.nps;
    if (vtkid) *vtkid = tb.vtk;
    FEM_CATCH
}

int fem_el_tables(int t, double* ipoints, double* weights, double* shape, double* n2inv, int* has_inv)
{
    FEM_TRY
    FEM_CHECK(t >= FEM_TRI && t <= FEM_COH, 3, "fem_el_tables: unknown element type");
    const ElemTable& tb = host_elem_tables()[t];
    for (int g = 0; g < tb.nip; g++) {
        if (ipoints) for (int d = 0; d < tb.dim; d++) ipoints[g + d * tb.nip] = tb.xi[g][d];
        if (weights) weights[g] = tb.w[g];
        if (shape) for (int a = 0; a < tb.npe; a++) shape[g + a * tb.nip] = tb.N[g][a];
    }
    if (n2inv && tb.has_inv)
        for (int a = 0; a < tb.npe; a++)
            for (int g = 0; g < tb.nip; g++) n2inv[a + g * tb.npe] = tb.n2inv[a][g];
    if (has_inv) *has_inv = tb.has_inv;
    FEM_CATCH
}

int fem_form_el_ke(fem_ctx ctx, int t, int nel, const double* ecoords, const double* E, const double* nu, double* k)
{
    FEM_TRY
    FEM_CHECK(ctx, 1, "null context");
    form_el_ke_impl(ctx, t, nel, ecoords, E, nu, k);
    FEM_CATCH
}

int fem_calc_el_stress(fem_ctx ctx, int t, int nel, const double* ecoords, const double* edisp, const double* E,
                       const double* nu, double* estress)
{
    FEM_TRY
    FEM_CHECK(ctx, 1, "null context");
    calc_el_stress_impl(ctx, t, nel, ecoords, edisp, E, nu, estress);
    FEM_CATCH
}

int fem_form_el_indx(int pdim, int n, const int* enodes, int* indx)
{
    FEM_TRY
    FEM_CHECK(pdim == 2 || pdim == 3, 2, "fem_form_el_indx: pdim must be 2 or 3");
    for (int j = 0; j < n; j++)
        for (int d = 0; d < pdim; d++) indx[pdim * j + d] = pdim * enodes[j] + d;
    FEM_CATCH
}

int fem_shape_func(int t, int gp, double* N)
{
    FEM_TRY
    FEM_CHECK(t >= FEM_TRI && t <= FEM_COH, 3, "fem_shape_func: unknown element type");
    const ElemTable& tb = host_elem_tables()[t];
    FEM_CHECK(gp >= 0 && gp < tb.nip && N, 2, "fem_shape_func: integration point out of range");
    for (int a = 0; a < tb.npe; a++) N[a] = tb.N[gp][a];
    FEM_CATCH
}

int fem_shape_funcd(fem_ctx ctx, int t, int npts, const double* coord, double* dN)
{
    FEM_TRY
    FEM_CHECK(ctx, 1, "null context");
    shape_funcd_impl(ctx, t, npts, coord, dN);
    FEM_CATCH
}

int fem_form_dn_detj(fem_ctx ctx, int t, int nel, const double* ipcoord, const double* ecoords, double* dN, double* detj)
{
    FEM_TRY
    FEM_CHECK(ctx, 1, "null context");
    form_dn_detj_impl(ctx, t, nel, ipcoord, ecoords, dN, detj);
    FEM_CATCH
}

int fem_bmat(fem_ctx ctx, int pdim, int nodecount, int nel, const double* dN, double* B)
{
    FEM_TRY
    FEM_CHECK(ctx, 1, "null context");
    bmat_impl(ctx, pdim, nodecount, nel, dN, B);
    FEM_CATCH
}

int fem_dmat(fem_ctx ctx, int pdim, int nmat, const double* E, const double* nu, double* D)
{
    FEM_TRY
    FEM_CHECK(ctx, 1, "null context");
    dmat_impl(ctx, pdim, nmat, E, nu, D);
    FEM_CATCH
}

int fem_coh_apply_1(fem_ctx ctx, int ncoh, const double* ecoords, const double* u8, const double* props, double dt,
                    double* tract, double* stiff, double* gp)
{
    FEM_TRY
    FEM_CHECK(ctx, 1, "fem_coh_apply_1: null context");
    coh_apply_1_impl(ctx, ncoh, ecoords, u8, props, dt, tract, stiff, gp);
    FEM_CATCH
}

int fem_rcm_order(int n, const int* ptr, const int* adj, int* perm)
{
    FEM_TRY
    FEM_CHECK(n >= 0 && ptr && perm && (adj || ptr[n] == 0), 2, "fem_rcm_order: bad arguments");
    std::vector<int> p(ptr, ptr + n + 1), a(adj, adj + ptr[n]);
    for (int v : a) FEM_CHECK(v >= 0 && v < n, 2, "fem_rcm_order: neighbour id out of range");
    const std::vector<int> out = fem::rcm_order(n, p, a);
    std::copy(out.begin(), out.end(), perm);
    FEM_CATCH
}

// ---- mesh -----------------------------------------------------------------------------
int fem_mesh_create(fem_ctx ctx, int pdim, int nels, int nnds, const int* eltype, const int* eptr, const int* conn,
                    const int* mat_id, const double* coords, int nmats, const double* mats, fem_mesh* mesh)
{
    FEM_TRY
    *mesh = mesh_create_impl(ctx, pdim, nels, nnds, eltype, eptr, conn, mat_id, coords, nmats, mats);
    FEM_CATCH
}

int fem_mesh_destroy(fem_mesh m)
{
    FEM_TRY
    if (m) { m->ctx->bind(); delete m; }
    FEM_CATCH
}

int fem_mesh_set_halo(fem_mesh mesh, fem_comm comm, int nneigh, const int* neigh_rank, const int* neigh_ptr,
                      const int* neigh_nodes)
{
    FEM_TRY
    mesh_set_halo_impl(mesh, comm, nneigh, neigh_rank, neigh_ptr, neigh_nodes);
    FEM_CATCH
}

// ---- Mat ---------------------------------------------------------------------------------
int fem_mat_create_aij(fem_mesh mesh, fem_mat* mat)
{
    FEM_TRY
    *mat = mat_create_impl(mesh);
    FEM_CATCH
}

int fem_mat_destroy(fem_mat A)
{
    FEM_TRY
    if (A) { A->ctx->bind(); delete A; }
    FEM_CATCH
}

int fem_mat_zero_entries(fem_mat A)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    A->ctx->bind();
    A->lazy_zero = true;   // materialised by the first reader; the assembly overwrites instead (fem_mat_s::values())
    A->state++;
    A->bc_saved = false;   // the next ApplyKBC sees a freshly assembled diagonal again
    FEM_CATCH
}

int fem_mat_assemble_elements(fem_mat A)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    A->state++;
    mat_assemble_elements_impl(A);
    FEM_CATCH
}

int fem_mat_set_mult_type(fem_mat A, const char* type)
{
    FEM_TRY
    FEM_CHECK(A && type, 1, "null argument");
    mat_set_mult_type_impl(A, type);
    FEM_CATCH
}

int fem_mat_set_profiling(fem_mat A, int on)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    mat_set_profiling_impl(A, on);
    FEM_CATCH
}

int fem_mat_get_profile(fem_mat A, int max_products, int* n, double* element_ms, double* gather_ms, double* penalty_ms)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    mat_get_profile_impl(A, max_products, n, element_ms, gather_ms, penalty_ms);
    FEM_CATCH
}

int fem_mat_get_mult_info(fem_mat A, int* kind, long long* tiles, long long* affine_tiles, long long* element_visits,
                          long long* lane_slots, long long* tile_nodes, double* setup_ms)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    mat_get_mult_info_impl(A, kind, tiles, affine_tiles, element_visits, lane_slots, tile_nodes, setup_ms);
    FEM_CATCH
}

int fem_mat_get_tile_geometry_info(fem_mat A, long long* uniform_tiles, int* per_tile_records, int* streaming_lists)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    mat_get_tile_geometry_info_impl(A, uniform_tiles, per_tile_records, streaming_lists);
    FEM_CATCH
}

int fem_mesh_tile_plan_digest(fem_mesh m, const char* builder, unsigned long long* digest, double* setup_ms)
{
    FEM_TRY
    FEM_CHECK(m && builder && digest, 1, "null argument");
    mesh_tile_plan_digest_impl(m, builder, digest, setup_ms);
    FEM_CATCH
}

int fem_mat_apply_kbc(fem_mat A, int nbcs, const int* bcnode)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    A->state++;
    mat_apply_kbc_impl(A, nbcs, bcnode);
    FEM_CATCH
}

int fem_mat_get_info(fem_mat A, long long* nrows, long long* nnz, long long* device_bytes)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    if (nrows) *nrows = (long long)A->nbrows * A->bs;
    if (nnz) *nnz = A->nblocks * A->bs * A->bs;
    if (device_bytes) *device_bytes = (long long)(A->val.bytes() + A->bcol.bytes() + A->boff.bytes() + A->rowlen.bytes());
    FEM_CATCH
}

int fem_mat_get_csr(fem_mat A, int* rowptr, int* colind, double* val)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    mat_get_csr_impl(A, rowptr, colind, val);
    FEM_CATCH
}

int fem_mat_set_values(fem_mat A, int m, const int* rows, int n, const int* cols, const double* v, int mode)
{
    FEM_TRY
    FEM_CHECK(A, 1, "null matrix");
    FEM_CHECK(m >= 0 && n >= 0 && (m == 0 || rows) && (n == 0 || cols) && (m == 0 || n == 0 || v), 2, "fem_mat_set_values: bad arguments");
    FEM_CHECK(mode == 1 || mode == 2, 2, "fem_mat_set_values: mode must be 1 (INSERT_VALUES) or 2 (ADD_VALUES)");
    A->state++;
    if (m && n) mat_set_values_impl(A, m, rows, n, cols, v, mode);
    FEM_CATCH
}

int fem_mat_mult(fem_mat A, fem_vec x, fem_vec y)
{
    FEM_TRY
    FEM_CHECK(A && x && y, 1, "null handle");
    const long long n = (long long)A->nbrows * A->bs;
    FEM_CHECK(x->n == n && y->n == n && x != y, 8, "fem_mat_mult: vector size mismatch");
    A->ctx->bind();
    mat_mult_raw(A, x->d.p, y->d.p, nullptr, nullptr);
    halo_sum(A->mesh, y->d.p, A->bs, A->bs, 1);
    FEM_HIP(hipGetLastError());
    FEM_CATCH
}

int fem_mat_mult_timed(fem_mat A, fem_vec x, fem_vec y, int reps, double* avg_ms)
{
    FEM_TRY
    FEM_CHECK(A && x && y && reps > 0, 1, "bad arguments");
    const long long n = (long long)A->nbrows * A->bs;
    FEM_CHECK(x->n == n && y->n == n && x != y, 8, "fem_mat_mult_timed: vector size mismatch");
    fem_ctx c = A->ctx;
    c->bind();
    mat_mult_raw(A, x->d.p, y->d.p, A->spmv_partial.p, nullptr);  // warm-up
    FEM_HIP(hipEventRecord(c->t0, c->stream));
    for (int i = 0; i < reps; i++) mat_mult_raw(A, x->d.p, y->d.p, A->spmv_partial.p, nullptr);
    FEM_HIP(hipEventRecord(c->t1, c->stream));
    FEM_HIP(hipEventSynchronize(c->t1));
    float ms = 0;
    FEM_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    *avg_ms = ms / reps;
    FEM_CATCH
}

int fem_mat_get_diagonal(fem_mat A, fem_vec d)
{
    FEM_TRY
    FEM_CHECK(A && d && d->n == (long long)A->nbrows * A->bs, 8, "fem_mat_get_diagonal: size mismatch");
    A->ctx->bind();
    mat_get_diagonal_raw(A, d->d.p);
    halo_sum(A->mesh, d->d.p, A->bs, A->bs, 1);
    FEM_HIP(hipGetLastError());
    FEM_CATCH
}

// ---- Vec ---------------------------------------------------------------------------------
int fem_vec_create(fem_ctx ctx, long long n, fem_vec* v)
{
    FEM_TRY
    FEM_CHECK(ctx && n >= 0, 1, "fem_vec_create: bad arguments");
    ctx->bind();
    std::unique_ptr<fem_vec_s> x(new fem_vec_s);
    x->ctx = ctx; x->n = n;
    x->d.alloc((size_t)std::max<long long>(n, 1));
    x->d.zero(ctx->stream);
    *v = x.release();
    FEM_CATCH
}

int fem_vec_destroy(fem_vec v)
{
    FEM_TRY
    if (v) { v->ctx->bind(); delete v; }
    FEM_CATCH
}

int fem_vec_get_size(fem_vec v, long long* n)
{
    FEM_TRY
    FEM_CHECK(v, 1, "null vector");
    *n = v->n;
    FEM_CATCH
}

int fem_vec_set(fem_vec v, double alpha)
{
    FEM_TRY
    FEM_CHECK(v, 1, "null vector");
    vec_fill(v, alpha);
    FEM_CATCH
}

int fem_vec_set_array(fem_vec v, const double* host)
{
    FEM_TRY
    FEM_CHECK(v && host, 1, "null argument");
    v->ctx->bind();
    v->d.upload(host, (size_t)v->n, v->ctx->stream);
    FEM_HIP(hipStreamSynchronize(v->ctx->stream));
    FEM_CATCH
}

int fem_vec_get_array(fem_vec v, double* host)
{
    FEM_TRY
    FEM_CHECK(v && host, 1, "null argument");
    v->ctx->bind();
    v->d.download(host, (size_t)v->n, v->ctx->stream);
    FEM_HIP(hipStreamSynchronize(v->ctx->stream));
    FEM_CATCH
}

int fem_vec_device_ptr(fem_vec v, void** dptr)
{
    FEM_TRY
    FEM_CHECK(v, 1, "null vector");
    *dptr = v->d.p;
    FEM_CATCH
}

int fem_vec_copy(fem_vec src, fem_vec dst)
{
    FEM_TRY
    FEM_CHECK(src && dst && src->n == dst->n, 8, "fem_vec_copy: size mismatch");
    src->ctx->bind();
    FEM_HIP(hipMemcpyAsync(dst->d.p, src->d.p, sizeof(double) * (size_t)src->n, hipMemcpyDeviceToDevice,
                           src->ctx->stream));
    FEM_CATCH
}

int fem_vec_axpy(fem_vec y, double a, fem_vec x)
{
    FEM_TRY
    FEM_CHECK(x && y && x->n == y->n, 8, "fem_vec_axpy: size mismatch");
    vec_axpy(y, a, x);
    FEM_CATCH
}

int fem_vec_dot(fem_vec x, fem_vec y, double* dot)
{
    FEM_TRY
    FEM_CHECK(x && y && x->n == y->n, 8, "fem_vec_dot: size mismatch");
    *dot = vec_dot(x, y);
    FEM_CATCH
}

int fem_vec_norm2(fem_vec x, double* nrm)
{
    FEM_TRY
    FEM_CHECK(x, 1, "null vector");
    *nrm = std::sqrt(vec_dot(x, x));
    FEM_CATCH
}

// ---- RHS ---------------------------------------------------------------------------------
int fem_form_rhs(fem_mesh mesh, int nbcs, const int* bcnode, const double* bcval, int nfrcs, const int* fnode,
                 const double* fval, int ntrcs, const int* telsd, const double* tval, double t_total, double t_init,
                 double dt, fem_vec F)
{
    FEM_TRY
    FEM_CHECK(mesh && F && F->n == (long long)mesh->nnds * mesh->pdim, 8, "fem_form_rhs: vector size mismatch");
    form_rhs_impl(mesh, nbcs, bcnode, bcval, nfrcs, fnode, fval, ntrcs, telsd, tval, t_total, t_init, dt, F->d.p);
    FEM_CATCH
}

// ---- KSP ---------------------------------------------------------------------------------
int fem_ksp_create(fem_ctx ctx, fem_ksp* ksp)
{
    FEM_TRY
    FEM_CHECK(ctx, 1, "null context");
    *ksp = ksp_new(ctx);
    FEM_CATCH
}
int fem_ksp_destroy(fem_ksp k)
{
    FEM_TRY
    if (k) ksp_delete(k);
    FEM_CATCH
}
int fem_ksp_set_operators(fem_ksp k, fem_mat A)
{
    FEM_TRY
    FEM_CHECK(k && A, 1, "null handle");
    ksp_set_operators(k, A);
    FEM_CATCH
}
int fem_ksp_set_type(fem_ksp k, const char* t)
{
    FEM_TRY
    FEM_CHECK(k, 1, "null handle");
    ksp_set_type(k, t);
    FEM_CATCH
}
int fem_ksp_set_pc_type(fem_ksp k, const char* t)
{
    FEM_TRY
    FEM_CHECK(k, 1, "null handle");
    ksp_set_pc_type(k, t);
    FEM_CATCH
}
int fem_ksp_set_tolerances(fem_ksp k, double rtol, double abstol, double dtol, int maxits)
{
    FEM_TRY
    FEM_CHECK(k, 1, "null handle");
    ksp_set_tolerances(k, rtol, abstol, dtol, maxits);
    FEM_CATCH
}
int fem_ksp_set_from_options(fem_ksp k, const char* options)
{
    FEM_TRY
    FEM_CHECK(k, 1, "null handle");
    ksp_set_from_options_impl(k, options);
    FEM_CATCH
}
int fem_ksp_set_up(fem_ksp k)
{
    FEM_TRY
    FEM_CHECK(k, 1, "null handle");
    ksp_set_up_impl(k);
    FEM_CATCH
}
int fem_ksp_solve(fem_ksp k, fem_vec b, fem_vec x)
{
    FEM_TRY
    FEM_CHECK(k, 1, "null handle");
    ksp_solve_impl(k, b, x);
    FEM_CATCH
}
int fem_ksp_get_iteration_number(fem_ksp k, int* its)
{
    FEM_TRY
    ksp_get_result(k, its, nullptr, nullptr);
    FEM_CATCH
}
int fem_ksp_get_residual_norm(fem_ksp k, double* rnorm)
{
    FEM_TRY
    ksp_get_result(k, nullptr, rnorm, nullptr);
    FEM_CATCH
}
int fem_ksp_get_converged_reason(fem_ksp k, int* reason)
{
    FEM_TRY
    ksp_get_result(k, nullptr, nullptr, reason);
    FEM_CATCH
}
int fem_ksp_get_residual_history(fem_ksp k, double* hist, int len, int* n)
{
    FEM_TRY
    *n = ksp_get_history(k, hist, len);
    FEM_CATCH
}
int fem_ksp_set_profiling(fem_ksp k, int on)
{
    FEM_TRY
    ksp_set_profiling(k, on);
    FEM_CATCH
}
int fem_ksp_get_profile(fem_ksp k, int* n_spmv, double* spmv_ms, double* solve_ms)
{
    FEM_TRY
    ksp_get_profile(k, n_spmv, spmv_ms, solve_ms);
    FEM_CATCH
}

int fem_ksp_get_reused_solves(fem_ksp k, int* n)
{
    FEM_TRY
    FEM_CHECK(k && n, 1, "fem_ksp_get_reused_solves: null argument");
    *n = ksp_get_reused(k);
    FEM_CATCH
}

int fem_ksp_get_resident_solves(fem_ksp k, int* n)
{
    FEM_TRY
    FEM_CHECK(k && n, 1, "fem_ksp_get_resident_solves: null argument");
    *n = ksp_get_resident(k);
    FEM_CATCH
}

// ---- cohesive elements + Newton ---------------------------------------------------------------
int fem_mesh_set_cohesive(fem_mesh mesh, const int* nlmat, int ncohmats, const double* props)
{
    FEM_TRY
    mesh_set_cohesive_impl(mesh, nlmat, ncohmats, props);
    FEM_CATCH
}

int fem_mat_duplicate(fem_mat A, fem_mat* B)
{
    FEM_TRY
    FEM_CHECK(A && B, 1, "fem_mat_duplicate: null argument");
    std::unique_ptr<fem_mat_s> D(mat_create_impl(A->mesh));
    const int rc = fem_mat_copy(A, D.get());
    if (rc) return rc;
    *B = D.release();
    FEM_CATCH
}

int fem_mat_copy(fem_mat src, fem_mat dst)
{
    FEM_TRY
    FEM_CHECK(src && dst && src->pat.get() == dst->pat.get(), 8, "fem_mat_copy: the matrices do not share a mesh (pattern)");
    src->ctx->bind();
    dst->lazy_zero = false;
    dst->state++;
    dst->unsymmetric = src->unsymmetric;
    // the copy holds the same penalty rows: with the values goes what ApplyKBC recorded about them (the constraint list, the
    // assembled diagonal under it, the shift of the element-by-element product), so that a later ApplyKBC with another list
    // restores the assembled diagonal and `-mat_mult_type ebe` on the copy keeps its penalty rows.  One launch for the five arrays.
    hipStream_t s = src->ctx->stream;
    void* dp[5]; const void* sp[5]; size_t nb[5];
    int k = 0;
    dp[k] = dst->val.p; sp[k] = src->values(); nb[k] = src->val.bytes(); k++;
    auto dup = [&](auto& d, const auto& a) {
        using T = std::remove_reference_t<decltype(*a.p)>;
        if (!a.p || !a.n) { d.release(); return; }
        if (d.n != a.n) d.alloc(a.n);
        dp[k] = d.p; sp[k] = a.p; nb[k] = a.n * sizeof(T); k++;
    };
    dup(dst->bc_dof, src->bc_dof); dup(dst->bc_shift, src->bc_shift); dup(dst->bc_diag, src->bc_diag); dup(dst->bc_node, src->bc_node);
    device_copy_segments(s, k, dp, sp, nb);
    dst->h_bc = src->h_bc;
    dst->nbc_dof = src->nbc_dof;
    dst->bc_saved = src->bc_saved;
    FEM_CATCH
}

int fem_mat_add_cohesive(fem_mat J, fem_vec u_total, double dt, int rels_mode, const int* rels_nodes)
{
    FEM_TRY
    FEM_CHECK(J && u_total && u_total->n == (long long)J->nbrows * J->bs, 8, "fem_mat_add_cohesive: vector size mismatch");
    J->state++;
    mat_add_cohesive_impl(J, u_total->d.p, dt, rels_mode, rels_nodes);
    FEM_CATCH
}

int fem_vec_add_cohesive_traction(fem_mesh mesh, fem_vec u_total, double dt, int rels_mode, const int* rels_nodes, fem_vec R)
{
    FEM_TRY
    FEM_CHECK(mesh && u_total && R, 1, "fem_vec_add_cohesive_traction: null handle");
    const long long n = (long long)mesh->nnds * mesh->pdim;
    FEM_CHECK(u_total->n == n && R->n == n, 8, "fem_vec_add_cohesive_traction: vector size mismatch");
    vec_add_cohesive_traction_impl(mesh, u_total->d.p, dt, rels_mode, rels_nodes, R->d.p);
    FEM_CATCH
}

int fem_enforce_bc_force(fem_mesh mesh, int nbcs, const int* bcnode, const double* bcval, double dt, double t_total, fem_vec R)
{
    FEM_TRY
    FEM_CHECK(mesh && R && R->n == (long long)mesh->nnds * mesh->pdim, 8, "fem_enforce_bc_force: vector size mismatch");
    enforce_bc_force_impl(mesh, nbcs, bcnode, bcval, dt, t_total, R->d.p);
    FEM_CATCH
}

int fem_mat_mult_add(fem_mat A, fem_vec x, fem_vec y, fem_vec z)
{
    FEM_TRY
    FEM_CHECK(A && x && y && z, 1, "null handle");
    const long long n = (long long)A->nbrows * A->bs;
    FEM_CHECK(x->n == n && y->n == n && z->n == n && x != z, 8, "fem_mat_mult_add: vector size mismatch (or x aliases z)");
    A->ctx->bind();
    fem::DBuf<double> t((size_t)n);
    mat_mult_raw(A, x->d.p, t.p, nullptr, nullptr);
    halo_sum(A->mesh, t.p, A->bs, A->bs, 1);
    fem_vec_s tv; tv.ctx = A->ctx; tv.n = n; tv.d = std::move(t);
    vec_waxpy(z, 1.0, &tv, y);   // (no wait for it: the pool orders the temporary's next use behind this stream, pool.cpp)
    FEM_CATCH
}

int fem_vec_scale(fem_vec v, double alpha)
{
    FEM_TRY
    FEM_CHECK(v, 1, "null vector");
    vec_scale(v, alpha);
    FEM_CATCH
}

int fem_snes_create(fem_ctx ctx, fem_snes* snes)
{
    FEM_TRY
    FEM_CHECK(ctx && snes, 1, "null argument");
    *snes = snes_new(ctx);
    FEM_CATCH
}
int fem_snes_destroy(fem_snes sn)
{
    FEM_TRY
    snes_delete(sn);
    FEM_CATCH
}
int fem_snes_get_ksp(fem_snes sn, fem_ksp* ksp)
{
    FEM_TRY
    FEM_CHECK(sn && ksp, 1, "null argument");
    *ksp = snes_get_ksp(sn);
    FEM_CATCH
}
int fem_snes_set_function(fem_snes sn, fem_vec r, fem_snes_function_cb_t f, void* c)
{
    FEM_TRY
    FEM_CHECK(sn && r && f, 1, "null argument");
    snes_set_function(sn, r, f, c);
    FEM_CATCH
}
int fem_snes_set_jacobian(fem_snes sn, fem_mat J, fem_mat P, fem_snes_jacobian_cb_t j, void* c)
{
    FEM_TRY
    FEM_CHECK(sn && J && j, 1, "null argument");
    snes_set_jacobian(sn, J, P, j, c);
    FEM_CATCH
}
static int fem_vec_waxpy(fem_vec w, double a, fem_vec x, fem_vec y)   // w = y + a x
{
    FEM_TRY
    FEM_CHECK(w && x && y && w->n == x->n && w->n == y->n, 8, "waxpy: vector size mismatch");
    vec_waxpy(w, a, x, y);
    FEM_CATCH
}
static int fem_mat_mult_sub(fem_mat A, fem_vec x, fem_vec r)           // r = A x - r
{
    FEM_TRY
    const long long n = (long long)A->nbrows * A->bs;
    FEM_CHECK(x->n == n && r->n == n && x != r, 8, "mult_sub: vector size mismatch");
    A->ctx->bind();
    fem::DBuf<double> t((size_t)n);
    mat_mult_raw(A, x->d.p, t.p, nullptr, nullptr);
    halo_sum(A->mesh, t.p, A->bs, A->bs, 1);
    fem_vec_s tv; tv.ctx = A->ctx; tv.n = n; tv.d = std::move(t);
    vec_waxpy(r, -1.0, r, &tv);
    FEM_CATCH
}
// The reference's two SNES callbacks as the library's own (src/m_global.F90:98-158 CalcJacobian, :161-235 CalcResidual): the same
// sequence of primitives a host writes by hand (fem_amd/driver.py keeps that hand-written form as well), issued
// from here -- for a host that does not want a round trip through its own language between two device calls (a Python host pays
// about 0.1 ms per call: a third of config 5's time once the inner solve takes 0.3 ms).
struct fem_defmod_callbacks_s {
    fem_mesh mesh; fem_mat K, J; fem_vec F, agg, utot;
    std::vector<int> bcnode; std::vector<double> bcval;
    int nbcs; double dt, t; int rels_mode; int rels_nodes[4]; bool have_nodes;
};
// (vector steps fused where one kernel gives the same roundings as the two calls it stands for: w = y + a x is what a copy followed
// by an axpy computes, t - R what a scaling by -1 followed by t + R does)
static int defmod_calc_jacobian(fem_snes, fem_vec du, fem_mat J, fem_mat, void* c)
{
    auto* d = static_cast<fem_defmod_callbacks_s*>(c);
    int rc;
    if ((rc = fem_mat_copy(d->K, J))) return rc;                                                     /* MatCopy(Mat_K, Jacobian) :123 */
    if ((rc = fem_vec_waxpy(d->utot, 1.0, d->agg, du))) return rc;                                   /* uu = uu + aggregate_u :127 */
    if ((rc = fem_mat_add_cohesive(J, d->utot, d->dt, d->rels_mode, d->have_nodes ? d->rels_nodes : nullptr))) return rc;   /* :129-147 */
    return fem_mat_apply_kbc(J, d->nbcs, d->bcnode.data());                                          /* ApplyKBC(Jacobian) :153 */
}
static int defmod_calc_residual(fem_snes, fem_vec du, fem_vec R, void* c)
{
    auto* d = static_cast<fem_defmod_callbacks_s*>(c);
    int rc;
    if ((rc = fem_vec_waxpy(d->utot, 1.0, d->agg, du))) return rc;                                   /* current_u = current_u + aggregate_u :182 */
    if ((rc = fem_vec_copy(d->F, R))) return rc;                                                     /* VecCopy(Vec_F, Residual) :184 */
    if ((rc = fem_vec_add_cohesive_traction(d->mesh, d->utot, d->dt, d->rels_mode, d->have_nodes ? d->rels_nodes : nullptr, R))) return rc;   /* :186-205 */
    if ((rc = fem_enforce_bc_force(d->mesh, d->nbcs, d->bcnode.data(), d->bcval.data(), d->dt, d->t, R))) return rc;   /* :210 */
    return fem_mat_mult_sub(d->J, du, R);                    /* VecScale(Residual, MINUS) :223 + MatMultAdd(Jacobian, du, Residual, Residual) :224 */
}
int fem_snes_set_defmod_callbacks(fem_snes sn, fem_mesh mesh, fem_mat K, fem_mat J, fem_vec R, fem_vec F, fem_vec aggregate_u, fem_vec work,
                                  int nbcs, const int* bcnode, const double* bcval, double dt, double t_total, int rels_mode,
                                  const int* rels_nodes, void** state)
{
    FEM_TRY
    FEM_CHECK(sn && mesh && K && J && R && F && aggregate_u && work && state, 1, "fem_snes_set_defmod_callbacks: null argument");
    FEM_CHECK(nbcs >= 0 && (nbcs == 0 || (bcnode && bcval)), 2, "fem_snes_set_defmod_callbacks: bad boundary-condition list");
    auto* d = new fem_defmod_callbacks_s;
    d->mesh = mesh; d->K = K; d->J = J; d->F = F; d->agg = aggregate_u; d->utot = work;
    d->nbcs = nbcs;
    d->bcnode.assign(bcnode, bcnode + (size_t)nbcs * (1 + mesh->pdim));
    d->bcval.assign(bcval, bcval + (size_t)nbcs * mesh->pdim);
    d->dt = dt; d->t = t_total; d->rels_mode = rels_mode; d->have_nodes = rels_nodes != nullptr;
    for (int i = 0; i < 4; i++) d->rels_nodes[i] = rels_nodes ? rels_nodes[i] : i;
    snes_set_function(sn, R, defmod_calc_residual, d);
    snes_set_jacobian(sn, J, J, defmod_calc_jacobian, d);
    *state = d;
    FEM_CATCH
}
int fem_snes_calc_defmod_jacobian(void* state, fem_vec du)
{
    FEM_TRY
    FEM_CHECK(state && du, 1, "null argument");
    auto* d = static_cast<fem_defmod_callbacks_s*>(state);
    const int rc = defmod_calc_jacobian(nullptr, du, d->J, d->J, d);
    if (rc) return rc;
    FEM_CATCH
}
int fem_snes_free_defmod_callbacks(void* state)
{
    FEM_TRY
    delete static_cast<fem_defmod_callbacks_s*>(state);
    FEM_CATCH
}
int fem_snes_set_from_options(fem_snes sn, const char* options)
{
    FEM_TRY
    FEM_CHECK(sn, 1, "null handle");
    snes_set_from_options_impl(sn, options);
    FEM_CATCH
}
int fem_snes_solve(fem_snes sn, fem_vec x)
{
    FEM_TRY
    FEM_CHECK(sn && x, 1, "null handle");
    snes_solve_impl(sn, x);
    FEM_CATCH
}
int fem_snes_get_iteration_number(fem_snes sn, int* its)
{
    FEM_TRY
    FEM_CHECK(sn, 1, "null handle");
    snes_get_result(sn, its, nullptr, nullptr, nullptr, nullptr);
    FEM_CATCH
}
int fem_snes_get_converged_reason(fem_snes sn, int* reason)
{
    FEM_TRY
    FEM_CHECK(sn, 1, "null handle");
    snes_get_result(sn, nullptr, reason, nullptr, nullptr, nullptr);
    FEM_CATCH
}
int fem_snes_get_function_norm(fem_snes sn, double* fnorm)
{
    FEM_TRY
    FEM_CHECK(sn, 1, "null handle");
    snes_get_result(sn, nullptr, nullptr, fnorm, nullptr, nullptr);
    FEM_CATCH
}
int fem_snes_get_linear_solve_iterations(fem_snes sn, int* lits)
{
    FEM_TRY
    FEM_CHECK(sn, 1, "null handle");
    snes_get_result(sn, nullptr, nullptr, nullptr, lits, nullptr);
    FEM_CATCH
}
int fem_snes_get_number_function_evals(fem_snes sn, int* nf)
{
    FEM_TRY
    FEM_CHECK(sn, 1, "null handle");
    snes_get_result(sn, nullptr, nullptr, nullptr, nullptr, nf);
    FEM_CATCH
}

// ---- nodal stress -------------------------------------------------------------------------
int fem_recover_nodal_stress_vec(fem_mesh mesh, fem_vec U, fem_vec S)
{
    FEM_TRY
    FEM_CHECK(mesh && U && S, 1, "null handle");
    FEM_CHECK(U->n == (long long)mesh->nnds * mesh->pdim && S->n == (long long)mesh->nnds * mesh->cpdim, 8,
              "fem_recover_nodal_stress: vector size mismatch");
    recover_nodal_stress_dev(mesh, U->d.p, S->d.p);
    FEM_CATCH
}

int fem_recover_nodal_stress(fem_mesh mesh, fem_vec U, double* stress_node)
{
    FEM_TRY
    FEM_CHECK(mesh && U && stress_node, 1, "null argument");
    FEM_CHECK(U->n == (long long)mesh->nnds * mesh->pdim, 8, "fem_recover_nodal_stress: vector size mismatch");
    mesh->ctx->bind();
    DBuf<double> S((size_t)mesh->nnds * mesh->cpdim);
    recover_nodal_stress_dev(mesh, U->d.p, S.p);
    S.download(stress_node, S.n, mesh->ctx->stream);
    FEM_HIP(hipStreamSynchronize(mesh->ctx->stream));
    FEM_CATCH
}

// ---- communicators ------------------------------------------------------------------------
int fem_comm_create_self(fem_ctx ctx, fem_comm* comm)
{
    FEM_TRY
    *comm = comm_create_self_impl(ctx);
    FEM_CATCH
}
int fem_comm_get_unique_id(char unique_id[128])
{
    FEM_TRY
    comm_get_unique_id_impl(unique_id);
    FEM_CATCH
}
int fem_comm_create_rccl(fem_ctx ctx, int nranks, int rank, const char unique_id[128], fem_comm* comm)
{
    FEM_TRY
    *comm = comm_create_rccl_impl(ctx, nranks, rank, unique_id);
    FEM_CATCH
}
int fem_comm_create_local_group(int n, const fem_ctx* ctxs, fem_comm* comms)
{
    FEM_TRY
    comm_create_local_group_impl(n, ctxs, comms);
    FEM_CATCH
}
int fem_comm_create_callbacks(fem_ctx ctx, int nranks, int rank, fem_comm_allreduce_fn allreduce, fem_comm_exchange_fn exchange,
                              void* user, fem_comm* comm)
{
    FEM_TRY
    FEM_CHECK(comm, 1, "fem_comm_create_callbacks: null output");
    *comm = comm_create_callbacks_impl(ctx, nranks, rank, allreduce, exchange, user);
    FEM_CATCH
}
int fem_comm_get_stats(fem_comm c, long long* stats)
{
    FEM_TRY
    FEM_CHECK(c && c->impl && stats, 1, "fem_comm_get_stats: null handle");
    const fem::Comm* m = c->impl;
    stats[0] = m->n_allreduce; stats[1] = m->max_allreduce; stats[2] = m->sum_allreduce;
    stats[3] = m->n_exchange; stats[4] = m->max_exchange; stats[5] = m->sum_exchange;
    FEM_CATCH
}
int fem_comm_destroy(fem_comm c)
{
    FEM_TRY
    if (c) { delete c->impl; delete c; }
    FEM_CATCH
}
int fem_comm_rank(fem_comm c, int* rank, int* size)
{
    FEM_TRY
    FEM_CHECK(c && c->impl, 1, "null communicator");
    if (rank) *rank = c->impl->rank;
    if (size) *size = c->impl->size;
    FEM_CATCH
}

int fem_comm_exchange(fem_comm c, int nneigh, const int* neigh_rank, const long long* off, fem_vec send, fem_vec recv)
{
    FEM_TRY
    FEM_CHECK(c && c->impl && send && recv, 1, "fem_comm_exchange: null handle");
    FEM_CHECK(nneigh >= 0 && (nneigh == 0 || (neigh_rank && off)), 2, "fem_comm_exchange: bad neighbour list");
    std::vector<size_t> o((size_t)nneigh + 1, 0);
    for (int i = 0; i <= nneigh && nneigh > 0; i++) {
        FEM_CHECK(off[i] >= 0 && (i == 0 || off[i] >= off[i - 1]), 2, "fem_comm_exchange: offsets must ascend");
        o[i] = (size_t)off[i];
    }
    FEM_CHECK((long long)o[nneigh] <= send->n && (long long)o[nneigh] <= recv->n, 8, "fem_comm_exchange: vectors too short");
    for (int i = 0; i < nneigh; i++)
        FEM_CHECK(neigh_rank[i] >= 0 && neigh_rank[i] < c->impl->size, 3, "fem_comm_exchange: bad neighbour rank");
    send->ctx->bind();
    c->impl->exchange(nneigh, neigh_rank, o.data(), send->d.p, recv->d.p, send->ctx->stream);
    FEM_HIP(hipStreamSynchronize(send->ctx->stream));
    FEM_CATCH
}

// the same with the sum of red[0 .. nred) over all domains in the same call (Comm::exchange_allreduce: on the RCCL back end the
// all-reduce sits inside the group of the sends and receives -- what `-ksp_fuse_allreduce 1` issues once per CG iteration)
int fem_comm_exchange_allreduce(fem_comm c, int nneigh, const int* neigh_rank, const long long* off, fem_vec send, fem_vec recv,
                                fem_vec red, int nred)
{
    FEM_TRY
    FEM_CHECK(c && c->impl && send && recv && red, 1, "fem_comm_exchange_allreduce: null handle");
    FEM_CHECK(nneigh >= 0 && (nneigh == 0 || (neigh_rank && off)), 2, "fem_comm_exchange_allreduce: bad neighbour list");
    FEM_CHECK(nred >= 1 && nred <= red->n, 8, "fem_comm_exchange_allreduce: bad reduction length");
    std::vector<size_t> o((size_t)nneigh + 1, 0);
    for (int i = 0; i <= nneigh && nneigh > 0; i++) {
        FEM_CHECK(off[i] >= 0 && (i == 0 || off[i] >= off[i - 1]), 2, "fem_comm_exchange_allreduce: offsets must ascend");
        o[i] = (size_t)off[i];
    }
    FEM_CHECK((long long)o[nneigh] <= send->n && (long long)o[nneigh] <= recv->n, 8, "fem_comm_exchange_allreduce: vectors too short");
    for (int i = 0; i < nneigh; i++)
        FEM_CHECK(neigh_rank[i] >= 0 && neigh_rank[i] < c->impl->size, 3, "fem_comm_exchange_allreduce: bad neighbour rank");
    send->ctx->bind();
    c->impl->exchange_allreduce(nneigh, neigh_rank, o.data(), send->d.p, recv->d.p, red->d.p, nred, send->ctx->stream);
    FEM_HIP(hipStreamSynchronize(send->ctx->stream));
    FEM_CATCH
}

}  // extern "C"
