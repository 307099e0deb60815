// fem_amd/csrc/elements.cpp -- host construction of the element tables
// (the reference's Initialize, src/m_elems.F90:124-133) that the kernels read
// from __constant__ memory.
#include "elements.h"

#include <cmath>
#include <cstring>
#include <mutex>

namespace fem {

// inverse of a small dense matrix by Gauss-Jordan elimination with partial
// pivoting (the reference's MatInv is LAPACK dgetrf/dgetri, src/m_utils.F90:45-57)
static bool invert_small(int n, const double (&A)[8][8], double (&Ai)[8][8])
{
    double M[8][16];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) { M[i][j] = A[i][j]; M[i][n + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < n; c++) {
        int p = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0) return false;
        if (p != c)
            for (int j = 0; j < 2 * n; j++) std::swap(M[p][j], M[c][j]);
        const double d = 1.0 / M[c][c];
        for (int j = 0; j < 2 * n; j++) M[c][j] *= d;
        for (int r = 0; r < n; r++)
            if (r != c && M[r][c] != 0.0) {
                const double f = M[r][c];
                for (int j = 0; j < 2 * n; j++) M[r][j] -= f * M[c][j];
            }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ai[i][j] = M[i][n + j];
    return true;
}

static void shape(int t, const double* x, double* N)
{
    const double e = x[0], n = x[1], s = x[2];
    switch (t) {
    case FEM_TRI: N[0] = 1.0 - e - n; N[1] = e; N[2] = n; break;
    case FEM_QUA:
        N[0] = 0.25 * (1.0 - e) * (1.0 - n); N[1] = 0.25 * (1.0 + e) * (1.0 - n);
        N[2] = 0.25 * (1.0 + e) * (1.0 + n); N[3] = 0.25 * (1.0 - e) * (1.0 + n);
        break;
    case FEM_TET: N[0] = 1.0 - e - n - s; N[1] = e; N[2] = n; N[3] = s; break;
    case FEM_HEX: {
        const double sg[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1},
                                 {-1, -1, 1},  {1, -1, 1},  {1, 1, 1},  {-1, 1, 1}};
        for (int a = 0; a < 8; a++)
            N[a] = 0.125 * (1.0 + sg[a][0] * e) * (1.0 + sg[a][1] * n) * (1.0 + sg[a][2] * s);
        break;
    }
    case FEM_COH: N[0] = 0.5 * (1.0 - e); N[1] = 0.5 * (1.0 + e); N[2] = N[1]; N[3] = N[0]; break;
    }
}

static void shape_d(int t, const double* x, double (&d)[3][8]) { elem_shape_d(t, x, d); }

int build_elem_table(int type, ElemTable* t)
{
    if (type < FEM_TRI || type > FEM_COH) return 1;
    std::memset(t, 0, sizeof(*t));
    static const int nps[6] = {0, 2, 2, 3, 4, 2}, vtk[6] = {0, 5, 9, 10, 12, 9};
    t->type = type; t->npe = type_npe(type); t->nip = type_nip(type); t->dim = type_dim(type);
    t->nps = nps[type]; t->vtk = vtk[type];
    const double g = std::sqrt(1.0 / 3.0);
    switch (type) {
    case FEM_TRI: t->xi[0][0] = 1.0 / 3.0; t->xi[0][1] = 1.0 / 3.0; t->w[0] = 0.5; break;
    case FEM_QUA: { // order (-,-),(-,+),(+,-),(+,+): src/m_elems.F90:311-314
        const int sx[4] = {-1, -1, 1, 1}, sy[4] = {-1, 1, -1, 1};
        for (int i = 0; i < 4; i++) { t->xi[i][0] = sx[i] * g; t->xi[i][1] = sy[i] * g; t->w[i] = 1.0; }
        break;
    }
    case FEM_TET: t->xi[0][0] = t->xi[0][1] = t->xi[0][2] = 0.25; t->w[0] = 1.0 / 6.0; break;
    case FEM_HEX: { // node order: src/m_elems.F90:500-507
        const int sg[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1},
                              {-1, -1, 1},  {1, -1, 1},  {1, 1, 1},  {-1, 1, 1}};
        for (int i = 0; i < 8; i++) { for (int d = 0; d < 3; d++) t->xi[i][d] = sg[i][d] * g; t->w[i] = 1.0; }
        break;
    }
    case FEM_COH:
        t->xi[0][0] = -0.5773502691896260; t->xi[1][0] = 0.5773502691896260; t->w[0] = t->w[1] = 1.0;
        break;
    }
    for (int gp = 0; gp < t->nip; gp++) {
        shape(type, t->xi[gp], t->N[gp]);
        shape_d(type, t->xi[gp], t->dNxi[gp]);
    }
    t->has_inv = 0;
    if (t->nip == t->npe && type != FEM_COH) {
        double Ai[8][8];
        if (invert_small(t->npe, t->N, Ai)) {
            t->has_inv = 1;
            std::memcpy(t->n2inv, Ai, sizeof(Ai));
        }
    }
    return 0;
}

const ElemTable* host_elem_tables()
{
    static ElemTable tabs[6];
    static std::once_flag once;
    std::call_once(once, [] {
        std::memset(tabs, 0, sizeof(tabs));
        for (int t = FEM_TRI; t <= FEM_COH; t++) build_elem_table(t, &tabs[t]);
    });
    return tabs;
}

}  // namespace fem
