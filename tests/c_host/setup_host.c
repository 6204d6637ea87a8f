/* A host without Python on the SET-UP part of the C ABI (include/stk.h): the unit square
 * cut into two triangles, refined with stk_tri_refine, its P1 matrices from
 * stk_p1_assemble_2d, a load vector from stk_p1_load_points_2d / stk_p1_load_sum_2d, the
 * union pattern of (M, A) from stk_csr_union_count / _fill and a processing order from
 * stk_tile_order -- all on the host threads of libstk.so, no GPU touched.  Checked in the
 * program against what the mesh dictates.  Plain C99:
 *     gcc -std=c99 setup_host.c -lstk -lm        usage: setup_host <refinements> */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stk.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        if ((call) != 0) {                                                            \
            fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #call, stk_last_error()); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)
#define REQUIRE(cond)                                                                 \
    do {                                                                              \
        if (!(cond)) {                                                                \
            fprintf(stderr, "%s:%d: %s does not hold\n", __FILE__, __LINE__, #cond);  \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

/* Dunavant's degree-4 rule on the triangle (6 points, weights summing to 1) */
static const double RULE_W[6] = {0.223381589678011, 0.223381589678011, 0.223381589678011,
                                 0.109951743655322, 0.109951743655322, 0.109951743655322};
#define A_ 0.445948490915965
#define B_ 0.108103018168070
#define C_ 0.091576213509771
#define D_ 0.816847572980459
static const double RULE_P[18] = {B_, A_, A_, A_, B_, A_, A_, A_, B_, D_, C_, C_, C_, D_, C_, C_, C_, D_};

int main(int argc, char **argv)
{
    const int levels = argc > 1 ? atoi(argv[1]) : 5;
    int64_t nv = 4, nt = 2;
    double *pts = malloc(sizeof(double) * 8);
    int64_t *tris = malloc(sizeof(int64_t) * 6), *cols = malloc(sizeof(int64_t) * 6);
    const double p0[8] = {0, 0, 1, 0, 1, 1, 0, 1};
    /* colour of the edge opposite each local vertex: the diagonal (0, 2) is colour 1 in both */
    const int64_t t0[6] = {0, 1, 2, 0, 2, 3}, c0[6] = {0, 1, 2, 2, 0, 1};
    memcpy(pts, p0, sizeof p0), memcpy(tris, t0, sizeof t0), memcpy(cols, c0, sizeof c0);

    for (int l = 0; l < levels; ++l) {
        const int64_t cap = 3 * nt;
        double *mid = malloc(sizeof(double) * 2 * cap);
        int64_t *par = malloc(sizeof(int64_t) * 2 * cap), *col = malloc(sizeof(int64_t) * cap);
        int64_t *kids = malloc(sizeof(int64_t) * 12 * nt), *kcol = malloc(sizeof(int64_t) * 12 * nt), ne = 0;
        CHECK(stk_tri_refine(nv, nt, pts, tris, cols, cap, mid, par, col, kids, kcol, &ne));
        REQUIRE(ne == nv + nt - 1); /* Euler: a triangulated disc */
        for (int64_t e = 0; e < ne; ++e) { /* a midpoint, its parents in ascending order, colours ascending */
            const int64_t a = par[2 * e], b = par[2 * e + 1];
            REQUIRE(0 <= a && a < b && b < nv);
            REQUIRE(mid[2 * e] == 0.5 * (pts[2 * a] + pts[2 * b]) && mid[2 * e + 1] == 0.5 * (pts[2 * a + 1] + pts[2 * b + 1]));
            REQUIRE(e == 0 || col[e - 1] <= col[e]);
        }
        pts = realloc(pts, sizeof(double) * 2 * (nv + ne));
        memcpy(pts + 2 * nv, mid, sizeof(double) * 2 * ne);
        free(tris), free(cols), free(mid), free(par), free(col);
        tris = kids, cols = kcol, nv += ne, nt *= 4;
    }
    const int64_t side = (int64_t)1 << levels; /* the mesh is the (side + 1)^2 grid */
    REQUIRE(nv == (side + 1) * (side + 1) && nt == 2 * side * side);

    uint8_t *bnd = malloc(nv);
    int64_t n_inner = 0;
    for (int64_t v = 0; v < nv; ++v) {
        const double x = pts[2 * v], y = pts[2 * v + 1];
        bnd[v] = x == 0.0 || x == 1.0 || y == 0.0 || y == 1.0;
        n_inner += !bnd[v];
    }
    REQUIRE(n_inner == (side - 1) * (side - 1));

    /* ---- matrices ---- */
    stk_p1_result *res = NULL;
    CHECK(stk_p1_assemble_2d(nv, nt, pts, tris, bnd, 1e-14, &res));
    int32_t n_free = 0;
    int64_t nnz_a = 0, nnz_m = 0;
    CHECK(stk_p1_result_sizes(res, &n_free, &nnz_a, &nnz_m));
    REQUIRE(n_free == n_inner);
    int32_t *pa = malloc(sizeof(int32_t) * (n_free + 1)), *ia = malloc(sizeof(int32_t) * nnz_a);
    int32_t *pm = malloc(sizeof(int32_t) * (n_free + 1)), *im = malloc(sizeof(int32_t) * nnz_m);
    double *va = malloc(sizeof(double) * nnz_a), *vm = malloc(sizeof(double) * nnz_m);
    CHECK(stk_p1_result_copy(res, 0, pa, ia, va));
    CHECK(stk_p1_result_copy(res, 1, pm, im, vm));
    CHECK(stk_p1_result_free(res));
    /* the three-direction mesh: 5-point stiffness, 7-point mass stencil */
    const int64_t s1 = side - 1;
    REQUIRE(nnz_a == 5 * s1 * s1 - 4 * s1);
    REQUIRE(nnz_m == 7 * s1 * s1 - 4 * s1 - 2 * (s1 + (s1 - 1)));
    double mass = 0.0;
    for (int64_t k = 0; k < nnz_m; ++k) mass += vm[k];

    /* ---- load vector of f = 1: the integrals of the hat functions, summing to the area ---- */
    double *qx = malloc(sizeof(double) * 6 * nt), *qy = malloc(sizeof(double) * 6 * nt), *f = malloc(sizeof(double) * 6 * nt);
    double *load = malloc(sizeof(double) * nv);
    CHECK(stk_p1_load_points_2d(nv, nt, pts, tris, 6, RULE_P, qx, qy));
    for (int64_t q = 0; q < 6 * nt; ++q) {
        REQUIRE(qx[q] > 0.0 && qx[q] < 1.0 && qy[q] > 0.0 && qy[q] < 1.0);
        f[q] = 1.0;
    }
    CHECK(stk_p1_load_sum_2d(nv, nt, pts, tris, 6, RULE_W, RULE_P, f, load));
    double area = 0.0, inner = 0.0;
    for (int64_t v = 0; v < nv; ++v) {
        area += load[v];
        if (!bnd[v]) inner += load[v];
    }
    REQUIRE(fabs(area - 1.0) < 1e-13);
    /* an inner hat function of this mesh has integral h^2 (six triangles of h^2 / 2, a third each) */
    const double h = 1.0 / (double)side;
    REQUIRE(fabs(inner - (double)n_inner * h * h) < 1e-13);
    /* the linear function x is integrated exactly: int x phi_v summed over v = int x = 1/2 */
    CHECK(stk_p1_load_sum_2d(nv, nt, pts, tris, 6, RULE_W, RULE_P, qx, load));
    double first = 0.0;
    for (int64_t v = 0; v < nv; ++v) first += load[v];
    REQUIRE(fabs(first - 0.5) < 1e-13);

    /* ---- union pattern of (M, A): the mass matrix's own ---- */
    const int32_t *ptrs[2] = {pm, pa}, *idxs[2] = {im, ia};
    const double *vals[2] = {vm, va};
    int32_t *pu = malloc(sizeof(int32_t) * (n_free + 1));
    CHECK(stk_csr_union_count(n_free, 2, ptrs, idxs, pu));
    REQUIRE(pu[n_free] == nnz_m);
    int32_t *iu = malloc(sizeof(int32_t) * nnz_m);
    double *um = malloc(sizeof(double) * nnz_m), *ua = malloc(sizeof(double) * nnz_m);
    double *outs[2] = {um, ua};
    CHECK(stk_csr_union_fill(n_free, 2, ptrs, idxs, vals, pu, iu, outs));
    double stiff = 0.0, stiff_ref = 0.0;
    int64_t zeros = 0;
    for (int64_t k = 0; k < nnz_m; ++k) {
        REQUIRE(iu[k] == im[k] && um[k] == vm[k]);
        stiff += ua[k];
        zeros += ua[k] == 0.0;
    }
    for (int64_t k = 0; k < nnz_a; ++k) stiff_ref += va[k];
    REQUIRE(zeros == nnz_m - nnz_a && fabs(stiff - stiff_ref) <= 1e-12 * fabs(stiff_ref));

    /* ---- processing order: a permutation of the free dofs ---- */
    double *free_pts = malloc(sizeof(double) * 2 * n_free);
    int64_t k = 0;
    for (int64_t v = 0; v < nv; ++v)
        if (!bnd[v]) free_pts[2 * k] = pts[2 * v], free_pts[2 * k + 1] = pts[2 * v + 1], ++k;
    const double lo[2] = {h, h};
    int32_t *order = malloc(sizeof(int32_t) * n_free);
    uint8_t *seen = calloc(n_free, 1);
    CHECK(stk_tile_order(n_free, 2, free_pts, lo, 8.0 * h, order));
    for (int64_t r = 0; r < n_free; ++r) {
        REQUIRE(order[r] >= 0 && order[r] < n_free && !seen[order[r]]);
        seen[order[r]] = 1;
    }
    REQUIRE(free_pts[2 * order[0]] == h && free_pts[2 * order[0] + 1] == h); /* the corner comes first */

    printf("setup_host ok: %d refinements, %lld vertices, %lld triangles, %d free dofs, nnz A %lld M %lld, "
           "sum of M %.15f, load of 1 sums to %.15f\n",
           levels, (long long)nv, (long long)nt, n_free, (long long)nnz_a, (long long)nnz_m, mass, area);
    return 0;
}
