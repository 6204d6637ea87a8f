/* A host WITHOUT Python or PyTorch on the C ABI of libstk (include/stk.h): plain C99,
 * compiled with gcc.  What a maintainer of the reference would call from a C / Fortran
 * / Julia driver instead of SumMPI([TridiagKronMatMPI(A_t, M_x), TridiagKronMatMPI(M_t,
 * A_x)]) @ x (reference source/mpi_kron.py:77-90, 204-222):
 *   plan from the CSR arrays the reference holds (stk_kron_plan_create), slab storage
 *   (stk_slab_alloc / _upload / _download: the reference's time-major X_loc[t][i] in,
 *   the same out), the apply (stk_kron_plan_apply), a dot product -- flat (stk_dot)
 *   and as KronVectorMPI.dot takes it since round 6, per time step in a fixed shape
 *   (stk_slab_dot, stk_sum_steps: one value on any partition of the time axis; the slab
 *   is also summed as two slabs of a two-rank partition, the sums must be EQUAL) --
 * checked here against the triple loop of the definition, on the host.
 * Test infrastructure: built and run by tests/test_c_host.py.
 *   usage: kron_host [nx ny n_loc]      prints "kron_host ok ..." and exits 0 on success */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

#include "stk.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        if ((call) != 0) {                                                            \
            fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #call,      \
                    stk_last_error());                                                \
            return 1;                                                                 \
        }                                                                             \
    } while (0)
#define HIP(call)                                                                     \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #call,             \
                    hipGetErrorString(e_));                                           \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

/* 7-point pattern of a P1 triangulation of an nx x ny grid of interior vertices
 * (neighbours W, E, S, N and the two on one diagonal), two matrices on it with a
 * handful of distinct values -- what uniformly refined meshes give -- as CSR. */
static void grid_matrices(int nx, int ny, int32_t **ptr, int32_t **idx, double **mass, double **stiff)
{
    const int M = nx * ny;
    static const int dx[7] = {-1, 0, -1, 0, 1, 0, 1}, dy[7] = {-1, -1, 0, 0, 0, 1, 1};
    *ptr = (int32_t *)malloc(sizeof(int32_t) * (M + 1));
    *idx = (int32_t *)malloc(sizeof(int32_t) * 7 * M);
    *mass = (double *)malloc(sizeof(double) * 7 * M);
    *stiff = (double *)malloc(sizeof(double) * 7 * M);
    int n = 0;
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            (*ptr)[j * nx + i] = n;
            for (int s = 0; s < 7; ++s) { /* ascending column order */
                const int ii = i + dx[s], jj = j + dy[s];
                if (ii < 0 || ii >= nx || jj < 0 || jj >= ny) continue;
                const int diag = (dx[s] == 0 && dy[s] == 0), across = (dx[s] * dy[s] != 0);
                (*idx)[n] = jj * nx + ii;
                (*mass)[n] = diag ? 0.5 : 1.0 / 12.0;
                (*stiff)[n] = diag ? 4.0 : (across ? 0.0 : -1.0);
                ++n;
            }
        }
    (*ptr)[M] = n;
}

int main(int argc, char **argv)
{
    const int nx = argc > 3 ? atoi(argv[1]) : 41, ny = argc > 3 ? atoi(argv[2]) : 37;
    const int n_loc = argc > 3 ? atoi(argv[3]) : 17;
    const int M = nx * ny;
    int32_t *ptr, *idx;
    double *mass, *stiff;
    grid_matrices(nx, ny, &ptr, &idx, &mass, &stiff);
    /* explicit zeros are not entries of a CSR matrix the reference would hold: drop them
       from the stiffness matrix (its own pattern, a subset of the mass matrix's) */
    int32_t *sptr = (int32_t *)malloc(sizeof(int32_t) * (M + 1)), *sidx = (int32_t *)malloc(sizeof(int32_t) * 7 * M);
    double *sval = (double *)malloc(sizeof(double) * 7 * M);
    int sn = 0;
    for (int i = 0; i < M; ++i) {
        sptr[i] = sn;
        for (int e = ptr[i]; e < ptr[i + 1]; ++e)
            if (stiff[e] != 0.0) sidx[sn] = idx[e], sval[sn] = stiff[e], ++sn;
    }
    sptr[M] = sn;

    /* time factors: tridiagonal, [3][n_loc] = sub / main / super diagonal per local row
       (the slicing of TridiagKronIdentityMPI, mpi_kron.py:165-183; one rank: no ghosts) */
    double *tri_host[2];
    for (int k = 0; k < 2; ++k) {
        tri_host[k] = (double *)malloc(sizeof(double) * 3 * n_loc);
        for (int t = 0; t < n_loc; ++t) {
            tri_host[k][t] = t > 0 ? -0.5 - 0.01 * k * t : 0.0;
            tri_host[k][n_loc + t] = 2.0 + 0.1 * k + 0.003 * t;
            tri_host[k][2 * n_loc + t] = t + 1 < n_loc ? -0.25 + 0.02 * k : 0.0;
        }
    }
    double *X = (double *)malloc(sizeof(double) * (size_t)n_loc * M), *Y = (double *)malloc(sizeof(double) * (size_t)n_loc * M);
    unsigned s = 12345u;
    for (size_t q = 0; q < (size_t)n_loc * M; ++q) {
        s = s * 1664525u + 1013904223u;
        X[q] = (double)(s >> 8) / 16777216.0;
    }

    /* ---- the library -------------------------------------------------------------- */
    const int32_t *ptrs[2] = {ptr, sptr}, *idxs[2] = {idx, sidx};
    const double *vals[2] = {mass, sval};
    stk_kron_plan *plan = NULL;
    CHECK(stk_kron_plan_create(M, 2, ptrs, idxs, vals, NULL, &plan));
    int32_t K = 0, codes = 0, packed = 0, rpu = 0;
    int64_t nnz = 0;
    CHECK(stk_kron_plan_info(plan, &K, &codes, &packed, &nnz, &rpu));
    int32_t ld = 0;
    double *x = NULL, *y = NULL, *tri_dev[2], *work = NULL, *dot_dev = NULL;
    CHECK(stk_slab_alloc(M, n_loc, &ld, &x));
    CHECK(stk_slab_alloc(M, n_loc, &ld, &y));
    CHECK(stk_slab_upload(NULL, M, n_loc, ld, X, x));
    stk_kron_pack_term terms[2];
    for (int k = 0; k < 2; ++k) {
        HIP(hipMalloc((void **)&tri_dev[k], sizeof(double) * 3 * n_loc));
        HIP(hipMemcpy(tri_dev[k], tri_host[k], sizeof(double) * 3 * n_loc, hipMemcpyHostToDevice));
        terms[k].tri = tri_dev[k];
        terms[k].mat = k;
    }
    CHECK(stk_kron_plan_apply(plan, NULL, n_loc, ld, 2, terms, x, NULL, NULL, NULL, 0.0, y));
    CHECK(stk_slab_download(NULL, M, n_loc, ld, y, Y));
    HIP(hipMalloc((void **)&work, sizeof(double) * (size_t)stk_dot_work_size()));
    HIP(hipMalloc((void **)&dot_dev, sizeof(double) * 2));
    CHECK(stk_dot(NULL, (int64_t)M * ld, x, y, work, dot_dev));
    double dot = 0.0;
    HIP(hipMemcpy(&dot, dot_dev, sizeof(double), hipMemcpyDeviceToHost));
    /* the same inner product per time step (mpi_vector.py:205-210 without its dependence on
       the number of ranks): as one slab, and as the two slabs [0, h) and [h, n_loc) of a
       two-rank partition whose per-step sums an all-reduce would add entry by entry */
    double *sd_work, *steps_dev, *steps = (double *)malloc(sizeof(double) * 3 * (size_t)n_loc);
    HIP(hipMalloc((void **)&sd_work, sizeof(double) * (size_t)stk_slab_dot_work_size(M, n_loc)));
    HIP(hipMalloc((void **)&steps_dev, sizeof(double) * (size_t)n_loc));
    CHECK(stk_slab_dot(NULL, M, n_loc, ld, x, y, sd_work, n_loc, 0, steps_dev));
    HIP(hipMemcpy(steps, steps_dev, sizeof(double) * (size_t)n_loc, hipMemcpyDeviceToHost));
    const double dot_steps = stk_sum_steps(steps, n_loc);
    int parts_equal = 1;
    if (n_loc >= 2) {
        const int h = n_loc / 2, n_part[2] = {h, n_loc - h}, t0_part[2] = {0, h};
        for (int q = 0; q < n_loc; ++q) steps[n_loc + q] = 0.0;
        for (int part = 0; part < 2; ++part) {
            double *xp, *yp;
            int32_t ldp;
            CHECK(stk_slab_alloc(M, n_part[part], &ldp, &xp));
            CHECK(stk_slab_alloc(M, n_part[part], &ldp, &yp));
            CHECK(stk_slab_upload(NULL, M, n_part[part], ldp, X + (size_t)t0_part[part] * M, xp));
            CHECK(stk_slab_upload(NULL, M, n_part[part], ldp, Y + (size_t)t0_part[part] * M, yp));
            CHECK(stk_slab_dot(NULL, M, n_part[part], ldp, xp, yp, sd_work, n_loc, t0_part[part], steps_dev));
            HIP(hipMemcpy(steps + 2 * n_loc, steps_dev, sizeof(double) * (size_t)n_loc, hipMemcpyDeviceToHost));
            for (int q = 0; q < n_loc; ++q) steps[n_loc + q] += steps[2 * n_loc + q];
            CHECK(stk_slab_free(xp));
            CHECK(stk_slab_free(yp));
        }
        for (int q = 0; q < n_loc; ++q) parts_equal = parts_equal && steps[q] == steps[n_loc + q];
        parts_equal = parts_equal && stk_sum_steps(steps + n_loc, n_loc) == dot_steps;
    }

    /* ---- the definition, on the host ------------------------------------------------ */
    double *Z = (double *)malloc(sizeof(double) * (size_t)n_loc * M);
    double err = 0.0, big = 0.0, dot_ref = 0.0;
    double *want = (double *)calloc((size_t)n_loc * M, sizeof(double));
    for (int k = 0; k < 2; ++k) {
        const int32_t *p = ptrs[k], *c = idxs[k];
        const double *v = vals[k];
        for (int t = 0; t < n_loc; ++t)
            for (int i = 0; i < M; ++i) {
                double z = 0.0;
                for (int e = p[i]; e < p[i + 1]; ++e) z += v[e] * X[(size_t)t * M + c[e]];
                Z[(size_t)t * M + i] = z;
            }
        for (int t = 0; t < n_loc; ++t)
            for (int i = 0; i < M; ++i) {
                double w = tri_host[k][n_loc + t] * Z[(size_t)t * M + i];
                if (t > 0) w += tri_host[k][t] * Z[(size_t)(t - 1) * M + i];
                if (t + 1 < n_loc) w += tri_host[k][2 * n_loc + t] * Z[(size_t)(t + 1) * M + i];
                want[(size_t)t * M + i] += w;
            }
    }
    for (size_t q = 0; q < (size_t)n_loc * M; ++q) {
        const double d = fabs(Y[q] - want[q]);
        if (d > err) err = d;
        if (fabs(want[q]) > big) big = fabs(want[q]);
        dot_ref += X[q] * want[q];
    }
    CHECK(stk_kron_plan_destroy(plan));
    CHECK(stk_slab_free(x));
    CHECK(stk_slab_free(y));
    const double rel = err / big, rel_dot = fabs(dot - dot_ref) / fabs(dot_ref);
    const double rel_steps = fabs(dot_steps - dot_ref) / fabs(dot_ref);
    const int ok = rel < 1e-13 && rel_dot < 1e-12 && rel_steps < 1e-12 && parts_equal;
    printf("kron_host %s: M=%d n_loc=%d ld=%d K=%d codes=%d packed=%d rows_per_unit=%d  max rel err %.2e  dot rel err %.2e  "
           "per-step dot rel err %.2e, two slabs %s\n",
           ok ? "ok" : "FAILED", M, n_loc, ld, K, codes, packed, rpu, rel, rel_dot, rel_steps,
           parts_equal ? "equal" : "DIFFER");
    return ok ? 0 : 1;
}
