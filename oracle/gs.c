/* Sequential Gauss-Seidel sweeps on a CSR matrix -- CPU oracle, TEST
 * INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Restates the pure-Python smoother of the reference,
 * source/multigrid.py:83-97:
 *     for i (ascending for PreSmooth, descending for PostSmooth):
 *         ax   = row_i @ u            (whole row, diagonal included)
 *         u[i] += invdiag[i] * (f[i] - ax)
 * repeated `its` times, which is what the reference asks of PETSc MatSOR with
 * omega = 1 and a nonzero initial guess (multigrid.py:116-127).
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC, no -ffast-math so the
 * summation order is the CSR order).
 */
#include <stdint.h>

void oracle_gs_sweeps(int32_t n, const int32_t *indptr, const int32_t *indices,
                      const double *data, const double *invdiag, double *u,
                      const double *f, int32_t its, int32_t backward)
{
    for (int32_t it = 0; it < its; ++it) {
        if (!backward) {
            for (int32_t i = 0; i < n; ++i) {
                double ax = 0.0;
                for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k)
                    ax += data[k] * u[indices[k]];
                u[i] += invdiag[i] * (f[i] - ax);
            }
        } else {
            for (int32_t i = n - 1; i >= 0; --i) {
                double ax = 0.0;
                for (int32_t k = indptr[i]; k < indptr[i + 1]; ++k)
                    ax += data[k] * u[indices[k]];
                u[i] += invdiag[i] * (f[i] - ax);
            }
        }
    }
}

/* k independent right-hand sides stored one per row: U, F are (k, n). */
void oracle_gs_sweeps_batch(int32_t n, const int32_t *indptr,
                            const int32_t *indices, const double *data,
                            const double *invdiag, double *U, const double *F,
                            int32_t k, int32_t its, int32_t backward)
{
    for (int32_t c = 0; c < k; ++c)
        oracle_gs_sweeps(n, indptr, indices, data, invdiag,
                         U + (int64_t)c * n, F + (int64_t)c * n, its, backward);
}
