// Host-side pieces of the hot loop for callers that bind the C ABI directly:
// the time-slab partition (reference source/mpi_vector.py:5-38) and the
// preconditioned CG recurrence (reference source/linalg.py:6-42) on device
// vectors, with the operators and the rank reduction supplied as callbacks.
#include <vector>

#include "stk_common.h"

extern "C" int stk_partition(int32_t N, int32_t size, int32_t rank, int32_t *t_begin, int32_t *t_end,
                             int32_t *counts, int32_t *displs)
{
    STK_REQUIRE(N >= 1 && size >= 1 && rank >= 0 && rank < size, "stk_partition: bad N=%d size=%d rank=%d", N, size,
                rank);
    STK_REQUIRE(N >= size, "stk_partition: more ranks (%d) than time steps (%d)", size, N);
    // mpi_vector.py:17-34: blocks of N / size rows; the remainder goes to the
    // LAST ranks, one extra row each
    const int base = N / size, extra = N % size;
    int start = 0;
    for (int p = 0; p < size; ++p) {
        const int n = base + (p >= size - extra ? 1 : 0);
        if (counts) counts[p] = n;
        if (displs) displs[p] = start;
        if (p == rank) {
            if (t_begin) *t_begin = start;
            if (t_end) *t_end = start + n;
        }
        start += n;
    }
    return 0;
}

extern "C" int64_t stk_pcg_work_size(int64_t n) { return 4 * n + stk_dot_work_size() + 2; }

extern "C" int stk_pcg_solve(void *stream, int64_t n, stk_operator_fn T, void *T_ctx, stk_operator_fn P, void *P_ctx,
                             stk_allreduce_fn allreduce, void *allreduce_ctx, const double *b, double *w,
                             double eps, int32_t kmax, double *work, double *history, int32_t *iters)
{
    STK_REQUIRE(n > 0 && (n & 1) == 0, "stk_pcg_solve: n=%lld must be positive and even", (long long)n);
    STK_REQUIRE(T && P && b && w && work && iters, "stk_pcg_solve: null argument");
    hipStream_t st = stk_stream(stream);
    double *r = work, *p = work + n, *q = work + 2 * n, *z = work + 3 * n;
    double *dot_work = work + 4 * n, *dot_out = dot_work + stk_dot_work_size();
    *iters = 0;

    // global dot product: deterministic device reduction, one scalar to the
    // host, sum over the ranks (mpi_vector.py:205-210)
    auto dot = [&](const double *x, const double *y, double *value) -> int {
        int rc = stk_dot(stream, n, x, y, dot_work, dot_out);
        if (rc) return rc;
        STK_HIP(hipMemcpyAsync(value, dot_out, sizeof(double), hipMemcpyDeviceToHost, st));
        STK_HIP(hipStreamSynchronize(st));
        if (allreduce) {
            rc = allreduce(allreduce_ctx, value, 1);
            if (rc) {
                stk_set_error("stk_pcg_solve: allreduce callback failed (%d)", rc);
                return rc;
            }
        }
        return 0;
    };
    auto apply = [&](stk_operator_fn op, void *ctx, const double *x, double *y, const char *name) -> int {
        const int rc = op(ctx, stream, x, y);
        if (rc) stk_set_error("stk_pcg_solve: operator %s failed (%d)", name, rc);
        return rc;
    };
#define STK_TRY(expr)      \
    do {                   \
        int rc_ = (expr);  \
        if (rc_) return rc_; \
    } while (0)

    double bb = 0.0;
    STK_TRY(dot(b, b, &bb));
    if (bb == 0.0) return 0;  // linalg.py:18
    STK_TRY(apply(T, T_ctx, w, q, "T"));
    STK_TRY(stk_axpbyz(stream, n, 1.0, b, -1.0, q, r));  // r = b - T w
    STK_TRY(apply(P, P_ctx, r, p, "P"));                 // p = P r
    double rho = 0.0;
    STK_TRY(dot(r, p, &rho));
    if (history) history[0] = rho;
    if (rho < eps * eps) return 0;  // linalg.py:24
    for (int k = 1; k < kmax; ++k) {
        ++*iters;
        STK_TRY(apply(T, T_ctx, p, q, "T"));
        double pq = 0.0;
        STK_TRY(dot(p, q, &pq));
        const double step = rho / pq;
        STK_TRY(stk_axpby(stream, n, step, p, 1.0, w));   // w += step p
        STK_TRY(stk_axpby(stream, n, -step, q, 1.0, r));  // r -= step T p
        STK_TRY(apply(P, P_ctx, r, z, "P"));
        const double rho_prev = rho;
        STK_TRY(dot(r, z, &rho));
        if (history) history[k] = rho;
        if (rho < eps * eps) break;  // linalg.py:37
        STK_TRY(stk_axpby(stream, n, 1.0, z, rho / rho_prev, p));  // p = z + beta p
    }
#undef STK_TRY
    return 0;
}
