// Host-side pieces of the hot loop for callers that bind the C ABI directly:
// the time-slab partition (reference source/mpi_vector.py:5-38) and the
// preconditioned CG recurrence (reference source/linalg.py:6-42) on device
// vectors, with the operators and the rank reduction supplied as callbacks.
#include <algorithm>
#include <cmath>
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

namespace {

// How the vectors of a Krylov loop are summed: as flat arrays (M = 0; the value then
// depends on how the time axis is split over the ranks, in the last digits) or as
// slabs with one partial per time step (stk_slab_dot: the same value on any number of
// ranks, bit for bit).
struct dot_shape {
    int32_t M, n_loc, ld, N, t_begin;
};

int64_t dot_scratch(const dot_shape &s)
{
    if (s.M == 0) return stk_dot_work_size() + 2;
    return stk_slab_dot_work_size(s.M, s.n_loc) + s.N + (s.N & 1);
}

// Global inner product (mpi_vector.py:205-210): deterministic device reduction, the
// scalar (flat) or the N per-step sums (slab) to the host, sum over the ranks.
int global_dot(void *stream, int64_t n, const dot_shape &s, const double *x, const double *y, double *dot_work,
               stk_allreduce_fn allreduce, void *allreduce_ctx, const char *who, double *value)
{
    hipStream_t st = stk_stream(stream);
    if (s.M == 0) {
        double *dot_out = dot_work + stk_dot_work_size();
        int rc = stk_dot(stream, n, x, y, dot_work, dot_out);
        if (rc) return rc;
        STK_HIP(hipMemcpyAsync(value, dot_out, sizeof(double), hipMemcpyDeviceToHost, st));
        STK_HIP(hipStreamSynchronize(st));
        if (allreduce) {
            rc = allreduce(allreduce_ctx, value, 1);
            if (rc) {
                stk_set_error("%s: allreduce callback failed (%d)", who, rc);
                return rc;
            }
        }
        return 0;
    }
    double *steps = dot_work + stk_slab_dot_work_size(s.M, s.n_loc);
    int rc = stk_slab_dot(stream, s.M, s.n_loc, s.ld, x, y, dot_work, s.N, s.t_begin, steps);
    if (rc) return rc;
    std::vector<double> host(s.N);
    STK_HIP(hipMemcpyAsync(host.data(), steps, sizeof(double) * s.N, hipMemcpyDeviceToHost, st));
    STK_HIP(hipStreamSynchronize(st));
    if (allreduce) {
        rc = allreduce(allreduce_ctx, host.data(), s.N);
        if (rc) {
            stk_set_error("%s: allreduce callback failed (%d)", who, rc);
            return rc;
        }
    }
    *value = stk_sum_steps(host.data(), s.N);
    return 0;
}

int check_slab_shape(const char *who, int32_t M, int32_t n_loc, int32_t ld, int32_t N, int32_t t_begin)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc && (ld & 1) == 0, "%s: bad slab M=%d n_loc=%d ld=%d (ld must be even)",
                who, M, n_loc, ld);
    STK_REQUIRE(t_begin >= 0 && t_begin + n_loc <= N, "%s: time steps [%d, %d) of %d", who, t_begin, t_begin + n_loc, N);
    return 0;
}

int pcg_core(void *stream, int64_t n, const dot_shape &shape, stk_operator_fn T, void *T_ctx, stk_operator_fn P,
             void *P_ctx, stk_allreduce_fn allreduce, void *allreduce_ctx, const double *b, double *w, double eps,
             int32_t kmax, double *work, double *history, int32_t *iters)
{
    STK_REQUIRE(n > 0 && (n & 1) == 0, "stk_pcg_solve: n=%lld must be positive and even", (long long)n);
    STK_REQUIRE(T && P && b && w && work && iters, "stk_pcg_solve: null argument");
    double *r = work, *p = work + n, *q = work + 2 * n, *z = work + 3 * n;
    double *dot_work = work + 4 * n;
    *iters = 0;

    auto dot = [&](const double *x, const double *y, double *value) -> int {
        return global_dot(stream, n, shape, x, y, dot_work, allreduce, allreduce_ctx, "stk_pcg_solve", value);
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

}  // namespace

extern "C" int64_t stk_pcg_work_size(int64_t n) { return 4 * n + dot_scratch(dot_shape{0, 0, 0, 0, 0}); }

extern "C" int stk_pcg_solve(void *stream, int64_t n, stk_operator_fn T, void *T_ctx, stk_operator_fn P, void *P_ctx,
                             stk_allreduce_fn allreduce, void *allreduce_ctx, const double *b, double *w,
                             double eps, int32_t kmax, double *work, double *history, int32_t *iters)
{
    return pcg_core(stream, n, dot_shape{0, 0, 0, 0, 0}, T, T_ctx, P, P_ctx, allreduce, allreduce_ctx, b, w, eps,
                    kmax, work, history, iters);
}

extern "C" int64_t stk_pcg_slab_work_size(int32_t M, int32_t n_loc, int32_t ld, int32_t N)
{
    return 4 * (int64_t)M * ld + dot_scratch(dot_shape{M, n_loc, ld, N, 0});
}

extern "C" int stk_pcg_solve_slab(void *stream, int32_t M, int32_t n_loc, int32_t ld, int32_t N, int32_t t_begin,
                                  stk_operator_fn T, void *T_ctx, stk_operator_fn P, void *P_ctx,
                                  stk_allreduce_fn allreduce, void *allreduce_ctx, const double *b, double *w,
                                  double eps, int32_t kmax, double *work, double *history, int32_t *iters)
{
    if (int rc = check_slab_shape("stk_pcg_solve_slab", M, n_loc, ld, N, t_begin)) return rc;
    return pcg_core(stream, (int64_t)M * ld, dot_shape{M, n_loc, ld, N, t_begin}, T, T_ctx, P, P_ctx, allreduce,
                    allreduce_ctx, b, w, eps, kmax, work, history, iters);
}

// ---- Lanczos estimate of lambda_max / lambda_min of P A (reference
// source/lanczos.py:87-159, Sturm bisection :20-85) ------------------------------
namespace {

double lz_pol(const std::vector<double> &al, const std::vector<double> &be, int k, double x)
{
    double prev = 1.0, cur = al[0] - x;  // lanczos.py:77-85
    for (int l = 1; l <= k; ++l) {
        const double nxt = (al[l] - x) * cur - be[l - 1] * be[l - 1] * prev;
        prev = cur;
        cur = nxt;
    }
    return cur;
}

void lz_bisec(const std::vector<double> &al, const std::vector<double> &be, int k, double tol, double *ymax,
              double *zmin)
{
    // Gershgorin bounds of the leading (k+1) x (k+1) matrix (lanczos.py:22-32)
    double zmax = al[0] + std::fabs(be[0]), ymin = al[0] - std::fabs(be[0]);
    for (int l = 1; l < k; ++l) {
        zmax = std::max(zmax, al[l] + std::fabs(be[l - 1]) + std::fabs(be[l]));
        ymin = std::min(ymin, al[l] - std::fabs(be[l - 1]) - std::fabs(be[l]));
    }
    zmax = std::max(zmax, al[k] + std::fabs(be[k - 1]));
    ymin = std::max(std::min(ymin, al[k] - std::fabs(be[k - 1])), 0.0);
    double pz = lz_pol(al, be, k, zmax);
    while (std::fabs(zmax - *ymax) > tol * std::min(std::fabs(zmax), std::fabs(*ymax))) {
        const double mid = (*ymax + zmax) / 2.0, pm = lz_pol(al, be, k, mid);
        if (std::signbit(pm) != std::signbit(pz))
            *ymax = mid;
        else
            zmax = mid, pz = pm;
    }
    double py = lz_pol(al, be, k, *ymax);
    if (std::signbit(pz) != std::signbit(py) && py != 0.0) *ymax = zmax;
    py = lz_pol(al, be, k, ymin);
    while (std::fabs(*zmin - ymin) > tol * std::min(std::fabs(*zmin), std::fabs(ymin))) {
        const double mid = (ymin + *zmin) / 2.0, pm = lz_pol(al, be, k, mid);
        if (std::signbit(pm) != std::signbit(py))
            *zmin = mid;
        else
            ymin = mid, py = pm;
    }
    pz = lz_pol(al, be, k, *zmin);
    if (std::signbit(pz) != std::signbit(py) && pz != 0.0) *zmin = ymin;
}

int lanczos_core(void *stream, int64_t n, const dot_shape &shape, stk_operator_fn A, void *A_ctx, stk_operator_fn P,
                 void *P_ctx, stk_allreduce_fn allreduce, void *allreduce_ctx, double *w, int32_t max_iterations,
                 double tol, double tol_bisec, double *work, double *alpha_host, double *beta_host, double *lmax,
                 double *lmin, int32_t *iterations, int32_t *converged)
{
    STK_REQUIRE(n > 0 && (n & 1) == 0, "stk_lanczos: n=%lld must be positive and even", (long long)n);
    STK_REQUIRE(A && P && w && work && lmax && lmin && iterations, "stk_lanczos: null argument");
    STK_REQUIRE(max_iterations >= 2, "stk_lanczos: max_iterations=%d too small", max_iterations);
    double *v = work, *u = work + n, *t = work + 2 * n, *wprev = work + 3 * n;
    double *dot_work = work + 4 * n;
    auto dot = [&](const double *x, const double *y, double *value) -> int {
        return global_dot(stream, n, shape, x, y, dot_work, allreduce, allreduce_ctx, "stk_lanczos", value);
    };
    auto apply = [&](stk_operator_fn op, void *ctx, const double *x, double *y, const char *name) -> int {
        const int rc = op(ctx, stream, x, y);
        if (rc) stk_set_error("stk_lanczos: operator %s failed (%d)", name, rc);
        return rc;
    };
#define STK_TRY(expr)        \
    do {                     \
        int rc_ = (expr);    \
        if (rc_) return rc_; \
    } while (0)
    std::vector<double> al(max_iterations, 0.0), be(max_iterations, 0.0);
    if (converged) *converged = 1;
    // normalise the start vector in the A inner product (lanczos.py:109-115)
    STK_TRY(apply(A, A_ctx, w, v, "A"));
    double nrm2 = 0.0;
    STK_TRY(dot(v, w, &nrm2));
    STK_REQUIRE(nrm2 > 0.0, "stk_lanczos: start vector has w.Aw = %g", nrm2);
    const double nrm = std::sqrt(nrm2);
    STK_TRY(stk_axpby(stream, n, 0.0, v, 1.0 / nrm, v));  // v /= nrm
    STK_TRY(stk_axpby(stream, n, 0.0, w, 1.0 / nrm, w));  // w /= nrm
    STK_TRY(apply(P, P_ctx, v, u, "P"));                  // v = P v (into u, then swap roles)
    std::swap(v, u);
    STK_TRY(apply(A, A_ctx, v, u, "A"));
    STK_TRY(dot(u, w, &al[0]));
    double hi = al[0], lo = al[0];
    int k = 0;
    while (true) {  // lanczos.py:121-150
        if (k == max_iterations - 1) {
            if (converged) *converged = 0;
            break;
        }
        STK_TRY(stk_axpby(stream, n, -al[k], w, 1.0, v));  // v -= alpha_k w
        STK_TRY(apply(A, A_ctx, v, u, "A"));
        double b2 = 0.0;
        STK_TRY(dot(u, v, &b2));
        be[k] = std::sqrt(std::max(b2, 0.0));
        if (be[k] == 0.0) {  // invariant subspace: the Ritz values are final
            ++k;
            al[k] = al[k - 1];
            break;
        }
        // w_prev, w = w, v / beta ; v = -beta w_prev
        STK_TRY(stk_axpbyz(stream, n, 1.0, w, 0.0, w, wprev));
        STK_TRY(stk_axpbyz(stream, n, 1.0 / be[k], v, 0.0, v, w));
        STK_TRY(stk_axpbyz(stream, n, -be[k], wprev, 0.0, wprev, v));
        STK_TRY(apply(A, A_ctx, w, u, "A"));
        STK_TRY(apply(P, P_ctx, u, t, "P"));
        STK_TRY(stk_axpby(stream, n, 1.0, t, 1.0, v));  // v += P A w
        ++k;
        STK_TRY(apply(A, A_ctx, v, u, "A"));
        STK_TRY(dot(u, w, &al[k]));
        const double hi_prev = hi, lo_prev = lo;
        lz_bisec(al, be, k, tol_bisec, &hi, &lo);
        if ((hi - hi_prev) < tol * hi_prev && (lo_prev - lo) < tol * lo) break;
    }
#undef STK_TRY
    *iterations = k + 1;
    *lmax = hi;
    *lmin = lo;
    if (alpha_host)
        for (int i = 0; i <= k && i < max_iterations; ++i) alpha_host[i] = al[i];
    if (beta_host)
        for (int i = 0; i < k && i < max_iterations - 1; ++i) beta_host[i] = be[i];
    return 0;
}

}  // namespace

extern "C" int64_t stk_lanczos_work_size(int64_t n) { return 4 * n + dot_scratch(dot_shape{0, 0, 0, 0, 0}); }

extern "C" int stk_lanczos(void *stream, int64_t n, stk_operator_fn A, void *A_ctx, stk_operator_fn P, void *P_ctx,
                           stk_allreduce_fn allreduce, void *allreduce_ctx, double *w, int32_t max_iterations,
                           double tol, double tol_bisec, double *work, double *alpha_host, double *beta_host,
                           double *lmax, double *lmin, int32_t *iterations, int32_t *converged)
{
    return lanczos_core(stream, n, dot_shape{0, 0, 0, 0, 0}, A, A_ctx, P, P_ctx, allreduce, allreduce_ctx, w,
                        max_iterations, tol, tol_bisec, work, alpha_host, beta_host, lmax, lmin, iterations,
                        converged);
}

extern "C" int64_t stk_lanczos_slab_work_size(int32_t M, int32_t n_loc, int32_t ld, int32_t N)
{
    return 4 * (int64_t)M * ld + dot_scratch(dot_shape{M, n_loc, ld, N, 0});
}

extern "C" int stk_lanczos_slab(void *stream, int32_t M, int32_t n_loc, int32_t ld, int32_t N, int32_t t_begin,
                                stk_operator_fn A, void *A_ctx, stk_operator_fn P, void *P_ctx,
                                stk_allreduce_fn allreduce, void *allreduce_ctx, double *w, int32_t max_iterations,
                                double tol, double tol_bisec, double *work, double *alpha_host, double *beta_host,
                                double *lmax, double *lmin, int32_t *iterations, int32_t *converged)
{
    if (int rc = check_slab_shape("stk_lanczos_slab", M, n_loc, ld, N, t_begin)) return rc;
    return lanczos_core(stream, (int64_t)M * ld, dot_shape{M, n_loc, ld, N, t_begin}, A, A_ctx, P, P_ctx, allreduce,
                        allreduce_ctx, w, max_iterations, tol, tol_bisec, work, alpha_host, beta_host, lmax, lmin,
                        iterations, converged);
}
