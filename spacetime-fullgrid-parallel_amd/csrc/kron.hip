// Kronecker-product applies on the space-major slab layout x[i*ld + t].
//
//  kron_sum_kernel : y = beta*y + sum_k (T_k kron X_k) x_k, T_k tridiagonal in
//                    time (or identity), X_k CSR on a shared pattern.
//                    Replaces TridiagKronMatMPI / SumMPI of the reference
//                    (source/mpi_kron.py:77-90, 186-201, 214-219).
//  spmm_kernel     : y = alpha*(I kron A) x + beta*z, A general CSR with
//                    optional per-time-slice values (mpi_kron.py:143-150,
//                    multigrid.py:174-180).
//  time_csr_kernel : y = (A_t kron I) x (+ x) for a small sparse time matrix
//                    with remote rows (mpi_kron.py:285-317).
//
// Work decomposition: one lane per (space row i, time index t).  Consecutive
// lanes walk t, so the gather of a CSR neighbour j is one contiguous run of
// n_loc doubles (coalesced whatever the spatial dof order is), and the CSR
// entries of row i are wave-broadcast loads.  HBM-bound; no MFMA.
#include "stk_common.h"

namespace {

template <int NT>
struct KronArgs {
    const int32_t *indptr;
    const int32_t *indices;
    const double *vals[NT];
    const double *tri[NT];
    const double *x[NT];
    const double *lo[NT];
    const double *hi[NT];
    double *y;
    double beta;
    int32_t M, n_loc, ld;
    int32_t has_lo, has_hi;  // ghost lanes present
    int32_t W;               // lanes per row = n_loc + has_lo + has_hi
    int32_t R;               // rows per block = BS / W
    int32_t any_tri;
};

// SHARED_IN: every term reads the same input vector (one gather feeds all
// terms); otherwise each term gathers from its own input.
template <int NT, bool SHARED_IN, int BS>
__global__ __launch_bounds__(BS) void kron_sum_kernel(const KronArgs<NT> a)
{
    extern __shared__ double sm[];  // [NT][R][W] spatial results, for the time stencil
    const int W = a.W, R = a.R;
    const int tid = threadIdx.x;
    const int r = tid / W;
    const int l = tid - r * W;
    const int tt = l - a.has_lo;  // -1 .. n_loc
    const int row = blockIdx.x * R + r;
    const bool active = (r < R) && (row < a.M);

    double s[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) s[k] = 0.0;

    if (active) {
        // per-lane source: the time column tt of the input, or a ghost row
        const double *src[NT];
        size_t stride;
        if (tt < 0) {
            stride = 1;
#pragma unroll
            for (int k = 0; k < NT; ++k) src[k] = a.lo[k];
        } else if (tt >= a.n_loc) {
            stride = 1;
#pragma unroll
            for (int k = 0; k < NT; ++k) src[k] = a.hi[k];
        } else {
            stride = (size_t)a.ld;
#pragma unroll
            for (int k = 0; k < NT; ++k) src[k] = a.x[k] + tt;
        }
        const int e0 = a.indptr[row], e1 = a.indptr[row + 1];
        if (SHARED_IN) {
            if (src[0] != nullptr) {
#pragma unroll 4
                for (int e = e0; e < e1; ++e) {
                    const double xv = src[0][(size_t)a.indices[e] * stride];
#pragma unroll
                    for (int k = 0; k < NT; ++k) s[k] = fma(a.vals[k][e], xv, s[k]);
                }
            }
        } else {
#pragma unroll 4
            for (int e = e0; e < e1; ++e) {
                const size_t off = (size_t)a.indices[e] * stride;
#pragma unroll
                for (int k = 0; k < NT; ++k)
                    if (src[k] != nullptr) s[k] = fma(a.vals[k][e], src[k][off], s[k]);
            }
        }
    }

    double acc = 0.0;
    if (a.any_tri) {
        if (r < R) {
#pragma unroll
            for (int k = 0; k < NT; ++k) sm[(k * R + r) * W + l] = s[k];
        }
        __syncthreads();
        if (active && tt >= 0 && tt < a.n_loc) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const double *t3 = a.tri[k];
                if (t3 != nullptr) {
                    const double *srow = sm + (k * R + r) * W;
                    double v = t3[a.n_loc + tt] * s[k];
                    if (l > 0) v = fma(t3[tt], srow[l - 1], v);
                    if (l < W - 1) v = fma(t3[2 * a.n_loc + tt], srow[l + 1], v);
                    acc += v;
                } else {
                    acc += s[k];
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) acc += s[k];
    }
    if (active && tt >= 0 && tt < a.n_loc) {
        double *yp = a.y + (size_t)row * a.ld + tt;
        *yp = (a.beta == 0.0) ? acc : fma(a.beta, *yp, acc);
    }
}

template <int NT, bool SHARED_IN>
int launch_kron(hipStream_t st, const KronArgs<NT> &a_in)
{
    KronArgs<NT> a = a_in;
    // pick the block size that wastes the fewest lanes (rows must not straddle
    // blocks because of the LDS time stencil)
    const int cand[3] = {256, 512, 1024};
    int best = 1024;
    double best_u = -1.0;
    for (int c : cand) {
        if (a.W > c) continue;
        double u = (double)((c / a.W) * a.W) / c;
        if (u > best_u + 0.04) {
            best_u = u;
            best = c;
        }
    }
    a.R = best / a.W;
    const unsigned grid = (unsigned)((a.M + a.R - 1) / a.R);
    const size_t lds = a.any_tri ? sizeof(double) * NT * a.R * a.W : 0;
    if (best == 256)
        hipLaunchKernelGGL((kron_sum_kernel<NT, SHARED_IN, 256>), dim3(grid), dim3(256), lds, st, a);
    else if (best == 512)
        hipLaunchKernelGGL((kron_sum_kernel<NT, SHARED_IN, 512>), dim3(grid), dim3(512), lds, st, a);
    else
        hipLaunchKernelGGL((kron_sum_kernel<NT, SHARED_IN, 1024>), dim3(grid), dim3(1024), lds, st, a);
    STK_LAUNCH_CHECK();
    return 0;
}

template <int NT>
int dispatch_kron(hipStream_t st, int32_t M, int32_t n_loc, int32_t ld, const int32_t *indptr,
                  const int32_t *indices, const stk_kron_term *t, double beta, double *y)
{
    KronArgs<NT> a;
    a.indptr = indptr;
    a.indices = indices;
    a.y = y;
    a.beta = beta;
    a.M = M;
    a.n_loc = n_loc;
    a.ld = ld;
    a.has_lo = a.has_hi = a.any_tri = 0;
    bool shared = true;
    for (int k = 0; k < NT; ++k) {
        a.vals[k] = t[k].vals;
        a.tri[k] = t[k].tri;
        a.x[k] = t[k].x;
        a.lo[k] = t[k].x_lo;
        a.hi[k] = t[k].x_hi;
        if (t[k].x_lo) a.has_lo = 1;
        if (t[k].x_hi) a.has_hi = 1;
        if (t[k].tri) a.any_tri = 1;
        if (t[k].x != t[0].x || t[k].x_lo != t[0].x_lo || t[k].x_hi != t[0].x_hi) shared = false;
    }
    a.W = n_loc + a.has_lo + a.has_hi;
    a.R = 0;
    return shared ? launch_kron<NT, true>(st, a) : launch_kron<NT, false>(st, a);
}

// ---------------------------------------------------------------------------
constexpr int SBS = 256;

__global__ __launch_bounds__(SBS) void spmm_kernel(int64_t total, int32_t n_loc, int32_t ld,
                                                   const int32_t *__restrict__ indptr,
                                                   const int32_t *__restrict__ indices,
                                                   const double *__restrict__ va, double ca,
                                                   const double *__restrict__ vm, const double *__restrict__ cm,
                                                   const double *__restrict__ x, double alpha, double beta,
                                                   const double *z, double *y)
{
    const int64_t stride = (int64_t)gridDim.x * SBS;
    for (int64_t idx = (int64_t)blockIdx.x * SBS + threadIdx.x; idx < total; idx += stride) {
        const int row = (int)(idx / n_loc);
        const int t = (int)(idx - (int64_t)row * n_loc);
        const int e0 = indptr[row], e1 = indptr[row + 1];
        const double *xt = x + t;
        double s = 0.0;
        if (vm != nullptr) {
            const double c = cm[t];
#pragma unroll 4
            for (int e = e0; e < e1; ++e)
                s = fma(fma(c, vm[e], ca * va[e]), xt[(size_t)indices[e] * ld], s);
        } else {
#pragma unroll 4
            for (int e = e0; e < e1; ++e) s = fma(ca * va[e], xt[(size_t)indices[e] * ld], s);
        }
        const size_t o = (size_t)row * ld + t;
        double out = alpha * s;
        if (beta != 0.0) out = fma(beta, z[o], out);
        y[o] = out;
    }
}

__global__ __launch_bounds__(SBS) void time_csr_kernel(int64_t total, int32_t M, int32_t n_loc, int32_t ld,
                                                       const int32_t *__restrict__ t_indptr,
                                                       const int32_t *__restrict__ t_cols,
                                                       const double *__restrict__ t_vals,
                                                       const double *__restrict__ x,
                                                       const double *__restrict__ recv, int add_identity,
                                                       double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * SBS;
    for (int64_t idx = (int64_t)blockIdx.x * SBS + threadIdx.x; idx < total; idx += stride) {
        const int i = (int)(idx / n_loc);
        const int t = (int)(idx - (int64_t)i * n_loc);
        const double *xi = x + (size_t)i * ld;
        double acc = add_identity ? xi[t] : 0.0;
        for (int e = t_indptr[t]; e < t_indptr[t + 1]; ++e) {
            const int c = t_cols[e];
            const double v = (c < n_loc) ? xi[c] : recv[(size_t)(c - n_loc) * M + i];
            acc = fma(t_vals[e], v, acc);
        }
        y[(size_t)i * ld + t] = acc;
    }
}

}  // namespace

extern "C" int stk_kron_sum_apply(void *stream, int32_t M, int32_t n_loc, int32_t ld, const int32_t *indptr,
                                  const int32_t *indices, int32_t n_terms, const stk_kron_term *t, double beta,
                                  double *y)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc, "stk_kron_sum_apply: bad sizes M=%d n_loc=%d ld=%d", M,
                n_loc, ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= STK_MAX_TERMS, "stk_kron_sum_apply: n_terms=%d not in 1..%d",
                n_terms, STK_MAX_TERMS);
    STK_REQUIRE(indptr && indices && t && y, "stk_kron_sum_apply: null pointer");
    STK_REQUIRE(n_loc + 2 <= 1024, "stk_kron_sum_apply: n_loc=%d too large for one workgroup row", n_loc);
    for (int k = 0; k < n_terms; ++k) {
        STK_REQUIRE(t[k].vals && t[k].x, "stk_kron_sum_apply: term %d has null vals/x", k);
        STK_REQUIRE(t[k].x != y, "stk_kron_sum_apply: input aliases output");  // mpi_kron.py:190
    }
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return dispatch_kron<1>(st, M, n_loc, ld, indptr, indices, t, beta, y);
        case 2: return dispatch_kron<2>(st, M, n_loc, ld, indptr, indices, t, beta, y);
        case 3: return dispatch_kron<3>(st, M, n_loc, ld, indptr, indices, t, beta, y);
        default: return dispatch_kron<4>(st, M, n_loc, ld, indptr, indices, t, beta, y);
    }
}

extern "C" int stk_csr_spmm(void *stream, int32_t rows, int32_t n_loc, int32_t ld, const int32_t *indptr,
                            const int32_t *indices, const double *vals_a, double ca, const double *vals_m,
                            const double *cm, const double *x, double alpha, double beta, const double *z,
                            double *y)
{
    if (rows == 0) return 0;
    STK_REQUIRE(rows > 0 && n_loc > 0 && ld >= n_loc, "stk_csr_spmm: bad sizes");
    STK_REQUIRE(indptr && indices && vals_a && x && y, "stk_csr_spmm: null pointer");
    STK_REQUIRE((vals_m == nullptr) == (cm == nullptr), "stk_csr_spmm: vals_m and cm go together");
    STK_REQUIRE(beta == 0.0 || z, "stk_csr_spmm: beta != 0 needs z");
    STK_REQUIRE(x != y, "stk_csr_spmm: input aliases output");
    const int64_t total = (int64_t)rows * n_loc;
    hipLaunchKernelGGL(spmm_kernel, dim3(stk_flat_grid(total, SBS)), dim3(SBS), 0, stk_stream(stream), total,
                       n_loc, ld, indptr, indices, vals_a, ca, vals_m, cm, x, alpha, beta, z, y);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_time_csr_apply(void *stream, int32_t M, int32_t n_loc, int32_t ld, const int32_t *t_indptr,
                                  const int32_t *t_cols, const double *t_vals, const double *x,
                                  const double *recv, int32_t add_identity, double *y)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc, "stk_time_csr_apply: bad sizes");
    STK_REQUIRE(t_indptr && t_cols && t_vals && x && y, "stk_time_csr_apply: null pointer");
    STK_REQUIRE(x != y, "stk_time_csr_apply: input aliases output");  // mpi_kron.py:296
    const int64_t total = (int64_t)M * n_loc;
    hipLaunchKernelGGL(time_csr_kernel, dim3(stk_flat_grid(total, SBS)), dim3(SBS), 0, stk_stream(stream),
                       total, M, n_loc, ld, t_indptr, t_cols, t_vals, x, recv, add_identity, y);
    STK_LAUNCH_CHECK();
    return 0;
}
