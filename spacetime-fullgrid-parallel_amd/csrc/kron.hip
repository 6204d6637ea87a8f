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
#include <cstring>

#include "stk_common.h"

int stk_kron_ell_set_tuning(const char *key, int32_t value);  // kron_ell.hip
int stk_kron_pack_set_tuning(const char *key, int32_t value);  // kron_pack.hip
int stk_kron_pack_terms_set_tuning(const char *key, int32_t value);  // kron_pack_multi.hip
extern int g_plan_pack_rows;                                   // plan.hip
int stk_rows_ell_set_tuning(const char *key, int32_t value);  // rows_ell.hip
int stk_wavelet_set_tuning(const char *key, int32_t value);   // wavelet.hip
extern int g_mg_gs_diag_free;  // mg_build.hip
extern int g_mg_restrict_one_pass;  // mg.hip
extern int g_mg_coarse_static_fetch;  // mg_coarse.hip
extern int g_mg_fuse_coarse, g_mg_coarse_max_rows, g_mg_fuse_restrict, g_mg_zero_start, g_mg_strip_mb, g_mg_strips_used, g_mg_strip_width;  // mg.hip
extern int g_mg_coarse_pairs, g_mg_coarse_lds, g_mg_coarse_uniform;  // mg_coarse.hip

namespace {

constexpr int MAXE = 16;  // CSR entries per row staged in LDS (longer rows spill to global reads)
constexpr int SE = 17;    // LDS stride of the staged entries (odd: rows land on different banks)

template <int NT>
struct KronArgs {
    const int32_t *indptr;
    const int32_t *indices;
    const int32_t *row_ids;  // output row of CSR row `pos` (NULL: identity)
    const double *vals[NT];
    const double *tri[NT];
    const double *x[NT];
    const double *lo[NT];
    const double *hi[NT];
    double *y;
    double beta;
    int32_t M, n_loc, ld;
    int32_t has_lo, has_hi, any_tri;
    int32_t P;  // pair lanes per row = ceil(n_loc / 2); each owns t = 2p, 2p+1
    int32_t W;  // lanes per row = P + has_lo + has_hi
    int32_t R;  // rows per workgroup
    int32_t nblocks, chunk;
};

// One workgroup = R CSR rows (in the order the CSR lists them, which the host
// chooses for L2 locality), W lanes per row.  Phases:
//   1. the rows' CSR entries are staged in LDS with coalesced loads;
//   2. every lane gathers its 16-byte piece of the time column of each
//      neighbour (ghost lanes: 8 bytes of the ghost row) and accumulates all
//      terms at once;
//   3. the tridiagonal time factors are applied through LDS and y is stored.
// SHARED_IN: every term reads the same input vector (one gather feeds all).
template <int NT, bool SHARED_IN, int BS>
__global__ __launch_bounds__(BS) void kron_sum_kernel(const KronArgs<NT> a)
{
    extern __shared__ double sm[];
    // workgroups b and b+8 share an XCD (and its L2): give every XCD one
    // contiguous run of the row order.  Speed only, never correctness.
    const int vb = (int)(blockIdx.x & 7) * a.chunk + (int)(blockIdx.x >> 3);
    if (vb >= a.nblocks) return;

    const int W = a.W, R = a.R, SW = a.n_loc + 3;
    double *s_w = sm;                                    // [NT][R][SW]
    double *s_val = s_w + (a.any_tri ? NT * R * SW : 0); // [NT][R][SE]
    int32_t *s_idx = reinterpret_cast<int32_t *>(s_val + NT * R * SE);  // [R][SE]

    const int tid = threadIdx.x;
    const int r = tid / W;
    const int l = tid - r * W;
    const int pos = vb * R + r;
    const bool rowok = (r < R) && (pos < a.M);

    int e0 = 0, ne = 0, row = 0;
    if (rowok) {
        e0 = a.indptr[pos];
        ne = a.indptr[pos + 1] - e0;
        row = a.row_ids ? a.row_ids[pos] : pos;
        const int nst = ne < MAXE ? ne : MAXE;
        for (int e = l; e < nst; e += W) {
            s_idx[r * SE + e] = a.indices[e0 + e];
#pragma unroll
            for (int k = 0; k < NT; ++k) s_val[(k * R + r) * SE + e] = a.vals[k][e0 + e];
        }
    }
    __syncthreads();

    const int p = l - a.has_lo;           // pair index; <0: lo ghost, >=P: hi ghost
    const bool is_pair = (p >= 0) && (p < a.P);
    const int t0 = 2 * p;
    double acc0[NT], acc1[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) acc0[k] = acc1[k] = 0.0;

    if (rowok) {
        const int32_t *si = s_idx + r * SE;
        const int nst = ne < MAXE ? ne : MAXE;
        // Gathers are issued in batches of UB independent loads (indices past
        // the end of the row are clamped to its last entry and their products
        // dropped), so a row costs one memory round trip, not one per entry.
        constexpr int UB = 8;
        if (is_pair) {
            if (SHARED_IN) {
                const double *xb = a.x[0] + t0;
                for (int eb = 0; eb < nst; eb += UB) {
                    double2 xv[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int e = min(eb + u, nst - 1);
                        xv[u] = *reinterpret_cast<const double2 *>(xb + (size_t)si[e] * a.ld);
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int e = eb + u;
                        if (e < nst) {
#pragma unroll
                            for (int k = 0; k < NT; ++k) {
                                const double v = s_val[(k * R + r) * SE + e];
                                acc0[k] = fma(v, xv[u].x, acc0[k]);
                                acc1[k] = fma(v, xv[u].y, acc1[k]);
                            }
                        }
                    }
                }
                for (int e = MAXE; e < ne; ++e) {  // rows longer than the LDS stage
                    const double2 xv =
                        *reinterpret_cast<const double2 *>(xb + (size_t)a.indices[e0 + e] * a.ld);
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const double v = a.vals[k][e0 + e];
                        acc0[k] = fma(v, xv.x, acc0[k]);
                        acc1[k] = fma(v, xv.y, acc1[k]);
                    }
                }
            } else {
                constexpr int UN = 4;
                for (int eb = 0; eb < nst; eb += UN) {
                    double2 xv[UN][NT];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int e = min(eb + u, nst - 1);
                        const size_t off = (size_t)si[e] * a.ld + t0;
#pragma unroll
                        for (int k = 0; k < NT; ++k)
                            xv[u][k] = *reinterpret_cast<const double2 *>(a.x[k] + off);
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int e = eb + u;
                        if (e < nst) {
#pragma unroll
                            for (int k = 0; k < NT; ++k) {
                                const double v = s_val[(k * R + r) * SE + e];
                                acc0[k] = fma(v, xv[u][k].x, acc0[k]);
                                acc1[k] = fma(v, xv[u][k].y, acc1[k]);
                            }
                        }
                    }
                }
                for (int e = MAXE; e < ne; ++e) {
                    const size_t off = (size_t)a.indices[e0 + e] * a.ld + t0;
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const double v = a.vals[k][e0 + e];
                        const double2 xv = *reinterpret_cast<const double2 *>(a.x[k] + off);
                        acc0[k] = fma(v, xv.x, acc0[k]);
                        acc1[k] = fma(v, xv.y, acc1[k]);
                    }
                }
            }
        } else {
            // ghost lane: one value of the neighbour rank's boundary time row
            const bool is_lo = p < 0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const double *g = is_lo ? a.lo[k] : a.hi[k];
                if (g == nullptr) continue;
                for (int eb = 0; eb < nst; eb += UB) {
                    double gv[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) gv[u] = g[si[min(eb + u, nst - 1)]];
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        if (eb + u < nst) acc0[k] = fma(s_val[(k * R + r) * SE + eb + u], gv[u], acc0[k]);
                }
                for (int e = MAXE; e < ne; ++e)
                    acc0[k] = fma(a.vals[k][e0 + e], g[a.indices[e0 + e]], acc0[k]);
            }
        }
    }

    double y0 = 0.0, y1 = 0.0;
    if (a.any_tri) {
        // s_w[k][r][q]: q = t + 1, so q = 0 is the lo ghost and q = n_loc + 1 the hi ghost
        if (rowok) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                double *w = s_w + (k * R + r) * SW;
                if (is_pair) {
                    w[t0 + 1] = acc0[k];
                    if (t0 + 1 < a.n_loc) w[t0 + 2] = acc1[k];
                } else {
                    w[p < 0 ? 0 : a.n_loc + 1] = acc0[k];
                }
            }
        }
        __syncthreads();
        if (rowok && is_pair) {
            const bool has1 = t0 + 1 < a.n_loc;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const double *t3 = a.tri[k];
                if (t3 != nullptr) {
                    const double *w = s_w + (k * R + r) * SW + t0 + 1;  // w[0] = value at t0
                    double v0 = t3[a.n_loc + t0] * acc0[k];
                    if (t0 > 0 || a.has_lo) v0 = fma(t3[t0], w[-1], v0);
                    if (has1 || a.has_hi) v0 = fma(t3[2 * a.n_loc + t0], has1 ? acc1[k] : w[1], v0);
                    y0 += v0;
                    if (has1) {
                        double v1 = t3[a.n_loc + t0 + 1] * acc1[k];
                        v1 = fma(t3[t0 + 1], acc0[k], v1);
                        if (t0 + 2 < a.n_loc || a.has_hi) v1 = fma(t3[2 * a.n_loc + t0 + 1], w[2], v1);
                        y1 += v1;
                    }
                } else {
                    y0 += acc0[k];
                    y1 += acc1[k];
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            y0 += acc0[k];
            y1 += acc1[k];
        }
    }
    if (rowok && is_pair) {
        // the second slot of the last pair is padding when n_loc is odd: keep it zero
        if (t0 + 1 >= a.n_loc) y1 = 0.0;
        double2 *yp = reinterpret_cast<double2 *>(a.y + (size_t)row * a.ld + t0);
        if (a.beta != 0.0) {
            const double2 old = *yp;
            y0 = fma(a.beta, old.x, y0);
            if (t0 + 1 < a.n_loc) y1 = fma(a.beta, old.y, y1);
        }
        *yp = make_double2(y0, y1);
    }
}

int g_kron_bs = 0;  // 0 = choose by lane utilisation

template <int NT, bool SHARED_IN>
int launch_kron(hipStream_t st, const KronArgs<NT> &a_in)
{
    KronArgs<NT> a = a_in;
    int best = g_kron_bs;
    if (best != 256 && best != 512 && best != 1024) {
        // rows must not straddle workgroups (LDS time stencil): pick the block
        // size that wastes the fewest lanes, preferring smaller blocks
        const int cand[3] = {256, 512, 1024};
        double best_u = -1.0;
        best = 1024;
        for (int c : cand) {
            if (a.W > c) continue;
            const double u = (double)((c / a.W) * a.W) / c;
            if (u > best_u + 0.04) {
                best_u = u;
                best = c;
            }
        }
    }
    a.R = best / a.W;
    a.nblocks = (a.M + a.R - 1) / a.R;
    a.chunk = (a.nblocks + 7) / 8;
    const unsigned grid = (unsigned)(a.chunk * 8);
    const size_t lds = sizeof(double) * ((a.any_tri ? (size_t)NT * a.R * (a.n_loc + 3) : 0) +
                                          (size_t)NT * a.R * SE) +
                       sizeof(int32_t) * (size_t)a.R * SE + 16;
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
                  const int32_t *indices, const int32_t *row_ids, const stk_kron_term *t, double beta,
                  double *y)
{
    KronArgs<NT> a;
    a.indptr = indptr;
    a.indices = indices;
    a.row_ids = row_ids;
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
        if (t[k].x != t[0].x) shared = false;
    }
    a.P = (n_loc + 1) / 2;
    a.W = a.P + a.has_lo + a.has_hi;
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
        const int row = (int)(idx / ld);
        const int t = (int)(idx - (int64_t)row * ld);
        if (t >= n_loc) {  // padding stays zero
            y[idx] = 0.0;
            continue;
        }
        const int e0 = indptr[row], e1 = indptr[row + 1];
        const double *xt = x + t;
        const double c = (vm != nullptr) ? cm[t] : 0.0;
        double s = 0.0;
        // batches of independent gathers; entries past the row end are clamped
        // to its last entry and dropped (one memory round trip per batch)
        for (int eb = e0; eb < e1; eb += 8) {
            double xv[8], av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(eb + u, e1 - 1);
                xv[u] = xt[(size_t)indices[e] * ld];
                av[u] = (vm != nullptr) ? fma(c, vm[e], ca * va[e]) : ca * va[e];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (eb + u < e1) s = fma(av[u], xv[u], s);
        }
        double out = alpha * s;
        if (beta != 0.0) out = fma(beta, z[idx], out);
        y[idx] = out;
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
        const int i = (int)(idx / ld);
        const int t = (int)(idx - (int64_t)i * ld);
        if (t >= n_loc) {
            y[idx] = 0.0;
            continue;
        }
        const double *xi = x + (size_t)i * ld;
        double acc = add_identity ? xi[t] : 0.0;
        for (int e = t_indptr[t]; e < t_indptr[t + 1]; ++e) {
            const int c = t_cols[e];
            const double v = (c < n_loc) ? xi[c] : recv[(size_t)(c - n_loc) * M + i];
            acc = fma(t_vals[e], v, acc);
        }
        y[idx] = acc;
    }
}

}  // namespace

int g_stk_tuning_epoch = 0;  // captured V-cycle graphs (mg.hip) belong to the tuning state they were recorded under
extern int g_mg_graph, g_mg_graph_replays;
extern int g_mg_band_merge;  // mg_build.hip

extern "C" int stk_set_tuning(const char *key, int32_t value)
{
    STK_REQUIRE(key != nullptr, "stk_set_tuning: null key");
    if (std::strcmp(key, "mg_graph_replays") == 0) {  // tests: reset (0) / require at least `value` replays
        if (value > 0 && g_mg_graph_replays < value) {
            stk_set_error("mg_graph_replays: %d graph launches so far, %d required", g_mg_graph_replays, value);
            return 2;
        }
        if (value == 0) g_mg_graph_replays = 0;
        return 0;
    }
    ++g_stk_tuning_epoch;
    if (std::strcmp(key, "mg_band_merge") == 0) {  // plans built by stk_mg_create_from_csr from now on
        g_mg_band_merge = value;
        return 0;
    }
    if (std::strcmp(key, "mg_graph") == 0) {
        g_mg_graph = value;
        return 0;
    }
    if (std::strcmp(key, "kron_block") == 0) {
        g_kron_bs = value;
        return 0;
    }
    if (stk_kron_ell_set_tuning(key, value) == 0) return 0;
    if (stk_kron_pack_set_tuning(key, value) == 0) return 0;
    if (stk_kron_pack_terms_set_tuning(key, value) == 0) return 0;
    if (stk_rows_ell_set_tuning(key, value) == 0) return 0;
    if (stk_wavelet_set_tuning(key, value) == 0) return 0;
    if (std::strcmp(key, "pack_rows") == 0) {  // plans created from now on: matrix rows per slot row (1 or 2)
        g_plan_pack_rows = value >= 2 ? 2 : 1;
        return 0;
    }
    if (std::strcmp(key, "mg_strip_mb") == 0) {
        g_mg_strip_mb = value;
        return 0;
    }
    if (std::strcmp(key, "mg_strip_width") == 0) {
        g_mg_strip_width = value;
        return 0;
    }
    if (std::strcmp(key, "mg_coarse_static_fetch") == 0) {
        g_mg_coarse_static_fetch = value;
        return 0;
    }
    if (std::strcmp(key, "mg_strips_used") == 0) {  // tests: reset (0) / require at least `value` strip launches
        if (value > 0 && g_mg_strips_used < value) {
            stk_set_error("mg_strips_used: %d strip launches so far, %d required", g_mg_strips_used, value);
            return 2;
        }
        if (value == 0) g_mg_strips_used = 0;
        return 0;
    }
    if (std::strcmp(key, "mg_zero_start") == 0) {
        g_mg_zero_start = value;
        return 0;
    }
    if (std::strcmp(key, "mg_gs_diag_free") == 0) {
        g_mg_gs_diag_free = value;
        return 0;
    }
    if (std::strcmp(key, "mg_fuse_restrict") == 0) {
        g_mg_fuse_restrict = value;
        return 0;
    }
    if (std::strcmp(key, "mg_coarse_max_rows") == 0) {
        g_mg_coarse_max_rows = value;
        return 0;
    }
    if (std::strcmp(key, "mg_coarse_lds") == 0) {
        g_mg_coarse_lds = value;
        return 0;
    }
    if (std::strcmp(key, "mg_restrict_one_pass") == 0) {
        g_mg_restrict_one_pass = value;
        return 0;
    }
    if (std::strcmp(key, "mg_coarse_uniform") == 0) {
        g_mg_coarse_uniform = value;
        return 0;
    }
    if (std::strcmp(key, "mg_coarse_pairs") == 0) {
        g_mg_coarse_pairs = value;
        return 0;
    }
    if (std::strcmp(key, "mg_fuse_coarse") == 0) {
        g_mg_fuse_coarse = value;
        return 0;
    }
    stk_set_error("stk_set_tuning: unknown key '%s'", key);
    return 2;
}

extern "C" int stk_kron_sum_apply(void *stream, int32_t M, int32_t n_loc, int32_t ld, const int32_t *indptr,
                                  const int32_t *indices, const int32_t *row_ids, int32_t n_terms,
                                  const stk_kron_term *t, double beta, double *y)
{
    const stk_timed timed_(STK_OP_KRON, stream);
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc, "stk_kron_sum_apply: bad sizes M=%d n_loc=%d ld=%d", M,
                n_loc, ld);
    STK_REQUIRE((ld & 1) == 0, "stk_kron_sum_apply: ld=%d must be even (16-byte time pairs)", ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= STK_MAX_TERMS, "stk_kron_sum_apply: n_terms=%d not in 1..%d",
                n_terms, STK_MAX_TERMS);
    STK_REQUIRE(indptr && indices && t && y, "stk_kron_sum_apply: null pointer");
    STK_REQUIRE((n_loc + 1) / 2 + 2 <= 1024, "stk_kron_sum_apply: n_loc=%d too large for one workgroup row",
                n_loc);
    STK_REQUIRE(((uintptr_t)y & 15) == 0, "stk_kron_sum_apply: y must be 16-byte aligned");
    for (int k = 0; k < n_terms; ++k) {
        STK_REQUIRE(t[k].vals && t[k].x, "stk_kron_sum_apply: term %d has null vals/x", k);
        STK_REQUIRE(t[k].x != y, "stk_kron_sum_apply: input aliases output");  // mpi_kron.py:190
        STK_REQUIRE(((uintptr_t)t[k].x & 15) == 0, "stk_kron_sum_apply: x must be 16-byte aligned");
    }
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return dispatch_kron<1>(st, M, n_loc, ld, indptr, indices, row_ids, t, beta, y);
        case 2: return dispatch_kron<2>(st, M, n_loc, ld, indptr, indices, row_ids, t, beta, y);
        case 3: return dispatch_kron<3>(st, M, n_loc, ld, indptr, indices, row_ids, t, beta, y);
        default: return dispatch_kron<4>(st, M, n_loc, ld, indptr, indices, row_ids, t, beta, y);
    }
}

extern "C" int stk_csr_spmm(void *stream, int32_t rows, int32_t n_loc, int32_t ld, const int32_t *indptr,
                            const int32_t *indices, const double *vals_a, double ca, const double *vals_m,
                            const double *cm, const double *x, double alpha, double beta, const double *z,
                            double *y)
{
    const stk_timed timed_(STK_OP_SPACE, stream);
    if (rows == 0) return 0;
    STK_REQUIRE(rows > 0 && n_loc > 0 && ld >= n_loc, "stk_csr_spmm: bad sizes");
    STK_REQUIRE(indptr && indices && vals_a && x && y, "stk_csr_spmm: null pointer");
    STK_REQUIRE((vals_m == nullptr) == (cm == nullptr), "stk_csr_spmm: vals_m and cm go together");
    STK_REQUIRE(beta == 0.0 || z, "stk_csr_spmm: beta != 0 needs z");
    STK_REQUIRE(x != y, "stk_csr_spmm: input aliases output");
    const int64_t total = (int64_t)rows * ld;
    hipLaunchKernelGGL(spmm_kernel, dim3(stk_flat_grid(total, SBS)), dim3(SBS), 0, stk_stream(stream), total,
                       n_loc, ld, indptr, indices, vals_a, ca, vals_m, cm, x, alpha, beta, z, y);
    STK_LAUNCH_CHECK();
    return 0;
}

namespace {
// y[i, r] = sum_c T[r, c] x[i, c]: a dense (n_out x n_in) time factor on every
// spatial row.  One thread per output; T and the x row come through the caches.
__global__ __launch_bounds__(256) void time_dense_kernel(int64_t total, int32_t n_in, int32_t ld_in, int32_t n_out,
                                                         int32_t ld_out, const double *__restrict__ T,
                                                         const double *__restrict__ x, double *__restrict__ y)
{
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t i = idx / ld_out;
        const int r = (int)(idx - i * ld_out);
        double s = 0.0;
        if (r < n_out) {
            const double *xi = x + (size_t)i * ld_in;
            const double *Tr = T + (size_t)r * n_in;
            for (int c = 0; c < n_in; ++c) s = fma(Tr[c], xi[c], s);
        }
        y[idx] = s;  // padding columns are written as zero
    }
}
}  // namespace

extern "C" int stk_time_dense_apply(void *stream, int32_t M, int32_t n_in, int32_t ld_in, int32_t n_out,
                                    int32_t ld_out, const double *T, const double *x, double *y)
{
    const stk_timed timed_(STK_OP_TIME, stream);
    STK_REQUIRE(M > 0 && n_in > 0 && n_out > 0 && ld_in >= n_in && ld_out >= n_out,
                "stk_time_dense_apply: bad sizes M=%d n_in=%d ld_in=%d n_out=%d ld_out=%d", M, n_in, ld_in, n_out,
                ld_out);
    STK_REQUIRE(T && x && y && x != y, "stk_time_dense_apply: null or aliased pointer");
    const int64_t total = (int64_t)M * ld_out;
    hipLaunchKernelGGL(time_dense_kernel, dim3(stk_flat_grid(total, 256)), dim3(256), 0, stk_stream(stream), total,
                       n_in, ld_in, n_out, ld_out, T, x, y);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_time_csr_apply(void *stream, int32_t M, int32_t n_loc, int32_t ld, const int32_t *t_indptr,
                                  const int32_t *t_cols, const double *t_vals, const double *x,
                                  const double *recv, int32_t add_identity, double *y)
{
    const stk_timed timed_(STK_OP_TIME, stream);
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc, "stk_time_csr_apply: bad sizes");
    STK_REQUIRE(t_indptr && t_cols && t_vals && x && y, "stk_time_csr_apply: null pointer");
    STK_REQUIRE(x != y, "stk_time_csr_apply: input aliases output");  // mpi_kron.py:296
    const int64_t total = (int64_t)M * ld;
    hipLaunchKernelGGL(time_csr_kernel, dim3(stk_flat_grid(total, SBS)), dim3(SBS), 0, stk_stream(stream),
                       total, M, n_loc, ld, t_indptr, t_cols, t_vals, x, recv, add_identity, y);
    STK_LAUNCH_CHECK();
    return 0;
}
