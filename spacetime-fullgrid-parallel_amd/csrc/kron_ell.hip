// Kronecker-sum apply, ELL form:  y = beta*y + sum_k (T_k kron X_k) x_k.
//
// Same operator as stk_kron_sum_apply (reference source/mpi_kron.py:77-90,
// 186-201, 214-219) on a sliced-ELL copy of the shared sparsity pattern:
// every row owns K entry slots (short rows are padded with a zero value and
// their own column), rows are listed in the order they are processed, and
// entries beyond K of very long rows live in an overflow CSR.
//
// The kernel is persistent: a workgroup walks a strided sequence of row groups
// (R rows each, one lane per pair of time steps).  While it gathers and reduces
// group g it already has the ELL entries of its next group in flight (register
// prefetch), so the only memory round trip on the critical path of a group is
// the gather of the x time columns themselves, and all K gathers of a lane are
// issued back to back.  K is a compile-time constant, so the inner loops carry
// no branches.  Workgroups that share an XCD (blockIdx % 8) take interleaved
// groups of one contiguous chunk of the row order, which keeps their gathers
// in that XCD's L2.  HBM-bound by design: x, y and the matrix arrays are each
// read or written once.
//
// Ghost time rows (a slab with neighbour ranks) are NOT part of this kernel: it
// computes the slab-local part, which needs no halo and can therefore run
// while the exchange is in flight, and kron_ell_ghost_kernel adds
//   y[:, 0] += sum_k sub_k[0] X_k x_lo_k,  y[:, n_loc-1] += sum_k sup_k[n_loc-1] X_k x_hi_k
// afterwards.  (An earlier version gave every row extra "ghost lanes" that
// gathered 8-byte values of the ghost rows inside the main kernel: those
// scattered gathers doubled its time -- 0.39 ms instead of 0.195 ms for the
// 33-step slab of a 2-GPU run.)
#include <cstring>

#include "stk_common.h"

namespace {

#ifndef STK_ELL_BS
#define STK_ELL_BS 512
#endif
constexpr int BS = STK_ELL_BS;

template <int NT>
struct EllArgs {
    const int32_t *ell_idx;     // [M][K] column of every slot
    const int32_t *row_ids;     // output row of ELL row pos (NULL: identity)
    const int32_t *ovf_indptr;  // overflow CSR (NULL: none)
    const int32_t *ovf_indices;
    const double *ell_vals[NT];  // [M][K]
    const double *ovf_vals[NT];
    const double *tri[NT];
    const double *x[NT];
    double *y;
    double beta;
    int32_t M, n_loc, ld;
    int32_t any_tri;
    int32_t P, R;
    int32_t ngroups, chunk;  // groups in total / per XCD
    uint32_t vec_bytes;      // M * ld * 8 (0 when wide)
    int32_t wide;            // slab of 4 GiB or more: 64-bit addressing
};

// K: slots per row (compile time).  NPF: ELL elements each thread prefetches
// per array and group, NPF * BS >= R * K.
// GENERIC = false compiles the overflow path out (no row longer than K): the
// common case runs without its branches.
// WIDE (with GENERIC only): slabs of 4 GiB and more, see stk_slab.
template <int NT, bool SHARED_IN, int K, int NPF, bool GENERIC, bool WIDE>
__global__ __launch_bounds__(BS, K >= 12 ? 4 : 6) void kron_ell_kernel(const EllArgs<NT> a)
{
    static_assert(GENERIC || !WIDE, "the fast path is for slabs below 4 GiB");
    constexpr int KS = (K + 3) & ~3;  // LDS stride of a row's slots (16-byte vectors)
    extern __shared__ double sm[];
    const int W = a.P, R = a.R, SW = a.n_loc + 3;
    double *s_w = sm;                                                    // [NT][R][SW]
    double *s_val = s_w + (a.any_tri ? NT * R * SW : 0);                 // [NT][R][KS]
    uint32_t *s_off = reinterpret_cast<uint32_t *>(s_val + NT * R * KS); // [R][KS] byte offset of the column
    uint32_t *s_row = s_off + R * KS;                                    // [R] byte offset of the output row
    // time-stencil coefficients [NT][3][LT], staged once (16-byte aligned rows)
    const int LT = (a.n_loc + 2) & ~1;
    double *s_tri = reinterpret_cast<double *>(s_row + ((R + 3) & ~3));

    const int tid = threadIdx.x;
    const int r = tid / W;
    const int p = tid - r * W;  // pair of time steps
    const bool is_pair = r < R;
    const int t0 = 2 * p;
    const bool has1 = t0 + 1 < a.n_loc;
    const uint32_t ld_bytes = stk_slab<WIDE>::row_stride(a.ld);  // row stride in offset units
    const uint32_t t0_bytes = (uint32_t)t0 * 8u;
    stk_slab<WIDE> sx[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) sx[k] = stk_slab<WIDE>(a.x[k], a.vec_bytes);
    const stk_slab<WIDE> sy(a.y, a.vec_bytes);

    if (a.any_tri) {
        for (int i = threadIdx.x; i < NT * 3 * LT; i += BS) {
            const int k = i / (3 * LT), rem = i - k * 3 * LT;
            const int d = rem / LT, t = rem - d * LT;
            s_tri[i] = (a.tri[k] != nullptr && t < a.n_loc) ? a.tri[k][d * a.n_loc + t] : 0.0;
        }
    }

    // staging role: element i of the group's flat [rows][K] chunk -> LDS [row][KS]
    int st_lds[NPF];
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        const int i = tid + q * BS;
        st_lds[q] = (i / K) * KS + (i % K);
    }

    // groups of this workgroup: interleaved with the other workgroups of its XCD
    const int xcd = blockIdx.x & 7;
    const int step = gridDim.x >> 3;
    const int gend = min((xcd + 1) * a.chunk, a.ngroups);
    int g = xcd * a.chunk + (int)(blockIdx.x >> 3);

    int32_t pidx[NPF];
    double pval[NT][NPF];
    int32_t prow = 0;
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        pidx[q] = 0;
#pragma unroll
        for (int k = 0; k < NT; ++k) pval[k][q] = 0.0;
    }
    if (g < gend) {
        const int rows = min(R, a.M - g * R);
        const size_t base = (size_t)g * R * K;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) {
                pidx[q] = a.ell_idx[base + i];
#pragma unroll
                for (int k = 0; k < NT; ++k) pval[k][q] = a.ell_vals[k][base + i];
            }
        }
        if (tid < rows) prow = a.row_ids ? a.row_ids[g * R + tid] : g * R + tid;
    }

    for (; g < gend; g += step) {
        const int rows = min(R, a.M - g * R);
        // ---- publish this group's ELL entries -------------------------------
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) {
                s_off[st_lds[q]] = (uint32_t)pidx[q] * ld_bytes;
#pragma unroll
                for (int k = 0; k < NT; ++k) s_val[k * R * KS + st_lds[q]] = pval[k][q];
            }
        }
        if (tid < rows) s_row[tid] = (uint32_t)prow * ld_bytes;
        __syncthreads();

        // ---- prefetch the next group (in flight behind the gathers) ----------
        {
            const int gn = g + step;
            if (gn < gend) {
                const int nrows = min(R, a.M - gn * R);
                const size_t base = (size_t)gn * R * K;
#pragma unroll
                for (int q = 0; q < NPF; ++q) {
                    const int i = tid + q * BS;
                    if (i < nrows * K) {
                        pidx[q] = a.ell_idx[base + i];
#pragma unroll
                        for (int k = 0; k < NT; ++k) pval[k][q] = a.ell_vals[k][base + i];
                    }
                }
                if (tid < nrows) prow = a.row_ids ? a.row_ids[gn * R + tid] : gn * R + tid;
            }
        }

        const bool rowok = r < rows;
        const int pos = g * R + r;
        // read now: s_row is rewritten at the top of the next iteration, which a
        // fast wave reaches while a slow one is still storing
        const uint32_t yo = rowok ? s_row[r] : 0u;
        double acc0[NT], acc1[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) acc0[k] = acc1[k] = 0.0;

        if (rowok && is_pair) {
            const uint32_t *so = s_off + r * KS;
            uint32_t off[K];
#pragma unroll
            for (int u = 0; u < K; ++u) off[u] = so[u];
            if (SHARED_IN) {
                double2 xv[K];
#pragma unroll
                for (int u = 0; u < K; ++u) xv[u] = sx[0].load(off[u], t0_bytes);
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const double *sv = s_val + (k * R + r) * KS;
#pragma unroll
                    for (int u = 0; u < K; ++u) {
                        const double v = sv[u];
                        acc0[k] = fma(v, xv[u].x, acc0[k]);
                        acc1[k] = fma(v, xv[u].y, acc1[k]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const double *sv = s_val + (k * R + r) * KS;
                    double2 xv[K];
#pragma unroll
                    for (int u = 0; u < K; ++u) xv[u] = sx[k].load(off[u], t0_bytes);
#pragma unroll
                    for (int u = 0; u < K; ++u) {
                        const double v = sv[u];
                        acc0[k] = fma(v, xv[u].x, acc0[k]);
                        acc1[k] = fma(v, xv[u].y, acc1[k]);
                    }
                }
            }
            if (GENERIC && a.ovf_indptr != nullptr) {  // entries beyond K of very long rows
                for (int e = a.ovf_indptr[pos]; e < a.ovf_indptr[pos + 1]; ++e) {
                    const size_t o = (size_t)a.ovf_indices[e] * a.ld + t0;
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const double2 xv = *reinterpret_cast<const double2 *>(a.x[k] + o);
                        const double v = a.ovf_vals[k][e];
                        acc0[k] = fma(v, xv.x, acc0[k]);
                        acc1[k] = fma(v, xv.y, acc1[k]);
                    }
                }
            }
        }

        // ---- time stencil through LDS, store ---------------------------------
        double y0 = 0.0, y1 = 0.0;
        if (a.any_tri) {
            // s_w[k][r][q]: q = t + 1
            if (rowok && is_pair) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    double *w = s_w + (k * R + r) * SW;
                    w[t0 + 1] = acc0[k];
                    if (has1) w[t0 + 2] = acc1[k];
                }
            }
            __syncthreads();
            if (rowok && is_pair) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    if (a.tri[k] != nullptr) {
                        const double *w = s_w + (k * R + r) * SW + t0 + 1;  // w[0]: value at t0
                        const double *c = s_tri + k * 3 * LT + t0;
                        const double2 sub = *reinterpret_cast<const double2 *>(c);
                        const double2 dia = *reinterpret_cast<const double2 *>(c + LT);
                        const double2 sup = *reinterpret_cast<const double2 *>(c + 2 * LT);
                        double v0 = dia.x * acc0[k];
                        if (t0 > 0) v0 = fma(sub.x, w[-1], v0);
                        if (has1) v0 = fma(sup.x, acc1[k], v0);
                        y0 += v0;
                        if (has1) {
                            double v1 = dia.y * acc1[k];
                            v1 = fma(sub.y, acc0[k], v1);
                            if (t0 + 2 < a.n_loc) v1 = fma(sup.y, w[2], v1);
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
            __syncthreads();  // the LDS entries are rewritten at the top of the loop
        }
        if (rowok && is_pair) {
            if (!has1) y1 = 0.0;  // padding slot stays zero
            if (a.beta != 0.0) {
                const double2 old = sy.load(yo, t0_bytes);
                y0 = fma(a.beta, old.x, y0);
                if (has1) y1 = fma(a.beta, old.y, y1);
            }
            sy.store(yo, t0_bytes, make_double2(y0, y1));
        }
    }
}

int g_ell_wg_per_cu = 0;
int g_ell_force_wide = 0;     // testing: 64-bit addressing on small slabs
int g_ell_force_generic = 0;  // benchmarking: run the generic kernel even when the fast path applies

template <int NT, bool SHARED_IN, int K, bool GENERIC, bool WIDE = false>
int launch4(hipStream_t st, const EllArgs<NT> &a, unsigned grid, size_t lds)
{
    const int npf = (a.R * K + BS - 1) / BS;
    if (npf <= 1)
        hipLaunchKernelGGL((kron_ell_kernel<NT, SHARED_IN, K, 1, GENERIC, WIDE>), dim3(grid), dim3(BS), lds, st, a);
    else if (npf <= 2)
        hipLaunchKernelGGL((kron_ell_kernel<NT, SHARED_IN, K, 2, GENERIC, WIDE>), dim3(grid), dim3(BS), lds, st, a);
    else if (npf <= 4)
        hipLaunchKernelGGL((kron_ell_kernel<NT, SHARED_IN, K, 4, GENERIC, WIDE>), dim3(grid), dim3(BS), lds, st, a);
    else {
        stk_set_error("stk_kron_ell_apply: %d slots per row with %d lanes per row not supported", K, a.P);
        return 2;
    }
    STK_LAUNCH_CHECK();
    return 0;
}

template <int NT, bool SHARED_IN, int K>
int launch3(hipStream_t st, const EllArgs<NT> &a, unsigned grid, size_t lds)
{
    if (a.wide) return launch4<NT, SHARED_IN, K, true, true>(st, a, grid, lds);
    const bool generic = a.ovf_indptr != nullptr || g_ell_force_generic;
    return generic ? launch4<NT, SHARED_IN, K, true>(st, a, grid, lds)
                   : launch4<NT, SHARED_IN, K, false>(st, a, grid, lds);
}

template <int NT, bool SHARED_IN>
int launch2(hipStream_t st, const EllArgs<NT> &a_in, int K)
{
    EllArgs<NT> a = a_in;
    a.R = BS / a.P;
    if (a.R * K > 4 * BS) a.R = 4 * BS / K;  // at most 4 prefetched entries per thread
    a.ngroups = (a.M + a.R - 1) / a.R;
    a.chunk = (a.ngroups + 7) / 8;
    a.wide = g_ell_force_wide || (int64_t)a.M * a.ld * 8 >= ((int64_t)1 << 32);
    a.vec_bytes = a.wide ? 0u : (uint32_t)((int64_t)a.M * a.ld * 8);
    const int KS = (K + 3) & ~3;
    const size_t lds = sizeof(double) * ((a.any_tri ? (size_t)NT * a.R * (a.n_loc + 3) : 0) +
                                          (size_t)NT * a.R * KS) +
                       sizeof(int32_t) * ((size_t)a.R * KS + a.R + 4) +
                       sizeof(double) * (size_t)NT * 3 * (a.n_loc + 2) + 32;
    const int n_cu = stk_cu_count();
    // wide rows (K >= 12, e.g. the 15-point mass matrix of the cube) get 128 VGPRs: 2 workgroups per CU
    int per_cu = g_ell_wg_per_cu > 0 ? g_ell_wg_per_cu : (K >= 12 ? 2 : 3);
    const int by_lds = (int)(160 * 1024 / (lds + 256));
    if (per_cu > by_lds) per_cu = by_lds > 0 ? by_lds : 1;
    int per_xcd = (n_cu / 8) * per_cu;
    if (per_xcd > a.chunk) per_xcd = a.chunk;
    if (per_xcd < 1) per_xcd = 1;
    const unsigned grid = (unsigned)per_xcd * 8;
    switch (K) {
        case 5: return launch3<NT, SHARED_IN, 5>(st, a, grid, lds);
        case 7: return launch3<NT, SHARED_IN, 7>(st, a, grid, lds);
        case 9: return launch3<NT, SHARED_IN, 9>(st, a, grid, lds);
        case 12: return launch3<NT, SHARED_IN, 12>(st, a, grid, lds);
        case 16: return launch3<NT, SHARED_IN, 16>(st, a, grid, lds);
    }
    stk_set_error("stk_kron_ell_apply: K=%d is not one of 5, 7, 9, 12, 16", K);
    return 2;
}

// ---- first and last local time step of a slab with neighbours ------------------
// The main kernel computed them without the received rows x_lo (time step -1) and
// x_hi (time step n_loc).  They are RECOMPUTED here once the halo is there, in the
// main kernel's order of operations,
//   y[row, t] = sum_k fma(sup_k[t], z_k[t+1], fma(sub_k[t], z_k[t-1], dia_k[t] * z_k[t])),
//   z_k[s] = the slot-ordered fma chain of row `row` of X_k over time column s of x_k,
// and overwrite what the main kernel left (reference mpi_kron.py:165-183, 186-201
// keep the received rows as the first and last row of the ghosted block; the rows
// that need them are finished "after the halo arrived", :199-200).  A slab boundary
// therefore leaves no trace: the apply is bit for bit the one-rank apply wherever
// the time axis is cut.  (Rounds 1-5 ADDED the received rows' share to the main
// kernel's value, which puts the neighbour's term last in the sum -- an order no
// interior time step has.)  One lane per matrix row and side.
template <int NT>
struct GhostArgs {
    const int32_t *ell_idx, *row_ids, *ovf_indptr, *ovf_indices;
    const double *ell_vals[NT];
    const double *ovf_vals[NT];
    const double *tri[NT];
    const double *x[NT];
    const double *lo[NT];
    const double *hi[NT];
    double *y;
    const double *old;  // beta != 0: [2][M] the boundary entries of y from before the main kernel
    double beta;
    int32_t M, n_loc, ld, side0;
};

constexpr int GBSZ = 256;

// old[0][row] = y[row][0], old[1][row] = y[row][n_loc - 1]
__global__ __launch_bounds__(GBSZ) void kron_ell_save_boundary_kernel(int32_t M, int32_t n_loc, int32_t ld,
                                                                      const double *__restrict__ y,
                                                                      double *__restrict__ old)
{
    const int row = blockIdx.x * GBSZ + threadIdx.x;
    if (row >= M) return;
    old[row] = y[(size_t)row * ld];
    old[(size_t)M + row] = y[(size_t)row * ld + n_loc - 1];
}

template <int NT, int K>
__global__ __launch_bounds__(GBSZ) void kron_ell_ghost_kernel(const GhostArgs<NT> a)
{
    const int pos = blockIdx.x * GBSZ + threadIdx.x;
    if (pos >= a.M) return;
    const int side = a.side0 + (int)blockIdx.y;
    const int t = side ? a.n_loc - 1 : 0;
    const size_t e0 = (size_t)pos * K;
    int32_t col[K];
#pragma unroll
    for (int u = 0; u < K; ++u) col[u] = a.ell_idx[e0 + u];
    const int o0 = a.ovf_indptr ? a.ovf_indptr[pos] : 0, o1 = a.ovf_indptr ? a.ovf_indptr[pos + 1] : 0;
    // z = row `pos` of X_k times one time column: `src` + column * `stride`
    auto row_dot = [&](int k, const double *src, size_t stride) {
        double xv[K];
#pragma unroll
        for (int u = 0; u < K; ++u) xv[u] = src[(size_t)col[u] * stride];
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < K; ++u) s = fma(a.ell_vals[k][e0 + u], xv[u], s);
        for (int e = o0; e < o1; ++e) s = fma(a.ovf_vals[k][e], src[(size_t)a.ovf_indices[e] * stride], s);
        return s;
    };
    double yv = 0.0;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double zc = row_dot(k, a.x[k] + t, (size_t)a.ld);
        if (a.tri[k] == nullptr) {
            yv += zc;
            continue;
        }
        double v = a.tri[k][a.n_loc + t] * zc;
        if (t > 0)
            v = fma(a.tri[k][t], row_dot(k, a.x[k] + t - 1, (size_t)a.ld), v);
        else if (a.lo[k] != nullptr)
            v = fma(a.tri[k][t], row_dot(k, a.lo[k], 1), v);
        if (t + 1 < a.n_loc)
            v = fma(a.tri[k][2 * a.n_loc + t], row_dot(k, a.x[k] + t + 1, (size_t)a.ld), v);
        else if (a.hi[k] != nullptr)
            v = fma(a.tri[k][2 * a.n_loc + t], row_dot(k, a.hi[k], 1), v);
        yv += v;
    }
    const int row = a.row_ids ? a.row_ids[pos] : pos;
    if (a.old != nullptr) yv = fma(a.beta, a.old[(size_t)side * a.M + row], yv);
    a.y[(size_t)row * a.ld + t] = yv;
}

template <int NT, int K>
int ghost_launch(hipStream_t st, const GhostArgs<NT> &a, unsigned sides)
{
    hipLaunchKernelGGL((kron_ell_ghost_kernel<NT, K>), dim3((a.M + GBSZ - 1) / GBSZ, sides), dim3(GBSZ), 0, st, a);
    STK_LAUNCH_CHECK();
    return 0;
}

// Does any term couple to a received row?  (lo, hi)
inline void ghost_sides(int n_terms, const stk_kron_ell_term *t, bool *lo, bool *hi)
{
    *lo = *hi = false;
    for (int k = 0; k < n_terms; ++k) {
        if (t[k].tri && t[k].x_lo) *lo = true;
        if (t[k].tri && t[k].x_hi) *hi = true;
    }
}

template <int NT>
int ghost_dispatch(hipStream_t st, const stk_ell_pattern *pat, int32_t n_loc, int32_t ld,
                   const stk_kron_ell_term *t, double *y, double beta = 0.0, const double *old = nullptr)
{
    GhostArgs<NT> a;
    bool lo, hi;
    ghost_sides(NT, t, &lo, &hi);
    if (!lo && !hi) return 0;
    for (int k = 0; k < NT; ++k) {
        a.ell_vals[k] = t[k].ell_vals;
        a.ovf_vals[k] = t[k].ovf_vals;
        a.tri[k] = t[k].tri;
        a.x[k] = t[k].x;
        a.lo[k] = t[k].x_lo;
        a.hi[k] = t[k].x_hi;
    }
    a.ell_idx = pat->ell_idx;
    a.row_ids = pat->row_ids;
    a.ovf_indptr = pat->ovf_indptr;
    a.ovf_indices = pat->ovf_indices;
    a.y = y;
    a.old = old;
    a.beta = beta;
    a.M = pat->M;
    a.n_loc = n_loc;
    a.ld = ld;
    // one step: both received rows meet in it; otherwise a side per received row
    a.side0 = (n_loc == 1 || lo) ? 0 : 1;
    const unsigned sides = (n_loc > 1 && lo && hi) ? 2u : 1u;
    switch (pat->K) {
        case 5: return ghost_launch<NT, 5>(st, a, sides);
        case 7: return ghost_launch<NT, 7>(st, a, sides);
        case 9: return ghost_launch<NT, 9>(st, a, sides);
        case 12: return ghost_launch<NT, 12>(st, a, sides);
        case 16: return ghost_launch<NT, 16>(st, a, sides);
    }
    stk_set_error("stk_kron_ell_apply: K=%d is not one of 5, 7, 9, 12, 16", pat->K);
    return 2;
}

template <int NT>
int dispatch(hipStream_t st, const stk_ell_pattern *pat, int32_t n_loc, int32_t ld, const stk_kron_ell_term *t,
             double beta, double *y)
{
    EllArgs<NT> a;
    a.ell_idx = pat->ell_idx;
    a.row_ids = pat->row_ids;
    a.ovf_indptr = pat->ovf_indptr;
    a.ovf_indices = pat->ovf_indices;
    a.y = y;
    a.beta = beta;
    a.M = pat->M;
    a.n_loc = n_loc;
    a.ld = ld;
    a.any_tri = 0;
    bool shared = true;
    for (int k = 0; k < NT; ++k) {
        a.ell_vals[k] = t[k].ell_vals;
        a.ovf_vals[k] = t[k].ovf_vals;
        a.tri[k] = t[k].tri;
        a.x[k] = t[k].x;
        if (t[k].tri) a.any_tri = 1;
        if (t[k].x != t[0].x) shared = false;
    }
    a.P = (n_loc + 1) / 2;
    bool lo, hi;
    ghost_sides(NT, t, &lo, &hi);
    // beta != 0 on a slab with neighbours: the boundary steps are rewritten after the
    // main kernel, which has by then replaced the old values they scale -- keep a copy
    double *old = nullptr;
    if ((lo || hi) && beta != 0.0) {
        STK_HIP(hipMallocAsync(reinterpret_cast<void **>(&old), sizeof(double) * 2 * (size_t)pat->M, st));
        hipLaunchKernelGGL(kron_ell_save_boundary_kernel, dim3((pat->M + GBSZ - 1) / GBSZ), dim3(GBSZ), 0, st, pat->M,
                           n_loc, ld, y, old);
    }
    int rc = shared ? launch2<NT, true>(st, a, pat->K) : launch2<NT, false>(st, a, pat->K);
    if (rc == 0) rc = ghost_dispatch<NT>(st, pat, n_loc, ld, t, y, beta, old);
    if (old) STK_HIP(hipFreeAsync(old, st));
    return rc;
}

}  // namespace

int stk_kron_ell_set_tuning(const char *key, int32_t value)
{
    if (std::strcmp(key, "ell_force_wide") == 0) {
        g_ell_force_wide = value;
        return 0;
    }
    if (std::strcmp(key, "ell_force_generic") == 0) {
        g_ell_force_generic = value;
        return 0;
    }
    if (std::strcmp(key, "ell_wg_per_cu") == 0) {
        g_ell_wg_per_cu = value;
        return 0;
    }
    return 1;
}

extern "C" int stk_kron_ell_ghost_apply(void *stream, const stk_ell_pattern *pat, int32_t n_loc, int32_t ld,
                                        int32_t n_terms, const stk_kron_ell_term *t, double *y)
{
    const stk_timed timed_(STK_OP_KRON, stream);
    STK_REQUIRE(pat && t && y, "stk_kron_ell_ghost_apply: null pointer");
    STK_REQUIRE(pat->M > 0 && pat->K >= 1 && pat->ell_idx, "stk_kron_ell_ghost_apply: bad pattern");
    STK_REQUIRE(n_loc > 0 && ld >= n_loc, "stk_kron_ell_ghost_apply: bad sizes n_loc=%d ld=%d", n_loc, ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_ell_ghost_apply: n_terms=%d not in 1..3", n_terms);
    for (int k = 0; k < n_terms; ++k) {
        STK_REQUIRE(t[k].ell_vals && t[k].x && t[k].x != y, "stk_kron_ell_ghost_apply: term %d has null vals / x", k);
        STK_REQUIRE(pat->ovf_indptr == nullptr || t[k].ovf_vals, "stk_kron_ell_ghost_apply: term %d lacks ovf_vals",
                    k);
    }
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return ghost_dispatch<1>(st, pat, n_loc, ld, t, y);
        case 2: return ghost_dispatch<2>(st, pat, n_loc, ld, t, y);
        default: return ghost_dispatch<3>(st, pat, n_loc, ld, t, y);
    }
}

extern "C" int stk_kron_ell_apply(void *stream, const stk_ell_pattern *pat, int32_t n_loc, int32_t ld,
                                  int32_t n_terms, const stk_kron_ell_term *t, double beta, double *y)
{
    const stk_timed timed_(STK_OP_KRON, stream);
    STK_REQUIRE(pat && t && y, "stk_kron_ell_apply: null pointer");
    STK_REQUIRE(pat->M > 0 && pat->K >= 1, "stk_kron_ell_apply: bad pattern M=%d K=%d", pat->M, pat->K);
    STK_REQUIRE(pat->ell_idx, "stk_kron_ell_apply: pattern has no ell_idx");
    STK_REQUIRE((pat->ovf_indptr == nullptr) == (pat->ovf_indices == nullptr),
                "stk_kron_ell_apply: overflow arrays go together");
    STK_REQUIRE(n_loc > 0 && ld >= n_loc && (ld & 1) == 0,
                "stk_kron_ell_apply: bad sizes n_loc=%d ld=%d (ld must be even)", n_loc, ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_ell_apply: n_terms=%d not in 1..3", n_terms);
    STK_REQUIRE((n_loc + 1) / 2 + 2 <= BS, "stk_kron_ell_apply: n_loc=%d too large", n_loc);
    STK_REQUIRE((int64_t)pat->M * ld * 8 < ((int64_t)1 << 36),
                "stk_kron_ell_apply: slab of %lld bytes exceeds 64 GiB; use stk_kron_sum_apply",
                (long long)pat->M * ld * 8);
    STK_REQUIRE(((uintptr_t)y & 15) == 0, "stk_kron_ell_apply: y must be 16-byte aligned");
    for (int k = 0; k < n_terms; ++k) {
        STK_REQUIRE(t[k].ell_vals && t[k].x, "stk_kron_ell_apply: term %d has null vals/x", k);
        STK_REQUIRE(t[k].x != y, "stk_kron_ell_apply: input aliases output");
        STK_REQUIRE(((uintptr_t)t[k].x & 15) == 0, "stk_kron_ell_apply: x must be 16-byte aligned");
        STK_REQUIRE(pat->ovf_indptr == nullptr || t[k].ovf_vals, "stk_kron_ell_apply: term %d lacks ovf_vals",
                    k);
    }
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return dispatch<1>(st, pat, n_loc, ld, t, beta, y);
        case 2: return dispatch<2>(st, pat, n_loc, ld, t, beta, y);
        default: return dispatch<3>(st, pat, n_loc, ld, t, beta, y);
    }
}
