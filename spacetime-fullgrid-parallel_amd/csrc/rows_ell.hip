// Row-gather engine on sliced-ELL matrices for the multigrid building blocks:
//
//   SPMM : y = alpha * A(t) x + beta * z          (residual, restriction,
//                                                  prolongation-correction,
//                                                  I kron A applies)
//   GS   : u_i += (f_i - sum_j a_ij(t) u_j) / a_ii(t)   for the rows of one
//          dependency group of a Gauss-Seidel sweep (reference
//          source/multigrid.py:83-97, 116-127; see mg.hip for the schedule)
//
// with a(t) = ca * va + cm[t] * vm (vm optional).  Same structure as
// kron_ell.hip: persistent workgroups, one lane per pair of time steps,
// compile-time slot count K, all K gathers of a lane issued back to back
// through a buffer descriptor, ELL entries of the next row group prefetched
// into registers and handed over through double-buffered LDS (one barrier per
// group).  What a pass costs is its wave-level VMEM instructions: K gathers, the
// right-hand side and the store per lane.  Gauss-Seidel copies are normally
// DIAGONAL-FREE (stk_ell_rows.diag_free): the slots hold the off-diagonal entries
// and u_i = (f_i - sum_{j != i} a_ij u_j) / a_ii needs no read of u_i at all -- one
// gather and one row of L2-miss traffic less per row (the launches are bound by
// exactly that traffic).  With the diagonal among the
// slots (diag_free = 0) the row's own value is taken from that gather.
#include <cstring>

#include "stk_common.h"

namespace {

constexpr int BS = 512;
enum { MODE_SPMM = 0, MODE_GS = 1 };

struct RowsArgs {
    const int32_t *idx;      // [n_pos][K]
    const double *va, *vm;   // [n_pos][K]
    const int32_t *row_ids;  // [n_pos] or NULL
    const double *dia_a, *dia_m;  // [n_pos] (GS)
    const double *cm;        // [n_loc] or NULL
    const double *x;         // gather source slab (GS: u)
    const double *z;         // SPMM: beta operand; GS: right-hand side f
    double *y;               // output slab (GS: u)
    double ca, alpha, beta;
    int32_t pos_begin, pos_end;
    int32_t n_loc, ld, P, R;
    int32_t ngroups, chunk;
    int32_t reverse;  // walk the row groups of every XCD chunk backwards
    int32_t zero_own;  // GS: the rows' own values are zero (first sweep from u = 0)
    uint32_t x_bytes, y_bytes;
    int32_t wide;  // a slab of 4 GiB or more: 64-bit addressing
    int32_t nt_store;  // y is stored with the non-temporal hint (it is not read again by this launch)
};

// The output row of an SPMM pass is written once and not gathered by the same launch:
// stored plainly it occupies the XCD's L2 next to the rows of x that neighbouring
// workgroups are about to gather again (PMC of round 4: I kron A_x fetched its input
// 1.75 times; kron_pack.hip, which stores y with the hint, 1.05 times).
template <bool WIDE>
__device__ inline void store_nt(const stk_slab<WIDE> &s, uint32_t row_off, uint32_t t_bytes, double2 v)
{
    if constexpr (WIDE) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d o;
        o.x = v.x, o.y = v.y;
        __builtin_nontemporal_store(
            o, reinterpret_cast<v2d *>(const_cast<char *>(s.base) + (((size_t)row_off) << 4) + t_bytes));
    } else {
        stk_v4i w;
        __builtin_memcpy(&w, &v, 16);
        __builtin_amdgcn_raw_buffer_store_b128(w, s.rs, row_off + t_bytes, 0, 2);  // aux bit 1: nt
    }
}

template <int MODE, int K, int NPF, bool HAS_M, bool WIDE>
__global__ __launch_bounds__(BS, K >= 12 ? 4 : 6) void rows_ell_kernel(const RowsArgs a)
{
    constexpr int KS = (K + 3) & ~3;
    extern __shared__ double sm[];
    const int R = a.R, W = a.P;
    // two LDS buffers of {va[R][KS], vm[R][KS], off[R][KS], row[R], dia_a[R], dia_m[R]}
    const int buf_doubles = ((HAS_M ? 2 : 1) * R * KS + 2 * R + (R * KS + R + 1) / 2 + 2) & ~1;
    const int tid = threadIdx.x;
    const int r = tid / W;
    const int p = tid - r * W;
    const bool lane_ok = r < R;
    const int t0 = 2 * p;
    const bool has1 = t0 + 1 < a.n_loc;
    const uint32_t ld_bytes = stk_slab<WIDE>::row_stride(a.ld);  // row stride in offset units
    const uint32_t t0_bytes = (uint32_t)t0 * 8u;
    const stk_slab<WIDE> sx(a.x, a.x_bytes), sy(a.y, a.y_bytes), sz(a.z, a.y_bytes);

    double cm0 = 0.0, cm1 = 0.0;
    if (HAS_M && lane_ok) {
        cm0 = a.cm[t0];
        if (has1) cm1 = a.cm[t0 + 1];
    }

    int st_lds[NPF];
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        const int i = tid + q * BS;
        st_lds[q] = (i / K) * KS + (i % K);
    }

    const int xcd = blockIdx.x & 7;
    const int step = gridDim.x >> 3;
    const int gend = min((xcd + 1) * a.chunk, a.ngroups);
    int g = xcd * a.chunk + (int)(blockIdx.x >> 3);

    int32_t pidx[NPF];
    double pva[NPF], pvm[NPF];
    int32_t prow = 0;
    double pda = 1.0, pdm = 0.0;
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        pidx[q] = 0;
        pva[q] = pvm[q] = 0.0;
    }
    // Consecutive launches alternate the direction in which they walk the rows:
    // a launch then starts on the data the previous one touched last, which is
    // still in the 256 MB Infinity Cache.
    auto load_group = [&](int gq) {
        const int gg = a.reverse ? (a.ngroups - 1 - gq) : gq;
        const int first = a.pos_begin + gg * R;
        const int rows = min(R, a.pos_end - first);
        const size_t base = (size_t)first * K;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) {
                pidx[q] = a.idx[base + i];
                pva[q] = a.va[base + i];
                if (HAS_M) pvm[q] = a.vm[base + i];
            }
        }
        if (tid < rows) {
            prow = a.row_ids ? a.row_ids[first + tid] : first + tid;
            if (MODE == MODE_GS) {
                pda = a.dia_a[first + tid];
                if (HAS_M) pdm = a.dia_m[first + tid];
            }
        }
    };
    if (g < gend) load_group(g);

    int flip = 0;
    for (; g < gend; g += step, flip ^= 1) {
        double *b_va = sm + flip * buf_doubles;
        double *b_vm = b_va + R * KS;
        double *b_da = b_va + (HAS_M ? 2 : 1) * R * KS;
        double *b_dm = b_da + R;
        uint32_t *b_off = reinterpret_cast<uint32_t *>(b_dm + R);
        uint32_t *b_row = b_off + R * KS;

        const int first = a.pos_begin + (a.reverse ? (a.ngroups - 1 - g) : g) * R;
        const int rows = min(R, a.pos_end - first);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) {
                b_off[st_lds[q]] = (uint32_t)pidx[q] * ld_bytes;
                b_va[st_lds[q]] = a.ca * pva[q];
                if (HAS_M) b_vm[st_lds[q]] = pvm[q];
            }
        }
        if (tid < rows) {
            b_row[tid] = (uint32_t)prow * ld_bytes;
            if (MODE == MODE_GS) {
                b_da[tid] = a.ca * pda;
                if (HAS_M) b_dm[tid] = pdm;
            }
        }
        __syncthreads();
        if (g + step < gend) load_group(g + step);  // in flight behind the gathers

        if (lane_ok && r < rows) {
            const uint32_t *so = b_off + r * KS;
            const double *sva = b_va + r * KS;
            const double *svm = b_vm + r * KS;
            const uint32_t yo = b_row[r];
            double2 xv[K];
#pragma unroll
            for (int u = 0; u < K; ++u) xv[u] = sx.load(so[u], t0_bytes);
            double2 zv = make_double2(0.0, 0.0), own = make_double2(0.0, 0.0);
            if (MODE == MODE_GS) {
                zv = sz.load(yo, t0_bytes);
                // the row's own value is one of the K gathers (every row of a
                // Gauss-Seidel matrix has its diagonal entry): no extra load
                if (!a.zero_own) {
#pragma unroll
                    for (int u = 0; u < K; ++u)
                        if (so[u] == yo) own = xv[u];
                }
            } else if (a.beta != 0.0) {
                zv = sz.load(yo, t0_bytes);
            }
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int u = 0; u < K; ++u) {
                double v0 = sva[u], v1 = v0;
                if (HAS_M) {
                    const double m = svm[u];
                    v0 = fma(cm0, m, v0);
                    v1 = fma(cm1, m, v1);
                }
                s0 = fma(v0, xv[u].x, s0);
                s1 = fma(v1, xv[u].y, s1);
            }
            double o0, o1;
            if (MODE == MODE_GS) {
                double d0 = b_da[r], d1 = d0;
                if (HAS_M) {
                    const double m = b_dm[r];
                    d0 = fma(cm0, m, d0);
                    d1 = fma(cm1, m, d1);
                }
                o0 = own.x + (1.0 / d0) * (zv.x - s0);
                o1 = own.y + (1.0 / d1) * (zv.y - s1);
            } else {
                o0 = a.alpha * s0;
                o1 = a.alpha * s1;
                if (a.beta != 0.0) {
                    o0 = fma(a.beta, zv.x, o0);
                    o1 = fma(a.beta, zv.y, o1);
                }
            }
            if (!has1) o1 = 0.0;  // padding slot stays zero
            if (a.nt_store)
                store_nt<WIDE>(sy, yo, t0_bytes, make_double2(o0, o1));
            else
                sy.store(yo, t0_bytes, make_double2(o0, o1));
        }
    }
}

// ---- the restricted residual in ONE pass: y = A(t) x - B x2 ------------------------
// d_c = (R A_j) u_j - R f_j (mg.hip: the fine residual is never formed) took two passes
// of the engine above -- d_c = R f_j, then d_c = (R A_j) u_j - d_c -- i.e. d_c written,
// read and written again, and a launch more on every level visit.  Here a row gathers
// its K entries of the first matrix from x and accumulates s (as the second pass did),
// then its K2 entries of the second matrix from x2 and accumulates s2 (as the first
// pass did, values unscaled), and stores fma(-1, s2, s): the very roundings of the two
// passes.  Same skeleton as rows_ell_kernel; both matrices list the same rows in the
// same order.
struct Rows2Args {
    RowsArgs a;              // first matrix, x, y as in the engine (alpha = 1, beta unused)
    const int32_t *idx2;     // [n_pos][K2]
    const double *va2;       // [n_pos][K2]
    const double *x2;
    uint32_t x2_bytes;
};

template <int K, int K2, int NPF, bool HAS_M, bool WIDE>
__global__ __launch_bounds__(BS, 4) void rows_ell2_kernel(const Rows2Args b)
{
    const RowsArgs &a = b.a;
    constexpr int KS = (K + 3) & ~3, KS2 = (K2 + 3) & ~3;
    extern __shared__ double sm[];
    const int R = a.R, W = a.P;
    // two LDS buffers of {va[R][KS], vm[R][KS], va2[R][KS2], off[R][KS], off2[R][KS2], row[R]}
    const int val_doubles = (HAS_M ? 2 : 1) * R * KS + R * KS2;
    const int buf_doubles = (val_doubles + (R * KS + R * KS2 + R + 1) / 2 + 2) & ~1;
    const int tid = threadIdx.x;
    const int r = tid / W;
    const int p = tid - r * W;
    const bool lane_ok = r < R;
    const int t0 = 2 * p;
    const bool has1 = t0 + 1 < a.n_loc;
    const uint32_t ld_bytes = stk_slab<WIDE>::row_stride(a.ld);
    const uint32_t t0_bytes = (uint32_t)t0 * 8u;
    const stk_slab<WIDE> sx(a.x, a.x_bytes), sx2(b.x2, b.x2_bytes), sy(a.y, a.y_bytes);

    double cm0 = 0.0, cm1 = 0.0;
    if (HAS_M && lane_ok) {
        cm0 = a.cm[t0];
        if (has1) cm1 = a.cm[t0 + 1];
    }
    const int xcd = blockIdx.x & 7;
    const int step = gridDim.x >> 3;
    const int gend = min((xcd + 1) * a.chunk, a.ngroups);
    int g = xcd * a.chunk + (int)(blockIdx.x >> 3);

    int32_t pidx[NPF], pidx2[NPF];
    double pva[NPF], pvm[NPF], pva2[NPF];
    int32_t prow = 0;
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        pidx[q] = pidx2[q] = 0;
        pva[q] = pvm[q] = pva2[q] = 0.0;
    }
    auto load_group = [&](int gq) {
        const int gg = a.reverse ? (a.ngroups - 1 - gq) : gq;
        const int first = a.pos_begin + gg * R;
        const int rows = min(R, a.pos_end - first);
        const size_t base = (size_t)first * K, base2 = (size_t)first * K2;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) {
                pidx[q] = a.idx[base + i];
                pva[q] = a.va[base + i];
                if (HAS_M) pvm[q] = a.vm[base + i];
            }
            if (i < rows * K2) {
                pidx2[q] = b.idx2[base2 + i];
                pva2[q] = b.va2[base2 + i];
            }
        }
        if (tid < rows) prow = a.row_ids ? a.row_ids[first + tid] : first + tid;
    };
    if (g < gend) load_group(g);

    int flip = 0;
    for (; g < gend; g += step, flip ^= 1) {
        double *b_va = sm + flip * buf_doubles;
        double *b_vm = b_va + R * KS;
        double *b_va2 = b_va + (HAS_M ? 2 : 1) * R * KS;
        uint32_t *b_off = reinterpret_cast<uint32_t *>(b_va2 + R * KS2);
        uint32_t *b_off2 = b_off + R * KS;
        uint32_t *b_row = b_off2 + R * KS2;

        const int first = a.pos_begin + (a.reverse ? (a.ngroups - 1 - g) : g) * R;
        const int rows = min(R, a.pos_end - first);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) {
                const int at = (i / K) * KS + (i % K);
                b_off[at] = (uint32_t)pidx[q] * ld_bytes;
                b_va[at] = a.ca * pva[q];
                if (HAS_M) b_vm[at] = pvm[q];
            }
            if (i < rows * K2) {
                const int at = (i / K2) * KS2 + (i % K2);
                b_off2[at] = (uint32_t)pidx2[q] * ld_bytes;
                b_va2[at] = pva2[q];  // (1.0 * va: what the separate pass multiplied with)
            }
        }
        if (tid < rows) b_row[tid] = (uint32_t)prow * ld_bytes;
        __syncthreads();
        if (g + step < gend) load_group(g + step);  // in flight behind the gathers

        if (lane_ok && r < rows) {
            const uint32_t yo = b_row[r];
            double s0 = 0.0, s1 = 0.0;
            {
                const uint32_t *so = b_off + r * KS;
                const double *sva = b_va + r * KS;
                const double *svm = b_vm + r * KS;
                double2 xv[K];
#pragma unroll
                for (int u = 0; u < K; ++u) xv[u] = sx.load(so[u], t0_bytes);
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    double v0 = sva[u], v1 = v0;
                    if (HAS_M) {
                        const double m = svm[u];
                        v0 = fma(cm0, m, v0);
                        v1 = fma(cm1, m, v1);
                    }
                    s0 = fma(v0, xv[u].x, s0);
                    s1 = fma(v1, xv[u].y, s1);
                }
            }
            double z0 = 0.0, z1 = 0.0;
            {
                const uint32_t *so2 = b_off2 + r * KS2;
                const double *sva2 = b_va2 + r * KS2;
                double2 xw[K2];
#pragma unroll
                for (int u = 0; u < K2; ++u) xw[u] = sx2.load(so2[u], t0_bytes);
#pragma unroll
                for (int u = 0; u < K2; ++u) {
                    z0 = fma(sva2[u], xw[u].x, z0);
                    z1 = fma(sva2[u], xw[u].y, z1);
                }
            }
            // second pass of the two-pass form: o = 1 * s, then fma(-1, z, o)
            double o0 = fma(-1.0, z0, s0), o1 = fma(-1.0, z1, s1);
            if (!has1) o1 = 0.0;  // padding slot stays zero
            if (a.nt_store)
                store_nt<WIDE>(sy, yo, t0_bytes, make_double2(o0, o1));
            else
                sy.store(yo, t0_bytes, make_double2(o0, o1));
        }
    }
}

template <int K, int K2, bool HAS_M, bool WIDE>
int launch2_npf(hipStream_t st, const Rows2Args &b, unsigned grid, size_t lds)
{
    const int npf = (b.a.R * K + BS - 1) / BS;
    STK_REQUIRE(lds <= 64 * 1024, "rows_ell2: %zu bytes of LDS per workgroup", lds);
    if (npf <= 1)
        hipLaunchKernelGGL((rows_ell2_kernel<K, K2, 1, HAS_M, WIDE>), dim3(grid), dim3(BS), lds, st, b);
    else if (npf <= 2)
        hipLaunchKernelGGL((rows_ell2_kernel<K, K2, 2, HAS_M, WIDE>), dim3(grid), dim3(BS), lds, st, b);
    else
        hipLaunchKernelGGL((rows_ell2_kernel<K, K2, 4, HAS_M, WIDE>), dim3(grid), dim3(BS), lds, st, b);
    STK_LAUNCH_CHECK();
    return 0;
}

int g_rows_wg_per_cu = 0;
int g_rows_force_wide = 0;  // testing: 64-bit addressing on small slabs
int g_rows_alternate = 1;   // alternate the walking direction between launches
int g_rows_nt_store = 1;    // bit 0: SPMM passes, bit 1: Gauss-Seidel groups, bit 2: the two-matrix pass
unsigned g_rows_launch_count = 0;

template <int MODE, int K, bool HAS_M, bool WIDE>
int launch_npf_w(hipStream_t st, const RowsArgs &a, unsigned grid, size_t lds)
{
    const int npf = (a.R * K + BS - 1) / BS;
    if (npf <= 1)
        hipLaunchKernelGGL((rows_ell_kernel<MODE, K, 1, HAS_M, WIDE>), dim3(grid), dim3(BS), lds, st, a);
    else if (npf <= 2)
        hipLaunchKernelGGL((rows_ell_kernel<MODE, K, 2, HAS_M, WIDE>), dim3(grid), dim3(BS), lds, st, a);
    else if (npf <= 4)
        hipLaunchKernelGGL((rows_ell_kernel<MODE, K, 4, HAS_M, WIDE>), dim3(grid), dim3(BS), lds, st, a);
    else {
        stk_set_error("rows_ell: %d slots per row with %d lanes per row not supported", K, a.P);
        return 2;
    }
    STK_LAUNCH_CHECK();
    return 0;
}

template <int MODE, int K, bool HAS_M>
int launch_npf(hipStream_t st, const RowsArgs &a, unsigned grid, size_t lds)
{
    return a.wide ? launch_npf_w<MODE, K, HAS_M, true>(st, a, grid, lds)
                  : launch_npf_w<MODE, K, HAS_M, false>(st, a, grid, lds);
}

template <int MODE, bool HAS_M>
int launch_k(hipStream_t st, const RowsArgs &a, int K, unsigned grid, size_t lds)
{
    switch (K) {
        case 2: return launch_npf<MODE, 2, HAS_M>(st, a, grid, lds);
        case 4: return launch_npf<MODE, 4, HAS_M>(st, a, grid, lds);
        case 5: return launch_npf<MODE, 5, HAS_M>(st, a, grid, lds);
        case 6: return launch_npf<MODE, 6, HAS_M>(st, a, grid, lds);
        case 7: return launch_npf<MODE, 7, HAS_M>(st, a, grid, lds);
        case 9: return launch_npf<MODE, 9, HAS_M>(st, a, grid, lds);
        case 12: return launch_npf<MODE, 12, HAS_M>(st, a, grid, lds);
        case 16: return launch_npf<MODE, 16, HAS_M>(st, a, grid, lds);
        case 20: return launch_npf<MODE, 20, HAS_M>(st, a, grid, lds);
    }
    stk_set_error("rows_ell: K=%d is not one of 2, 4, 5, 6, 7, 9, 12, 16, 20", K);
    return 2;
}

}  // namespace

int stk_rows_ell_set_tuning(const char *key, int32_t value)
{
    if (std::strcmp(key, "rows_force_wide") == 0) {
        g_rows_force_wide = value;
        return 0;
    }
    if (std::strcmp(key, "rows_alternate") == 0) {
        g_rows_alternate = value;
        return 0;
    }
    if (std::strcmp(key, "rows_nt_store") == 0) {
        g_rows_nt_store = value;
        return 0;
    }
    if (std::strcmp(key, "rows_wg_per_cu") == 0) {
        g_rows_wg_per_cu = value;
        return 0;
    }
    return 1;
}

// Shared launcher (also used by mg.hip).  mode: 0 = SPMM, 1 = GS.
int stk_rows_ell_launch(hipStream_t st, int mode, const stk_ell_rows *e, int32_t pos_begin, int32_t pos_end,
                        int32_t n_loc, int32_t ld, int64_t x_rows, int64_t y_rows, double ca, const double *cm,
                        const double *x, double alpha, double beta, const double *z, double *y, int zero_own)
{
    if (pos_end <= pos_begin) return 0;
    STK_REQUIRE(e && e->idx && e->va, "rows_ell: incomplete matrix");
    STK_REQUIRE((cm == nullptr) || e->vm, "rows_ell: cm given but the matrix has no vm");
    STK_REQUIRE(n_loc > 0 && ld >= n_loc && (ld & 1) == 0, "rows_ell: bad n_loc=%d ld=%d (ld must be even)", n_loc,
                ld);
    STK_REQUIRE(x_rows * ld * 8 < ((int64_t)1 << 36) && y_rows * ld * 8 < ((int64_t)1 << 36),
                "rows_ell: slab exceeds 64 GiB");
    STK_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)z) & 15) == 0, "rows_ell: slabs must be 16-byte aligned");
    STK_REQUIRE((n_loc + 1) / 2 <= BS, "rows_ell: n_loc too large");
    RowsArgs a;
    a.idx = e->idx;
    a.va = e->va;
    a.vm = cm ? e->vm : nullptr;
    a.row_ids = e->row_ids;
    a.dia_a = e->dia_a;
    a.dia_m = e->dia_m;
    a.cm = cm;
    a.x = x;
    a.z = z ? z : y;
    a.y = y;
    a.ca = ca;
    a.alpha = alpha;
    a.beta = z ? beta : 0.0;
    a.pos_begin = pos_begin;
    a.pos_end = pos_end;
    a.n_loc = n_loc;
    a.ld = ld;
    a.P = (n_loc + 1) / 2;
    a.R = BS / a.P;
    if (a.R * e->K > 4 * BS) a.R = 4 * BS / e->K;  // at most 4 prefetched entries per thread
    a.ngroups = (pos_end - pos_begin + a.R - 1) / a.R;
    a.chunk = (a.ngroups + 7) / 8;
    a.reverse = g_rows_alternate ? (int)(g_rows_launch_count++ & 1u) : 0;
    a.zero_own = (zero_own || e->diag_free) ? 1 : 0;  // diagonal-free rows: u_i = (f_i - sum_{j != i}) / a_ii
    a.wide = g_rows_force_wide || x_rows * ld * 8 >= ((int64_t)1 << 32) || y_rows * ld * 8 >= ((int64_t)1 << 32);
    a.x_bytes = a.wide ? 0u : (uint32_t)(x_rows * ld * 8);
    a.y_bytes = a.wide ? 0u : (uint32_t)(y_rows * ld * 8);
    a.nt_store = (g_rows_nt_store >> (mode == MODE_GS ? 1 : 0)) & 1;
    if (mode == MODE_GS) STK_REQUIRE(e->dia_a && (!cm || e->dia_m), "rows_ell: GS needs the diagonal arrays");
    const int K = e->K;
    const int KS = (K + 3) & ~3;
    const bool has_m = cm != nullptr;
    const size_t buf_doubles =
        ((size_t)(has_m ? 2 : 1) * a.R * KS + 2 * a.R + ((size_t)a.R * KS + a.R + 1) / 2 + 2) & ~(size_t)1;
    const size_t lds = 2 * buf_doubles * sizeof(double) + 16;
    const int n_cu = stk_cu_count();
    // wide rows (K >= 12) get 128 VGPRs: 2 workgroups per CU
    int per_cu = g_rows_wg_per_cu > 0 ? g_rows_wg_per_cu : (K >= 12 ? 2 : 3);
    int per_xcd = (n_cu / 8) * per_cu;
    if (per_xcd > a.chunk) per_xcd = a.chunk;
    if (per_xcd < 1) per_xcd = 1;
    const unsigned grid = (unsigned)per_xcd * 8;
    if (mode == MODE_GS)
        return has_m ? launch_k<MODE_GS, true>(st, a, K, grid, lds) : launch_k<MODE_GS, false>(st, a, K, grid, lds);
    return has_m ? launch_k<MODE_SPMM, true>(st, a, K, grid, lds) : launch_k<MODE_SPMM, false>(st, a, K, grid, lds);
}

// y = A(t) x - B x2 for the instantiated pair of slot counts (20, 7): the restricted
// residual of mg.hip.  Returns -1 when this pair of matrices has no instantiation (the
// caller then takes the two passes).
int stk_rows_ell2_launch(hipStream_t st, const stk_ell_rows *e, const stk_ell_rows *e2, int32_t n_loc, int32_t ld,
                         int64_t x_rows, int64_t y_rows, double ca, const double *cm, const double *x,
                         const double *x2, double *y)
{
    if (e->K != 20 || e2->K != 7 || e->n_pos != e2->n_pos || e->n_pos <= 0) return -1;
    STK_REQUIRE(e->idx && e->va && e2->idx && e2->va, "rows_ell2: incomplete matrix");
    STK_REQUIRE((cm == nullptr) || e->vm, "rows_ell2: cm given but the matrix has no vm");
    STK_REQUIRE(n_loc > 0 && ld >= n_loc && (ld & 1) == 0, "rows_ell2: bad n_loc=%d ld=%d", n_loc, ld);
    STK_REQUIRE(x_rows * ld * 8 < ((int64_t)1 << 36) && y_rows * ld * 8 < ((int64_t)1 << 36), "rows_ell2: slab exceeds 64 GiB");
    STK_REQUIRE((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)y) & 15) == 0, "rows_ell2: slabs must be 16-byte aligned");
    STK_REQUIRE((n_loc + 1) / 2 <= BS, "rows_ell2: n_loc too large");
    constexpr int K = 20, K2 = 7, KS = 20, KS2 = 8;
    Rows2Args b;
    RowsArgs &a = b.a;
    a.idx = e->idx;
    a.va = e->va;
    a.vm = cm ? e->vm : nullptr;
    a.row_ids = e->row_ids;
    a.dia_a = a.dia_m = nullptr;
    a.cm = cm;
    a.x = x;
    a.z = y;
    a.y = y;
    a.ca = ca;
    a.alpha = 1.0;
    a.beta = 0.0;
    a.pos_begin = 0;
    a.pos_end = e->n_pos;
    a.n_loc = n_loc;
    a.ld = ld;
    a.P = (n_loc + 1) / 2;
    a.R = BS / a.P;
    if (a.R * K > 4 * BS) a.R = 4 * BS / K;  // at most 4 prefetched entries per thread and matrix
    a.ngroups = (e->n_pos + a.R - 1) / a.R;
    a.chunk = (a.ngroups + 7) / 8;
    a.reverse = g_rows_alternate ? (int)(g_rows_launch_count++ & 1u) : 0;
    a.zero_own = 0;
    a.wide = g_rows_force_wide || x_rows * ld * 8 >= ((int64_t)1 << 32) || y_rows * ld * 8 >= ((int64_t)1 << 32);
    a.x_bytes = a.wide ? 0u : (uint32_t)(x_rows * ld * 8);
    a.y_bytes = a.wide ? 0u : (uint32_t)(y_rows * ld * 8);
    a.nt_store = (g_rows_nt_store >> 2) & 1;
    b.idx2 = e2->idx;
    b.va2 = e2->va;
    b.x2 = x2;
    b.x2_bytes = a.x_bytes;
    const bool has_m = cm != nullptr;
    auto lds_of = [&](int R) {
        const size_t val_doubles = (size_t)(has_m ? 2 : 1) * R * KS + (size_t)R * KS2;
        const size_t buf_doubles = (val_doubles + ((size_t)R * KS + (size_t)R * KS2 + R + 1) / 2 + 2) & ~(size_t)1;
        return 2 * buf_doubles * sizeof(double) + 16;
    };
    while (a.R > 1 && lds_of(a.R) > 60 * 1024) --a.R;  // short slabs: two workgroups per CU keep their LDS
    a.ngroups = (e->n_pos + a.R - 1) / a.R;
    a.chunk = (a.ngroups + 7) / 8;
    const size_t lds = lds_of(a.R);
    const int n_cu = stk_cu_count();
    int per_xcd = (n_cu / 8) * 2;
    if (per_xcd > a.chunk) per_xcd = a.chunk;
    if (per_xcd < 1) per_xcd = 1;
    const unsigned grid = (unsigned)per_xcd * 8;
    if (has_m)
        return a.wide ? launch2_npf<K, K2, true, true>(st, b, grid, lds) : launch2_npf<K, K2, true, false>(st, b, grid, lds);
    return a.wide ? launch2_npf<K, K2, false, true>(st, b, grid, lds) : launch2_npf<K, K2, false, false>(st, b, grid, lds);
}

extern "C" int stk_ell_spmm(void *stream, const stk_ell_rows *ell, int32_t n_loc, int32_t ld, int32_t x_rows,
                            double ca, const double *cm, const double *x, double alpha, double beta,
                            const double *z, double *y)
{
    const stk_timed timed_(STK_OP_SPACE, stream);
    STK_REQUIRE(ell && x && y && x != y, "stk_ell_spmm: bad pointers");
    STK_REQUIRE(beta == 0.0 || z, "stk_ell_spmm: beta != 0 needs z");
    return stk_rows_ell_launch(stk_stream(stream), MODE_SPMM, ell, 0, ell->n_pos, n_loc, ld, x_rows, ell->n_rows, ca,
                               cm, x, alpha, beta, beta != 0.0 ? z : nullptr, y);
}
