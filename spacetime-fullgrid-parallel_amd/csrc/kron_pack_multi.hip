// Kronecker-sum apply on the packed slot stream with an INPUT SLAB PER TERM,
//     y = beta*y + sum_k (T_k kron X_k) xs[k],
// with a LANE GROUP PER TERM: the last stage of the regrouped Schur complement,
// (I kron M_x) v1 + (I kron A_x) v2 + (G_t kron M_x) x (reference
// heateq_mpi.py:166-181; three TridiagKronMatMPI / IdentityKronMatMPI applies and the
// sum of SumMPI._matvec, mpi_kron.py:77-90, 135-150, 214-219, in one pass).
//
// Round 4 let the terms take TURNS in one lane (kron_pack.hip, MULTI = true: K gathers
// from xs[0], its sums, K gathers from xs[1], ...).  Its counters (round 5,
// profiles/r05_refetch_pmc_turns_*.txt) showed what that costs: a workgroup of 15 slot rows x 33
// lanes keeps TWO slabs' neighbourhoods of 30 matrix rows alive, the 64 workgroups of
// an XCD together more than its 4 MiB L2 holds, and the launch fetched 3.1 GB where
// 1.3 GB suffice; with half the workgroups the traffic halved but the time grew --
// too few gathers in flight.  Here every slot row gets one lane per (term, pair of
// time steps) that the term's time factor reaches:
//   * a lane runs the ONE-input instruction stream of kron_pack.hip (K gathers of 16
//     bytes back to back, the slot words re-read for their codes, the dictionary values
//     of ITS term), so twice the gathers are in flight per slot row and a workgroup
//     holds half as many slot rows: the XCD's window shrinks to what its L2 keeps;
//   * the sums z_k[row][t] of all lanes meet in LDS (as the time stencil's neighbours
//     always did), and the first P lanes of a slot row -- one per pair of time steps --
//     apply the time factors and add the terms IN TERM ORDER with the roundings of the
//     one-input form: results are bit for bit those of stk_kron_pack_apply term by term
//     (and of the round-4 kernel);
//   * a term whose time factor reaches few time steps (G_t has the single entry
//     (0, 0)) gets lanes for those pairs only: the caller states the range
//     (stk_kron_pack_apply_multi_steps), and entries of z_k outside it are the zeros
//     the kernel start leaves in LDS.
#include <cstring>

#include "stk_common.h"

namespace {

constexpr int BS = 512;

template <int NT>
struct TermArgs {
    const uint32_t *slots;   // [n_units][K]
    const int32_t *row_ids;  // [n_units][RP] (-1: no row) or NULL (RP = 1, index order)
    const double *dict[NT];  // [n_codes][RP] values of term k's matrix per code
    const double *tri[NT];   // [3][n_loc] or NULL (identity)
    const double *xk[NT];    // the input slab of term k
    double *y;
    double beta;
    int32_t n_units, n_loc, ld;
    int32_t P, W, R;         // output lanes per unit, lanes per unit, units per group
    int32_t lane0[NT + 1];   // lanes [lane0[k], lane0[k + 1]) of a unit belong to term k ...
    int32_t pair0[NT];       // ... and cover the pairs of time steps pair0[k], pair0[k] + 1, ...
    int32_t ngroups, chunk;  // groups in total / per XCD
    int32_t col_bits, n_codes;
    int32_t flags;  // bit 0: non-temporal y stores, bit 1: non-temporal slot loads
};

typedef double stk_v2d __attribute__((ext_vector_type(2)));

__device__ inline double2 load2(const char *p) { return *reinterpret_cast<const double2 *>(p); }

template <int NT, int K, int NPF, int RP>
__global__ __launch_bounds__(BS, 4) void kron_pack_terms_kernel(const TermArgs<NT> a)
{
    constexpr int KS = (K + 3) & ~3;  // LDS stride of a row's slots (16-byte vectors)
    extern __shared__ double sm[];
    const int W = a.W, R = a.R, SW = a.n_loc + 3;
    uint32_t *s_slot = reinterpret_cast<uint32_t *>(sm);             // [R][KS]
    int32_t *s_row = reinterpret_cast<int32_t *>(s_slot + R * KS);  // [R][RP]
    const int LT = (a.n_loc + 2) & ~1;
    double *s_tri = reinterpret_cast<double *>(s_row + ((R * RP + 3) & ~3));  // [NT][3][LT]
    double *s_dict = s_tri + NT * 3 * LT;                                      // [n_codes][RP][NT]
    // s_w[k][r * RP + j][q], q = t + 1: z_k[row][t] for t = -1 .. n_loc (the two ends stay zero)
    double *s_w = s_dict + a.n_codes * RP * NT;

    const int tid = threadIdx.x;
    const int r = tid / W;
    const int l = tid - r * W;
    const bool in_row = r < R;
    // which term and which pair of time steps this lane gathers
    int term = 0;
#pragma unroll
    for (int k = 1; k < NT; ++k)
        if (l >= a.lane0[k]) term = k;
    const int t0 = 2 * (a.pair0[term] + l - a.lane0[term]);
    const bool has1 = t0 + 1 < a.n_loc;
    const bool out_lane = l < a.P;  // also responsible for the pair of time steps 2 l, 2 l + 1 of y
    const int o0 = 2 * l;
    const bool ohas1 = o0 + 1 < a.n_loc;
    const uint32_t stride = (uint32_t)a.ld * 8u;
    const uint32_t col_mask = (1u << a.col_bits) - 1u;

    for (int i = tid; i < a.n_codes * RP * NT; i += BS) {
        const int c = i / NT, k = i - c * NT;  // c = code * RP + row of the pair
        s_dict[i] = a.dict[k][c];
    }
    for (int i = tid; i < NT * 3 * LT; i += BS) {
        const int k = i / (3 * LT), rem = i - k * 3 * LT;
        const int d = rem / LT, t = rem - d * LT;
        s_tri[i] = (a.tri[k] != nullptr && t < a.n_loc) ? a.tri[k][d * a.n_loc + t] : 0.0;
    }
    for (int i = tid; i < NT * R * RP * SW; i += BS) s_w[i] = 0.0;
    __syncthreads();

    // does this lane's term reach its pair of time steps at all?  z[t] enters the output
    // through sub[t + 1], dia[t] and super[t - 1]; an identity factor reaches every step.
    bool need = true;
    const char *base_lane = nullptr;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if (k != term) continue;
        base_lane = reinterpret_cast<const char *>(a.xk[k]) + (size_t)t0 * 8;
        if (a.tri[k] != nullptr) {
            const double *c = s_tri + k * 3 * LT;
            bool used = false;
            for (int t = t0; t < min(t0 + 2, a.n_loc); ++t) {
                used = used || c[LT + t] != 0.0;
                if (t + 1 < a.n_loc) used = used || c[t + 1] != 0.0;
                if (t >= 1) used = used || c[2 * LT + t - 1] != 0.0;
            }
            need = used;
        }
    }
    if (t0 >= a.n_loc) need = false;

    const int xcd = blockIdx.x & 7;
    const int step = gridDim.x >> 3;
    const int gend = min((xcd + 1) * a.chunk, a.ngroups);
    int g = xcd * a.chunk + (int)(blockIdx.x >> 3);

    uint32_t pslot[NPF];
    int32_t prow = 0;
#pragma unroll
    for (int q = 0; q < NPF; ++q) pslot[q] = 0;
    auto fetch = [&](int gq) {
        const int rows = min(R, a.n_units - gq * R);
        const uint32_t *src = a.slots + (size_t)gq * R * K;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) pslot[q] = (a.flags & 2) ? __builtin_nontemporal_load(src + i) : src[i];
        }
        if (tid < rows * RP) prow = a.row_ids ? a.row_ids[(size_t)gq * R * RP + tid] : gq * R + tid;
    };
    if (g < gend) fetch(g);

    for (; g < gend; g += step) {
        const int rows = min(R, a.n_units - g * R);
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) s_slot[i + (i / K) * (KS - K)] = pslot[q];
        }
        if (tid < rows * RP) s_row[tid] = prow;
        __syncthreads();
        if (g + step < gend) fetch(g + step);  // in flight behind the gathers

        const bool active = in_row && r < rows;
        // read now: s_row is rewritten at the top of the next iteration, which a fast wave
        // reaches while a slow one is still storing
        int32_t yrow[RP];
#pragma unroll
        for (int j = 0; j < RP; ++j) yrow[j] = (active && out_lane) ? s_row[r * RP + j] : -1;

        if (active && need) {
            double acc0[RP], acc1[RP];
#pragma unroll
            for (int j = 0; j < RP; ++j) acc0[j] = acc1[j] = 0.0;
            int ro = r * KS;
            double2 xv[K];
            {
                uint32_t sl[KS];
                const uint4 *so = reinterpret_cast<const uint4 *>(s_slot + ro);
#pragma unroll
                for (int u = 0; u < KS / 4; ++u) {
                    const uint4 v = so[u];
                    sl[4 * u] = v.x, sl[4 * u + 1] = v.y, sl[4 * u + 2] = v.z, sl[4 * u + 3] = v.w;
                }
#pragma unroll
                for (int u = 0; u < K; ++u) xv[u] = load2(base_lane + (size_t)(sl[u] & col_mask) * stride);
            }
            // the slot words are read a second time for their codes rather than kept in
            // registers across the gathers (as in kron_pack.hip)
            asm volatile("" : "+v"(ro));
            uint32_t sl[KS];
            const uint4 *so = reinterpret_cast<const uint4 *>(s_slot + ro);
#pragma unroll
            for (int u = 0; u < KS / 4; ++u) {
                const uint4 v = so[u];
                sl[4 * u] = v.x, sl[4 * u + 1] = v.y, sl[4 * u + 2] = v.z, sl[4 * u + 3] = v.w;
            }
#pragma unroll
            for (int u = 0; u < K; ++u) {
                const double *dv = s_dict + (sl[u] >> a.col_bits) * (RP * NT) + term;
#pragma unroll
                for (int j = 0; j < RP; ++j, dv += NT) {
                    const double v = dv[0];
                    acc0[j] = fma(v, xv[u].x, acc0[j]);
                    acc1[j] = fma(v, xv[u].y, acc1[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                double *w = s_w + ((term * R + r) * RP + j) * SW + t0 + 1;
                w[0] = acc0[j];
                if (has1) w[1] = acc1[j];
            }
        }
        __syncthreads();

        if (active && out_lane) {
            double y0[RP], y1[RP];
#pragma unroll
            for (int j = 0; j < RP; ++j) y0[j] = y1[j] = 0.0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (a.tri[k] != nullptr) {
                    const double *c = s_tri + k * 3 * LT + o0;
                    const double2 sub = *reinterpret_cast<const double2 *>(c);
                    const double2 dia = *reinterpret_cast<const double2 *>(c + LT);
                    const double2 sup = *reinterpret_cast<const double2 *>(c + 2 * LT);
#pragma unroll
                    for (int j = 0; j < RP; ++j) {
                        const double *w = s_w + ((k * R + r) * RP + j) * SW + o0;  // w[q]: z at step o0 - 1 + q
                        double v0 = dia.x * w[1];
                        v0 = fma(sub.x, w[0], v0);
                        v0 = fma(sup.x, w[2], v0);
                        y0[j] += v0;
                        if (ohas1) {
                            double v1 = dia.y * w[2];
                            v1 = fma(sub.y, w[1], v1);
                            v1 = fma(sup.y, w[3], v1);
                            y1[j] += v1;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < RP; ++j) {
                        const double *w = s_w + ((k * R + r) * RP + j) * SW + o0;
                        y0[j] += w[1];
                        y1[j] += w[2];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                if (RP > 1 && yrow[j] < 0) continue;  // a slot row that serves one matrix row only
                if (!ohas1) y1[j] = 0.0;              // padding slot stays zero
                double2 *dst = reinterpret_cast<double2 *>(
                    reinterpret_cast<char *>(a.y) + (size_t)(uint32_t)yrow[j] * ((size_t)a.ld * 8) + (size_t)o0 * 8);
                if (a.beta != 0.0) {
                    const double2 old = *dst;
                    y0[j] = fma(a.beta, old.x, y0[j]);
                    if (ohas1) y1[j] = fma(a.beta, old.y, y1[j]);
                }
                if (a.flags & 1) {
                    stk_v2d out;
                    out.x = y0[j], out.y = y1[j];
                    __builtin_nontemporal_store(out, reinterpret_cast<stk_v2d *>(dst));
                } else {
                    *dst = make_double2(y0[j], y1[j]);
                }
            }
        }
        // the next iteration's barrier (after its slot words are published) orders these
        // reads of s_w before the next writes to it
    }
}

int g_terms_wg_per_cu = 0;  // 0: two workgroups per CU
int g_terms_flags = 3;
int g_terms_r = 0;  // cap on the slot rows of a group (0: as many as 512 lanes hold)

template <int NT, int K, int RP>
int launch_npf(hipStream_t st, const TermArgs<NT> &a, unsigned grid, size_t lds)
{
    const int npf = (a.R * K + BS - 1) / BS;
    if (npf <= 1)
        hipLaunchKernelGGL((kron_pack_terms_kernel<NT, K, 1, RP>), dim3(grid), dim3(BS), lds, st, a);
    else if (npf <= 2)
        hipLaunchKernelGGL((kron_pack_terms_kernel<NT, K, 2, RP>), dim3(grid), dim3(BS), lds, st, a);
    else
        hipLaunchKernelGGL((kron_pack_terms_kernel<NT, K, 4, RP>), dim3(grid), dim3(BS), lds, st, a);
    STK_LAUNCH_CHECK();
    return 0;
}

template <int NT, int RP>
int launch(hipStream_t st, TermArgs<NT> a, int K)
{
    a.R = BS / a.W;
    if (g_terms_r > 0 && a.R > g_terms_r) a.R = g_terms_r;
    if (a.R * K > 4 * BS) a.R = 4 * BS / K;  // at most 4 prefetched words per thread
    const int KS = (K + 3) & ~3;
    auto lds_of = [&](int R) {
        return sizeof(double) * ((size_t)NT * R * RP * (a.n_loc + 3) + (size_t)a.n_codes * RP * NT +
                                 (size_t)NT * 3 * (a.n_loc + 2)) +
               sizeof(uint32_t) * ((size_t)R * KS + (size_t)R * RP + 4) + 32;
    };
    while (a.R > 1 && lds_of(a.R) > 64 * 1024) --a.R;
    a.ngroups = (a.n_units + a.R - 1) / a.R;
    a.chunk = (a.ngroups + 7) / 8;
    a.flags = g_terms_flags;
    const size_t lds = lds_of(a.R);
    STK_REQUIRE(lds <= 64 * 1024, "stk_kron_pack_apply_multi: %zu bytes of LDS per workgroup", lds);
    int per_cu = g_terms_wg_per_cu > 0 ? g_terms_wg_per_cu : 2;
    const int by_lds = (int)(160 * 1024 / (lds + 256));
    if (per_cu > by_lds) per_cu = by_lds > 0 ? by_lds : 1;
    int per_xcd = (stk_cu_count() / 8) * per_cu;
    if (per_xcd > a.chunk) per_xcd = a.chunk;
    if (per_xcd < 1) per_xcd = 1;
    const unsigned grid = (unsigned)per_xcd * 8;
#define STK_TERMS_CASE(KK) \
    case KK: return launch_npf<NT, KK, RP>(st, a, grid, lds);
    if constexpr (RP == 1) {
        switch (K) {
            STK_TERMS_CASE(5)
            STK_TERMS_CASE(7)
            STK_TERMS_CASE(9)
            STK_TERMS_CASE(12)
            STK_TERMS_CASE(16)
        }
    } else {
        switch (K) {
            STK_TERMS_CASE(8)
            STK_TERMS_CASE(10)
            STK_TERMS_CASE(12)
        }
    }
#undef STK_TERMS_CASE
    stk_set_error("stk_kron_pack_apply_multi: K=%d has no instantiation for %d rows per slot row", K, RP);
    return 2;
}

template <int NT>
int dispatch(hipStream_t st, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld, const stk_kron_pack_term *t,
             const double *const *xs, const int32_t *t_begin, const int32_t *t_end, double beta, double *y)
{
    TermArgs<NT> a;
    a.slots = pat->slots;
    a.row_ids = pat->row_ids;
    a.y = y;
    a.beta = beta;
    a.n_units = pat->n_units;
    a.n_loc = n_loc;
    a.ld = ld;
    a.col_bits = pat->col_bits;
    a.n_codes = pat->n_codes;
    a.P = (n_loc + 1) / 2;
    // the pairs of time steps term k's lanes cover; the first P lanes of a slot row are the
    // output lanes (lane l: steps 2 l, 2 l + 1), so the first lane group must cover all
    // pairs: term 0's does (a stated range of term 0 only spares its lanes the gathers)
    int lanes = 0;
    for (int k = 0; k < NT; ++k) {
        int p0 = 0, p1 = a.P;
        if (k > 0 && t_begin && t_end) {
            const int b = t_begin[k] < 0 ? 0 : t_begin[k], e = t_end[k] > n_loc ? n_loc : t_end[k];
            p0 = b / 2;
            p1 = e > b ? (e + 1) / 2 : p0;
        }
        a.dict[k] = pat->dict + (size_t)t[k].mat * pat->n_codes * pat->rows_per_unit;
        a.tri[k] = t[k].tri;
        a.xk[k] = xs[k];
        a.lane0[k] = lanes;
        a.pair0[k] = p0;
        lanes += p1 - p0;
    }
    a.lane0[NT] = lanes;
    a.W = lanes;
    // fewer than two slot rows per workgroup (more than 256 lanes per slot row: three
    // full-range terms from about 171 time steps on): up to half the lanes of the
    // workgroup would idle behind a barrier pair per slot row -- the caller takes the
    // turn form (nothing was ever measured in that regime)
    if (BS / a.W < 2) return -1;
    return pat->rows_per_unit == 2 ? launch<NT, 2>(st, a, pat->K) : launch<NT, 1>(st, a, pat->K);
}

}  // namespace

int stk_kron_pack_terms_set_tuning(const char *key, int32_t value)
{
    if (std::strcmp(key, "terms_wg_per_cu") == 0) {
        g_terms_wg_per_cu = value;
        return 0;
    }
    if (std::strcmp(key, "terms_r") == 0) {
        g_terms_r = value;
        return 0;
    }
    if (std::strcmp(key, "terms_flags") == 0) {
        g_terms_flags = value;
        return 0;
    }
    return 1;
}

// Shared with kron_pack.hip (stk_kron_pack_apply_multi): arguments already checked there.
// Returns -1 when fewer than two slot rows would fit a workgroup (more than 256 lanes per slot row).
int stk_kron_pack_terms_launch(hipStream_t st, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld, int32_t n_terms,
                               const stk_kron_pack_term *t, const double *const *xs, const int32_t *t_begin,
                               const int32_t *t_end, double beta, double *y)
{
    if (n_terms == 2) return dispatch<2>(st, pat, n_loc, ld, t, xs, t_begin, t_end, beta, y);
    return dispatch<3>(st, pat, n_loc, ld, t, xs, t_begin, t_end, beta, y);
}
