// Kronecker-sum apply, WAVE-AUTONOMOUS gather kernel:
//     y = beta*y + sum_k (T_k kron X_k) x          (all terms read the same x)
//
// Same operator and same packed matrix stream as stk_kron_pack_apply (reference
// source/mpi_kron.py:77-90, 186-201, 214-219), without any workgroup barrier in
// its loop.  The workgroup-synchronous kernels spend their time in lockstep: per
// row group two barriers, an LDS publish of the entries and an LDS exchange of
// the sums (profiles/r02_*: two thirds of the wave cycles are waits, and the run
// time does not react to where the gathers hit).  Here a WAVEFRONT is the unit:
//
//  * Tasks are numbered row after row in processing order, task = pos * W + p
//    (p = pair of time steps, plus two ghost tasks per row on a slab with
//    neighbours).  A wavefront takes 62 consecutive tasks and recomputes one halo
//    task on either side, so the sums of the neighbouring time steps -- the
//    neighbouring lanes -- arrive by two DPP wave shifts: no LDS round trip.
//  * The entries of a row are one 32-byte record, 7 slot words `code <<
//    col_bits | column` and the output row id (longer rows: 64 bytes); every
//    lane loads the record of its own row (the lanes of a row hit the same
//    line), prefetched one round ahead.  Nothing is published, so nothing is
//    waited for: waves drift apart and hide each other's gather latency.
//  * Dictionary of value tuples and time coefficients are read-only in LDS.
//
// Bit-identical with the other forms (same products, same order of additions).
#include <cstring>

#include "stk_common.h"

namespace {

template <int NT>
struct WaveArgs {
    const uint32_t *recs;    // [M][KS]: K slot words, padding, row id in the last word
    const double *dict[NT];
    const double *tri[NT];
    const double *x;
    const double *gh;  // [M][2] (x_lo, x_hi) or NULL
    double *y;
    double beta;
    int32_t M, n_loc, ld, any_tri;
    int32_t P, W;
    uint32_t magic_W;        // ceil(2^32 / W) (W > 1)
    int64_t n_tasks;         // M * W
    int32_t n_units, chunk;  // units of 62 tasks in total / per XCD
    int32_t col_bits, n_codes, flags;
};

__device__ inline double dpp_prev(double v)  // lane i receives lane i-1's value (0 into lane 0)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ inline double dpp_next(double v)  // lane i receives lane i+1's value (0 into lane 63)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);  // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

typedef double stk_v2d __attribute__((ext_vector_type(2)));

constexpr int wave_ks(int K) { return ((K + 1) + 3) & ~3; }

template <int NT, int K, bool GHOST, int BS>
__global__ __launch_bounds__(BS, K >= 12 ? 4 : 6) void kron_wave_kernel(const WaveArgs<NT> a)
{
    constexpr int KS = wave_ks(K);
    constexpr int WAVES = BS / 64;
    extern __shared__ double sm[];
    const int LT = (a.n_loc + 2) & ~1;
    double *s_tri = sm;                     // [NT][3][LT]
    double *s_dict = s_tri + NT * 3 * LT;   // [n_codes][NT]
    const int tid = threadIdx.x;
    for (int i = tid; i < a.n_codes * NT; i += BS) {
        const int c = i / NT, k = i - c * NT;
        s_dict[i] = a.dict[k][c];
    }
    if (a.any_tri) {
        for (int i = tid; i < NT * 3 * LT; i += BS) {
            const int k = i / (3 * LT), rem = i - k * 3 * LT;
            const int d = rem / LT, t = rem - d * LT;
            s_tri[i] = (a.tri[k] != nullptr && t < a.n_loc) ? a.tri[k][d * a.n_loc + t] : 0.0;
        }
    }
    __syncthreads();  // the only barrier: the tables above are read-only from here on

    const int lane = tid & 63, wave = tid >> 6;
    const int W = a.W;
    const uint32_t col_mask = (1u << a.col_bits) - 1u;
    // units of this wavefront: the waves of an XCD's workgroups (blockIdx % 8) take
    // interleaved units of one contiguous chunk of the task list
    const int xcd = blockIdx.x & 7;
    const int step = (gridDim.x >> 3) * WAVES;
    const int uend = min((xcd + 1) * a.chunk, a.n_units);
    int u = xcd * a.chunk + (int)(blockIdx.x >> 3) * WAVES + wave;

    // task of this lane in unit `uu`: position (row in processing order) and pair
    auto locate = [&](int uu, int &pos, int &p) __attribute__((always_inline)) -> bool {
        const int64_t tau = (int64_t)uu * 62 + lane - 1;  // lanes 0 and 63: halo tasks
        const bool valid = tau >= 0 && tau < a.n_tasks;
        const uint32_t t32 = valid ? (uint32_t)tau : 0u;
        pos = (W == 1) ? (int)t32 : (int)__umulhi(t32, a.magic_W);
        p = (int)t32 - pos * W;
        return valid;
    };
    uint4 rec[KS / 4];
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) rec[q] = make_uint4(0, 0, 0, 0);
    auto fetch = [&](int uu) __attribute__((always_inline)) {
        int pos, p;
        if (locate(uu, pos, p)) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.recs + (size_t)pos * KS);
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) rec[q] = src[q];
        }
    };
    if (u < uend) fetch(u);

    // results of the previous round, stored one round late: vmcnt counts in order, so
    // a wait for the prefetched records issued after a store would also wait for
    // that store to complete.  The store goes out right after the records have been
    // taken over, and overlaps with the gathers that follow.
    double dy0 = 0.0, dy1 = 0.0;
    double2 *ddst = nullptr;
    bool dpend = false;
    auto flush = [&]() __attribute__((always_inline)) {
        if (dpend) {
            if (a.flags & 1) {
                stk_v2d out;
                out.x = dy0, out.y = dy1;
                __builtin_nontemporal_store(out, reinterpret_cast<stk_v2d *>(ddst));
            } else {
                *ddst = make_double2(dy0, dy1);
            }
        }
        dpend = false;
    };
    for (; u < uend; u += step) {
        int pos, p;
        const bool valid = locate(u, pos, p);
        uint32_t sl[KS];
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            // (the empty statement makes every lane take the wait for the records here)
            asm volatile("" ::"v"(rec[q].x), "v"(rec[q].y), "v"(rec[q].z), "v"(rec[q].w));
            sl[4 * q] = rec[q].x, sl[4 * q + 1] = rec[q].y, sl[4 * q + 2] = rec[q].z, sl[4 * q + 3] = rec[q].w;
        }
        flush();
        // next round's records first: they are older than this round's gathers in
        // the (in-order) vmcnt queue, so waiting for the gathers never waits for them
        if (u + step < uend) fetch(u + step);
        const bool g_lo = GHOST && p == 0, g_hi = GHOST && p == W - 1;
        const bool ghost = g_lo || g_hi;
        const int t0 = 2 * (p - (GHOST ? 1 : 0));
        const bool has1 = t0 + 1 < a.n_loc;
        // what distinguishes the lanes of a row: where their 16 bytes of a column start
        const char *base_lane = ghost ? reinterpret_cast<const char *>(a.gh)
                                      : reinterpret_cast<const char *>(a.x) + (size_t)(ghost ? 0 : t0) * 8;
        const uint32_t stride_lane = ghost ? 16u : (uint32_t)a.ld * 8u;

        double2 xv[K];
        if (valid) {
#pragma unroll
            for (int e = 0; e < K; ++e)
                xv[e] = *reinterpret_cast<const double2 *>(base_lane + (size_t)(sl[e] & col_mask) * stride_lane);
        } else {
#pragma unroll
            for (int e = 0; e < K; ++e) xv[e] = make_double2(0.0, 0.0);
        }

        double acc0[NT], acc1[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) acc0[k] = acc1[k] = 0.0;
#pragma unroll
        for (int e = 0; e < K; ++e) {
            const double *dv = s_dict + (sl[e] >> a.col_bits) * NT;
            double v[NT];
            if constexpr (NT == 2) {
                const double2 vv = *reinterpret_cast<const double2 *>(dv);
                v[0] = vv.x, v[1] = vv.y;
            } else {
#pragma unroll
                for (int k = 0; k < NT; ++k) v[k] = dv[k];
            }
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                acc0[k] = fma(v[k], xv[e].x, acc0[k]);
                acc1[k] = fma(v[k], xv[e].y, acc1[k]);
            }
        }
        // What a lane shows its neighbours: to the right its last step (acc1), to
        // the left its first (acc0).  The ghost task before the row shows z[-1] =
        // the x_lo sums (acc0 of a ghost pair) to its right; the one behind it
        // shows z[n_loc] = the x_hi sums (acc1) to its left.
        double y0 = 0.0, y1 = 0.0;
        if (a.any_tri) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const double show_r = g_lo ? acc0[k] : acc1[k];
                const double show_l = g_hi ? acc1[k] : acc0[k];
                double zl = dpp_prev(show_r);  // z[t0 - 1]
                double zr = dpp_next(show_l);  // z[t0 + 2] (z[t0 + 1] on an odd tail)
                if (p == 0) zl = 0.0;          // nothing before the row
                if (p == W - 1) zr = 0.0;      // nothing behind it
                if (a.tri[k] != nullptr) {
                    const double *c = s_tri + k * 3 * LT + (ghost ? 0 : t0);
                    const double2 sub = *reinterpret_cast<const double2 *>(c);
                    const double2 dia = *reinterpret_cast<const double2 *>(c + LT);
                    const double2 sup = *reinterpret_cast<const double2 *>(c + 2 * LT);
                    double v0 = dia.x * acc0[k];
                    v0 = fma(sub.x, zl, v0);
                    v0 = fma(sup.x, has1 ? acc1[k] : zr, v0);
                    y0 += v0;
                    double v1 = dia.y * acc1[k];
                    v1 = fma(sub.y, acc0[k], v1);
                    v1 = fma(sup.y, zr, v1);
                    y1 += v1;
                } else {
                    y0 += acc0[k];
                    y1 += acc1[k];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                y0 += acc0[k];
                y1 += acc1[k];
            }
        }
        dpend = valid && !ghost && lane >= 1 && lane <= 62;
        if (dpend) {
            if (!has1) y1 = 0.0;  // padding slot stays zero
            ddst = reinterpret_cast<double2 *>(a.y + (size_t)sl[KS - 1] * a.ld + t0);
            if (a.beta != 0.0) {
                const double2 old = *ddst;
                y0 = fma(a.beta, old.x, y0);
                if (has1) y1 = fma(a.beta, old.y, y1);
            }
            dy0 = y0, dy1 = y1;
        }
    }
    flush();
}

int g_wave_wg_per_cu = 0;
int g_wave_flags = 0;
int g_wave_block = 256;

template <int NT, int K, int BS>
int launch_bs(hipStream_t st, WaveArgs<NT> a, bool ghost)
{
    constexpr int WAVES = BS / 64;
    const int LT = (a.n_loc + 2) & ~1;
    const size_t lds = 8 * ((size_t)NT * 3 * LT + (size_t)a.n_codes * NT) + 32;
    STK_REQUIRE(lds <= 48 * 1024, "stk_kron_wave_apply: %zu bytes of LDS per workgroup (dictionary too large?)", lds);
    const int n_cu = stk_cu_count();
    // waves per CU as the gather kernels: 24 (16 for wide rows)
    int per_cu = g_wave_wg_per_cu > 0 ? g_wave_wg_per_cu : (K >= 12 ? 16 : 24) / WAVES;
    int per_xcd = (n_cu / 8) * per_cu;
    const int need = (a.chunk + WAVES - 1) / WAVES;
    if (per_xcd > need) per_xcd = need;
    if (per_xcd < 1) per_xcd = 1;
    const unsigned grid = (unsigned)per_xcd * 8;
    if (ghost)
        hipLaunchKernelGGL((kron_wave_kernel<NT, K, true, BS>), dim3(grid), dim3(BS), lds, st, a);
    else
        hipLaunchKernelGGL((kron_wave_kernel<NT, K, false, BS>), dim3(grid), dim3(BS), lds, st, a);
    STK_LAUNCH_CHECK();
    return 0;
}

template <int NT, int K>
int launch(hipStream_t st, const WaveArgs<NT> &a, bool ghost)
{
    return g_wave_block == 512 ? launch_bs<NT, K, 512>(st, a, ghost) : launch_bs<NT, K, 256>(st, a, ghost);
}

template <int NT>
int dispatch(hipStream_t st, const stk_wave_pattern *pat, int32_t n_loc, int32_t ld, const stk_kron_pack_term *t,
             const double *x, const double *gh, double beta, double *y)
{
    WaveArgs<NT> a;
    a.recs = pat->recs;
    a.x = x;
    a.gh = gh;
    a.y = y;
    a.beta = beta;
    a.M = pat->M;
    a.n_loc = n_loc;
    a.ld = ld;
    a.col_bits = pat->col_bits;
    a.n_codes = pat->n_codes;
    a.any_tri = 0;
    for (int k = 0; k < NT; ++k) {
        a.dict[k] = pat->dict + (size_t)t[k].mat * pat->n_codes;
        a.tri[k] = t[k].tri;
        if (t[k].tri) a.any_tri = 1;
    }
    a.P = (n_loc + 1) / 2;
    a.W = a.P + (gh ? 2 : 0);
    a.magic_W = a.W > 1 ? (uint32_t)((((uint64_t)1 << 32) + a.W - 1) / a.W) : 0u;
    a.n_tasks = (int64_t)pat->M * a.W;
    STK_REQUIRE(a.n_tasks < ((int64_t)1 << 31), "stk_kron_wave_apply: %lld tasks exceed 2^31", (long long)a.n_tasks);
    a.n_units = (int32_t)((a.n_tasks + 61) / 62);
    a.chunk = (a.n_units + 7) / 8;
    a.flags = g_wave_flags;
    switch (pat->K) {
        case 5: return launch<NT, 5>(st, a, gh != nullptr);
        case 7: return launch<NT, 7>(st, a, gh != nullptr);
        case 9: return launch<NT, 9>(st, a, gh != nullptr);
        case 12: return launch<NT, 12>(st, a, gh != nullptr);
        case 16: return launch<NT, 16>(st, a, gh != nullptr);
    }
    stk_set_error("stk_kron_wave_apply: K=%d is not one of 5, 7, 9, 12, 16", pat->K);
    return 2;
}

}  // namespace

int stk_kron_wave_set_tuning(const char *key, int32_t value)
{
    if (std::strcmp(key, "wave_wg_per_cu") == 0) {
        g_wave_wg_per_cu = value;
        return 0;
    }
    if (std::strcmp(key, "wave_flags") == 0) {
        g_wave_flags = value;
        return 0;
    }
    if (std::strcmp(key, "wave_block") == 0) {
        g_wave_block = value == 512 ? 512 : 256;
        return 0;
    }
    return 1;
}

extern "C" int stk_kron_wave_apply(void *stream, const stk_wave_pattern *pat, int32_t n_loc, int32_t ld,
                                   int32_t n_terms, const stk_kron_pack_term *t, const double *x,
                                   const double *ghosts, double beta, double *y)
{
    STK_REQUIRE(pat && t && x && y, "stk_kron_wave_apply: null pointer");
    STK_REQUIRE(pat->M > 0 && pat->K >= 1 && pat->recs && pat->dict, "stk_kron_wave_apply: bad pattern");
    STK_REQUIRE(pat->col_bits >= 1 && pat->col_bits <= 31 && ((int64_t)1 << pat->col_bits) >= pat->M,
                "stk_kron_wave_apply: col_bits=%d cannot address %d columns", pat->col_bits, pat->M);
    STK_REQUIRE(pat->n_codes >= 1 && (int64_t)pat->n_codes <= ((int64_t)1 << (32 - pat->col_bits)),
                "stk_kron_wave_apply: %d codes do not fit %d bits", pat->n_codes, 32 - pat->col_bits);
    STK_REQUIRE(n_loc > 0 && ld >= n_loc && (ld & 1) == 0,
                "stk_kron_wave_apply: bad sizes n_loc=%d ld=%d (ld must be even)", n_loc, ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_wave_apply: n_terms=%d not in 1..3", n_terms);
    STK_REQUIRE(x != y, "stk_kron_wave_apply: input aliases output");
    STK_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)ghosts | (uintptr_t)pat->recs) & 15) == 0,
                "stk_kron_wave_apply: x, y, ghosts and records must be 16-byte aligned");
    for (int k = 0; k < n_terms; ++k)
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < pat->n_mats, "stk_kron_wave_apply: term %d names matrix %d of %d", k,
                    t[k].mat, pat->n_mats);
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return dispatch<1>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
        case 2: return dispatch<2>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
        default: return dispatch<3>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
    }
}
