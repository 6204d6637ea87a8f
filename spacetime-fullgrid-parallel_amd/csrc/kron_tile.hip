// Kronecker-sum apply, TILE-STAGED:
//     y = beta*y + sum_k (T_k kron X_k) x          (all terms read the same x)
//
// Same operator as stk_kron_ell_apply / stk_kron_pack_apply (reference
// source/mpi_kron.py:77-90, 186-201, 214-219).  The gather kernels spend their
// time waiting: every row group is one dependent chain "entries -> K gathers per
// lane -> sums -> exchange -> store" with two workgroup barriers in it, and each
// time column is fetched through the vector memory path once per row that
// references it (7 times on a triangulation).  Measured on the packed gather
// kernel (profiles/r02_*): the run time does not move when all K gathers of a
// lane hit the same column, and drops by only 16 % with one gather instead of 7
// -- it is the chain, not the bytes.  This kernel cuts the chain:
//
//  * Rows are processed in TILES of consecutive rows of a patch-wise order
//    (source/assembly.py: strips of a few mesh rows inside L2-sized macro tiles).
//    The host lists the DISTINCT columns a tile references (about 2 per row
//    instead of 7).  A workgroup copies those time columns into LDS once, with
//    coalesced 16-byte loads issued a whole tile AHEAD (they land in registers
//    while the previous tile is being computed), so no load latency sits on the
//    path of a row, and the vector memory path moves each column once per tile.
//  * The K "gathers" of a lane are LDS reads.  Slots are 16-bit words,
//    `code << 7 | local column`, the values come from the dictionary of distinct
//    value tuples (as in kron_pack.hip): 2 bytes per slot of matrix stream.
//  * The time stencil needs the sums of the neighbouring time steps, i.e. of the
//    neighbouring lanes of the same row.  Lanes are laid out task after task
//    (task = row * W + pair) and a wavefront takes 62 consecutive tasks plus one
//    halo task on either side, recomputed from LDS for free; the neighbours'
//    sums then arrive by two DPP wave shifts -- no LDS round trip, no barrier.
//    What is left per tile is the pair of barriers around the LDS fill.
//  * Ghost time steps (a slab with neighbour ranks) are two more tasks per row
//    that read the tile's ghost pairs (x_lo[j], x_hi[j]) instead of a time
//    column: same instruction stream, different base and stride.
//
// Results are bit-identical with the other two forms when there are no ghost
// rows (same products, same order of additions).
#include <cstring>

#include "stk_common.h"

namespace {

constexpr int BS = 512;
constexpr int WAVES = BS / 64;
constexpr int TASKS_PER_PASS = WAVES * 62;  // 62 output lanes per wavefront
// NPF (template): 16-byte pieces of the next tile a thread holds in registers, 3
// or 6.  Three keep the kernel within 80 VGPRs (three workgroups per CU) and suit
// one-pass tiles; six hold a two-pass tile at two workgroups per CU.

template <int NT>
struct TileArgs {
    const int32_t *tile_row_ptr, *tile_col_ptr, *tile_cols;
    const uint16_t *slots;   // [M][K]
    const int32_t *row_ids;  // [M] or NULL
    const double *dict[NT];
    const double *tri[NT];
    const double *x;
    const double *gh;  // [M][2] (x_lo, x_hi) or NULL
    double *y;
    double beta;
    int32_t M, n_loc, ld, any_tri;
    int32_t P, W;            // own pairs per row, tasks per row
    uint32_t magic_W, magic_P;  // ceil(2^32 / W), ceil(2^32 / P)
    int32_t n_tiles, chunk;  // tiles in total / per XCD
    int32_t nc_max, tr_max;  // LDS sizing: columns / rows of the largest tile
    int32_t n_codes, flags;
};

__device__ inline double dpp_from_prev(double v)  // lane i receives lane i-1's value (0 into lane 0)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ inline double dpp_from_next(double v)  // lane i receives lane i+1's value (0 into lane 63)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);  // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

typedef double stk_v2d __attribute__((ext_vector_type(2)));

template <int NT, int K, bool GHOST, int NPF>
__global__ __launch_bounds__(BS, NPF <= 3 ? 6 : 4) void kron_tile_kernel(const TileArgs<NT> a)
{
    constexpr int KS = (K + 7) & ~7;  // 16-bit slots of a row padded to 16 bytes
    extern __shared__ double sm[];
    const int P = a.P, W = a.W;
    // LDS: time columns of the tile [nc][P] pairs | ghost pairs [nc] | slots | row ids | tri | dict
    double2 *s_x = reinterpret_cast<double2 *>(sm);
    double2 *s_g = s_x + (size_t)a.nc_max * P;
    uint16_t *s_slot = reinterpret_cast<uint16_t *>(s_g + (GHOST ? a.nc_max : 0));  // [tr_max][KS]
    int32_t *s_row = reinterpret_cast<int32_t *>(s_slot + (size_t)a.tr_max * KS);   // [tr_max]
    const int LT = (a.n_loc + 2) & ~1;
    double *s_tri = reinterpret_cast<double *>(s_row + ((a.tr_max + 3) & ~3));  // [NT][3][LT]
    double *s_dict = s_tri + NT * 3 * LT;                                       // [n_codes][NT]

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

    // tiles of this workgroup: interleaved with the other workgroups of its XCD
    const int xcd = blockIdx.x & 7;
    const int step = gridDim.x >> 3;
    const int gend = min((xcd + 1) * a.chunk, a.n_tiles);
    int g = xcd * a.chunk + (int)(blockIdx.x >> 3);

    // ---- prefetch registers: what this thread copies into LDS for the next tile --
    // (named scalars, not an array: the loads must stay in registers until the
    // LDS fill, and an array captured by the lambda ends up in scratch memory)
    double2 pf0, pf1, pf2, pf3, pf4, pf5, pgh = make_double2(0.0, 0.0);
    pf0 = pf1 = pf2 = pf3 = pf4 = pf5 = pgh;
    static_assert(NPF == 3 || NPF == 6, "three or six prefetch registers are spelled out");
    uint16_t pslot = 0;
    int32_t prow = 0;
    int r0n = 0, nrn = 0, ncn = 0;  // rows / columns of the prefetched tile
#define STK_TILE_LOAD(Q, DST)                                                                        \
    {                                                                                                \
        const int c = tid + Q * BS;                                                                  \
        if (c < total) {                                                                             \
            const int j = P == 1 ? c : (int)__umulhi((uint32_t)c, a.magic_P); /* c / P */            \
            DST = *reinterpret_cast<const double2 *>(a.x + (size_t)cols[j] * a.ld + 2 * (c - j * P)); \
        }                                                                                            \
    }
    auto prefetch = [&](int t) __attribute__((always_inline)) {
        r0n = a.tile_row_ptr[t];
        nrn = a.tile_row_ptr[t + 1] - r0n;
        const int c0 = a.tile_col_ptr[t];
        ncn = a.tile_col_ptr[t + 1] - c0;
        const int32_t *cols = a.tile_cols + c0;
        const int total = ncn * P;
        STK_TILE_LOAD(0, pf0)
        STK_TILE_LOAD(1, pf1)
        STK_TILE_LOAD(2, pf2)
        if constexpr (NPF > 3) {
            STK_TILE_LOAD(3, pf3)
            STK_TILE_LOAD(4, pf4)
            STK_TILE_LOAD(5, pf5)
        }
        if (GHOST && tid < ncn) pgh = *reinterpret_cast<const double2 *>(a.gh + 2 * (size_t)cols[tid]);
        if (tid < nrn * K) pslot = a.slots[(size_t)r0n * K + tid];
        if (tid < nrn) prow = a.row_ids ? a.row_ids[r0n + tid] : r0n + tid;
    };
#undef STK_TILE_LOAD
    if (g < gend) prefetch(g);

    const int wave = tid >> 6, lane = tid & 63;
    for (; g < gend; g += step) {
        const int nr = nrn, nc = ncn;
        __syncthreads();  // every wave is done with the previous tile's LDS image
        {
            const int total = nc * P;
            if (tid < total) s_x[tid] = pf0;
            if (tid + BS < total) s_x[tid + BS] = pf1;
            if (tid + 2 * BS < total) s_x[tid + 2 * BS] = pf2;
            if constexpr (NPF > 3) {
                if (tid + 3 * BS < total) s_x[tid + 3 * BS] = pf3;
                if (tid + 4 * BS < total) s_x[tid + 4 * BS] = pf4;
                if (tid + 5 * BS < total) s_x[tid + 5 * BS] = pf5;
            }
            if (GHOST && tid < nc) s_g[tid] = pgh;
            if (tid < nr * K) s_slot[tid + (tid / K) * (KS - K)] = pslot;
            if (tid < nr) s_row[tid] = prow;
        }
        __syncthreads();
        if (g + step < gend) prefetch(g + step);  // lands while this tile is computed

        const int tasks = nr * W;
        for (int base = 0; base < tasks; base += TASKS_PER_PASS) {
            const int tau = base + wave * 62 + lane - 1;  // lanes 0 and 63: halo tasks
            const bool valid = tau >= 0 && tau < tasks;
            const int r = !valid ? 0 : (W == 1 ? tau : (int)__umulhi((uint32_t)tau, a.magic_W));  // tau / W
            const int p = tau - r * W;
            const bool g_lo = GHOST && p == 0, g_hi = GHOST && p == W - 1;
            const bool ghost = g_lo || g_hi;
            const int pp = p - (GHOST ? 1 : 0);  // own pair index
            const int t0 = 2 * pp;
            const bool has1 = t0 + 1 < a.n_loc;
            // a lane's 16 bytes of local column j: own pairs s_x[j*P + pp], ghosts s_g[j]
            const double2 *base_lane = ghost ? s_g : s_x + pp;
            const int stride_lane = ghost ? 1 : P;

            double acc0[NT], acc1[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k) acc0[k] = acc1[k] = 0.0;
            if (valid) {
                uint16_t sl[KS];
                const uint4 *so = reinterpret_cast<const uint4 *>(s_slot + r * KS);
#pragma unroll
                for (int u = 0; u < KS / 8; ++u) {
                    const uint4 v = so[u];
                    sl[8 * u] = v.x & 0xffff, sl[8 * u + 1] = v.x >> 16;
                    sl[8 * u + 2] = v.y & 0xffff, sl[8 * u + 3] = v.y >> 16;
                    sl[8 * u + 4] = v.z & 0xffff, sl[8 * u + 5] = v.z >> 16;
                    sl[8 * u + 6] = v.w & 0xffff, sl[8 * u + 7] = v.w >> 16;
                }
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    const double2 xv = base_lane[(sl[u] & 127) * stride_lane];
                    const double *dv = s_dict + (sl[u] >> 7) * NT;
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
                        acc0[k] = fma(v[k], xv.x, acc0[k]);
                        acc1[k] = fma(v[k], xv.y, acc1[k]);
                    }
                }
            }
            // What a lane shows its neighbours: to the right its last step (acc1),
            // to the left its first (acc0).  The ghost task before the row shows
            // z[-1] = the x_lo sums (acc0 of a ghost pair) to its right; the one
            // behind it shows z[n_loc] = the x_hi sums (acc1) to its left.
            double y0 = 0.0, y1 = 0.0;
            if (a.any_tri) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const double show_r = g_lo ? acc0[k] : acc1[k];
                    const double show_l = g_hi ? acc1[k] : acc0[k];
                    double zl = dpp_from_prev(show_r);  // z[t0 - 1]
                    double zr = dpp_from_next(show_l);  // z[t0 + 2] (z[t0 + 1] on an odd tail)
                    if (p == 0) zl = 0.0;      // nothing before the row
                    if (p == W - 1) zr = 0.0;  // nothing behind it
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
            if (valid && !ghost && lane >= 1 && lane <= 62) {
                if (!has1) y1 = 0.0;  // padding slot stays zero
                double2 *dst = reinterpret_cast<double2 *>(a.y + (size_t)s_row[r] * a.ld + t0);
                if (a.beta != 0.0) {
                    const double2 old = *dst;
                    y0 = fma(a.beta, old.x, y0);
                    if (has1) y1 = fma(a.beta, old.y, y1);
                }
                if (a.flags & 1) {
                    stk_v2d out;
                    out.x = y0, out.y = y1;
                    __builtin_nontemporal_store(out, reinterpret_cast<stk_v2d *>(dst));
                } else {
                    *dst = make_double2(y0, y1);
                }
            }
        }
    }
}

int g_tile_wg_per_cu = 0;
int g_tile_flags = 0;

template <int NT, int K>
int launch(hipStream_t st, const TileArgs<NT> &a, bool ghost)
{
    const int KS = (K + 7) & ~7;
    const int LT = (a.n_loc + 2) & ~1;
    const size_t lds = 16 * ((size_t)a.nc_max * a.P + (ghost ? a.nc_max : 0)) + 2 * (size_t)a.tr_max * KS +
                       4 * (size_t)((a.tr_max + 3) & ~3) + 8 * ((size_t)NT * 3 * LT + (size_t)a.n_codes * NT) + 32;
    STK_REQUIRE(lds <= 64 * 1024, "stk_kron_tile_apply: %zu bytes of LDS per workgroup; plan smaller tiles", lds);
    const int n_cu = stk_cu_count();
    const bool small = (int64_t)a.nc_max * a.P <= 3 * BS;  // the tile fits three pieces per thread
    int per_cu = g_tile_wg_per_cu > 0 ? g_tile_wg_per_cu : (small ? 3 : 2);
    const int by_lds = (int)(160 * 1024 / (lds + 256));
    if (per_cu > by_lds) per_cu = by_lds > 0 ? by_lds : 1;
    int per_xcd = (n_cu / 8) * per_cu;
    if (per_xcd > a.chunk) per_xcd = a.chunk;
    if (per_xcd < 1) per_xcd = 1;
    const unsigned grid = (unsigned)per_xcd * 8;
    if (ghost && small)
        hipLaunchKernelGGL((kron_tile_kernel<NT, K, true, 3>), dim3(grid), dim3(BS), lds, st, a);
    else if (ghost)
        hipLaunchKernelGGL((kron_tile_kernel<NT, K, true, 6>), dim3(grid), dim3(BS), lds, st, a);
    else if (small)
        hipLaunchKernelGGL((kron_tile_kernel<NT, K, false, 3>), dim3(grid), dim3(BS), lds, st, a);
    else
        hipLaunchKernelGGL((kron_tile_kernel<NT, K, false, 6>), dim3(grid), dim3(BS), lds, st, a);
    STK_LAUNCH_CHECK();
    return 0;
}

template <int NT>
int dispatch(hipStream_t st, const stk_tile_pattern *pat, int32_t n_loc, int32_t ld, const stk_kron_pack_term *t,
             const double *x, const double *gh, double beta, double *y)
{
    TileArgs<NT> a;
    a.tile_row_ptr = pat->tile_row_ptr;
    a.tile_col_ptr = pat->tile_col_ptr;
    a.tile_cols = pat->tile_cols;
    a.slots = pat->slots;
    a.row_ids = pat->row_ids;
    a.x = x;
    a.gh = gh;
    a.y = y;
    a.beta = beta;
    a.M = pat->M;
    a.n_loc = n_loc;
    a.ld = ld;
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
    a.magic_P = a.P > 1 ? (uint32_t)((((uint64_t)1 << 32) + a.P - 1) / a.P) : 0u;
    a.n_tiles = pat->n_tiles;
    a.chunk = (pat->n_tiles + 7) / 8;
    a.nc_max = pat->nc_max;
    a.tr_max = pat->tr_max;
    a.flags = g_tile_flags;
    STK_REQUIRE((int64_t)pat->nc_max * a.P <= (int64_t)6 * BS,
                "stk_kron_tile_apply: a tile of %d columns x %d pairs exceeds the %d chunks a workgroup stages",
                pat->nc_max, a.P, 6 * BS);
    STK_REQUIRE(pat->nc_max <= BS && pat->tr_max <= BS && pat->tr_max * pat->K <= BS,
                "stk_kron_tile_apply: tile too large (columns %d, rows %d)", pat->nc_max, pat->tr_max);
    STK_REQUIRE((int64_t)pat->tr_max * a.W < 65536, "stk_kron_tile_apply: too many tasks per tile");
    switch (pat->K) {
        case 5: return launch<NT, 5>(st, a, gh != nullptr);
        case 7: return launch<NT, 7>(st, a, gh != nullptr);
        case 9: return launch<NT, 9>(st, a, gh != nullptr);
        case 12: return launch<NT, 12>(st, a, gh != nullptr);
        case 16: return launch<NT, 16>(st, a, gh != nullptr);
    }
    stk_set_error("stk_kron_tile_apply: K=%d is not one of 5, 7, 9, 12, 16", pat->K);
    return 2;
}

}  // namespace

int stk_kron_tile_set_tuning(const char *key, int32_t value)
{
    if (std::strcmp(key, "tile_wg_per_cu") == 0) {
        g_tile_wg_per_cu = value;
        return 0;
    }
    if (std::strcmp(key, "tile_flags") == 0) {
        g_tile_flags = value;
        return 0;
    }
    return 1;
}

extern "C" int stk_kron_tile_apply(void *stream, const stk_tile_pattern *pat, int32_t n_loc, int32_t ld,
                                   int32_t n_terms, const stk_kron_pack_term *t, const double *x,
                                   const double *ghosts, double beta, double *y)
{
    STK_REQUIRE(pat && t && x && y, "stk_kron_tile_apply: null pointer");
    STK_REQUIRE(pat->M > 0 && pat->K >= 1 && pat->n_tiles > 0 && pat->slots && pat->dict && pat->tile_row_ptr &&
                    pat->tile_col_ptr && pat->tile_cols,
                "stk_kron_tile_apply: bad pattern");
    STK_REQUIRE(pat->nc_max >= 1 && pat->nc_max <= 128, "stk_kron_tile_apply: nc_max=%d not in 1..128", pat->nc_max);
    STK_REQUIRE(pat->n_codes >= 1 && pat->n_codes <= 512, "stk_kron_tile_apply: n_codes=%d not in 1..512",
                pat->n_codes);
    STK_REQUIRE(n_loc > 0 && ld >= n_loc && (ld & 1) == 0,
                "stk_kron_tile_apply: bad sizes n_loc=%d ld=%d (ld must be even)", n_loc, ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_tile_apply: n_terms=%d not in 1..3", n_terms);
    STK_REQUIRE(x != y, "stk_kron_tile_apply: input aliases output");
    STK_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)ghosts) & 15) == 0,
                "stk_kron_tile_apply: x, y and ghosts must be 16-byte aligned");
    for (int k = 0; k < n_terms; ++k)
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < pat->n_mats, "stk_kron_tile_apply: term %d names matrix %d of %d", k,
                    t[k].mat, pat->n_mats);
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return dispatch<1>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
        case 2: return dispatch<2>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
        default: return dispatch<3>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
    }
}
