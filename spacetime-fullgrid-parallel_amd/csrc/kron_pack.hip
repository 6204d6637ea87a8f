// Kronecker-sum apply on a PACKED sliced-ELL pattern:
//     y = beta*y + sum_k (T_k kron X_k) x          (all terms read the same x)
//
// Same operator as stk_kron_ell_apply (reference source/mpi_kron.py:77-90,
// 186-201, 214-219), same walk over the rows (persistent workgroups, XCD-
// interleaved row groups in a mesh-tile order, one lane per pair of time steps,
// all K gathers of a lane issued back to back, next group's entries prefetched
// into registers).  What differs is what the kernel streams and how it sees the
// halo:
//
//  * Matrix stream: ONE 32-bit word per slot, `code << col_bits | col`.  `code`
//    indexes a dictionary of the distinct value tuples (X_0[i,j], X_1[i,j], ..)
//    of the union pattern, kept in LDS.  P1 matrices on uniformly refined meshes
//    have a handful of distinct entries (3 tuples for (M_x, A_x) on the square
//    and the L-shape, 11 on the cube), so the 20 bytes per slot of the plain
//    form (index + two doubles) shrink to 4: at J_space = 9 the matrix stream of
//    one apply is 29 MB instead of 146 MB.  The values are the exact doubles, so
//    results are bit-identical with the plain form.  The host falls back to
//    stk_kron_ell_apply when the tuples do not fit the code bits (general
//    unstructured meshes) or a row is longer than K.
//  * Ghost time rows (a slab with neighbour ranks; the reference's X_loc_bdr[0]
//    and X_loc_bdr[-1], mpi_vector.py:148-175) arrive interleaved, gh[j] =
//    (x_lo[j], x_hi[j]), and every row gets ONE extra lane that runs the very
//    same instruction stream on them (its gathers are 16 bytes like everyone
//    else's, from `gh + col*16` instead of `x + col*ld*8 + t0*8`): the space
//    factors of the two ghost steps come out of the same pass over the matrix,
//    and the time stencil picks them up from LDS like any other step.  No second
//    kernel, no second pass over the matrices, no read-modify-write of y.
//    (Lanes that took a different BRANCH for the ghost rows doubled the time of
//    an earlier version; here the lanes differ only in a base and a stride.)
//  * Addresses are 64-bit (`base_lane + col * stride_lane`, one v_mad_u64_u32 per
//    gather), so slabs of 4 GiB and more need no separate instantiation.
//  * Row PAIRS (RP = 2): a slot row may serve two matrix rows that share columns
//    (mesh neighbours: 4 of their 7 columns on a P1 triangulation).  The lane
//    gathers the UNION of their columns once -- 10 gathers for two rows instead of
//    14 -- and every dictionary entry carries the values of both rows (zero where
//    a row has no entry in that column).  The time of this kernel grows by ~6 %
//    per gather instruction of a lane; the columns of a
//    row keep their ascending order inside the union, so every row's sum is
//    accumulated in the same order as in the one-row form (an absent column adds
//    0 * x = 0 exactly) and the results stay bit-identical.  (The kernel is written
//    for RP rows; three and four rows per slot row -- 4.3 and 4 gathers per row --
//    were measured slower than pairs, 0.322 and 0.576 ms against 0.301 ms, for
//    their registers and LDS, and are not instantiated:
//    profiles/r02_rows_per_unit.log.)
//  * EXPLICIT VALUES (DICT = false, row pairs only): matrices whose entries do not
//    repeat -- an unstructured mesh, reference mpi_kron.py:135-150 takes any CSR --
//    have no dictionary.  Pairs need shared COLUMNS, not repeated values: the slot
//    word is then the column alone and the values of both rows of every slot travel
//    beside it, vals[unit][slot][row][term] (zero where a row has no entry in the
//    column; the matrices of the call's terms, in term order), prefetched into
//    registers and handed over through LDS like the slot words.  36 bytes per slot instead of 20 in the one-row plain form
//    (kron_ell.hip), but 10 gathers for two rows instead of 14; accumulation order
//    and results are those of the plain form, bit for bit.
//  * INPUTS PER TERM (MULTI = true, stk_kron_pack_apply_multi): term k gathers from
//    a slab of its own, xk[k] -- the last stage of the regrouped Schur complement,
//    (I kron M_x) v1 + (I kron A_x) v2 + (G_t kron M_x) x (reference
//    heateq_mpi.py:166-181), which the plain form (kron_ell.hip, 20 bytes per slot,
//    7 gathers per row and term) ran at 0.28 of the HBM peak.  The terms take turns
//    on the one packed slot stream: K gathers from xk[k], then the sums of term k.
//    A lane skips the turn of a term whose time factor never multiplies its pair of
//    time steps (G_t has the one entry (0, 0): only the first lane of a row gathers
//    x), so a term costs the traffic of the time steps it really reads.  The sums are
//    accumulated in the order of the one-input form, term by term.
#include <cstring>
#include <vector>

#include "stk_common.h"

namespace {

template <int NT>
struct PackArgs {
    const uint32_t *slots;   // [n_units][K]
    const int32_t *row_ids;  // [n_units][RP] (-1: no row) or NULL (RP = 1, index order)
    const double *dict[NT];  // [n_codes][RP] values of term k's matrix per code
    const double *tri[NT];   // [3][n_loc] or NULL
    const double *x;
    const double *gh;  // [M][2] interleaved ghost steps (t = -1, t = n_loc) or NULL
    double *y;
    double beta;
    int32_t M, n_units, n_loc, ld;
    int32_t any_tri;
    int32_t P, W, R;         // own lanes per unit, lanes per unit, units per group
    int32_t ngroups, chunk;  // groups in total / per XCD
    int32_t col_bits, n_codes;
    int32_t flags;  // bit 0: non-temporal y stores, bit 1: non-temporal slot loads
    unsigned long long *diag;  // DIAG instantiation: [waves][4] cycle sums
    // explicit values (no dictionary): [n_units][K][RP][NT], term k reads entry k
    const double *vals;
    int32_t n_mats;
    int32_t mat[NT];
    const double *xk[NT];  // MULTI: the input slab of term k
};

typedef double stk_v2d __attribute__((ext_vector_type(2)));

__device__ inline double2 load2(const char *p) { return *reinterpret_cast<const double2 *>(p); }

// One stamp of the shader clock, ordered against the instruction stream (diagnostic
// builds only; see the DIAG parameter below).
__device__ inline unsigned long long stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// DIAG = true is a separate diagnostic instantiation (tools/kron_ab.py --phases): every
// wave sums the shader-clock cycles it spends in the four segments of an
// iteration -- publish + barrier, gathers + space factors, exchange + barrier,
// time stencil + store -- into a.diag[wave][4].  Its outputs are still correct;
// its run time is not quoted anywhere.
template <int NT, int K, int NPF, bool GHOST, int BS, bool DIAG, int RP, bool DICT = true, bool MULTI = false>
__global__ __launch_bounds__(BS, (K >= 12 || RP > 1) ? 4 : 6) void kron_pack_kernel(const PackArgs<NT> a)
{
    static_assert(!MULTI || (!GHOST && !DIAG && DICT), "inputs per term: no ghost lanes, dictionary form");
    constexpr int NPV = DICT ? 1 : 2;  // explicit values prefetched per thread (R is sized for it)
    constexpr int KS = (K + 3) & ~3;  // LDS stride of a row's slots (16-byte vectors)
    extern __shared__ double sm[];
    const int W = a.W, R = a.R, SW = a.n_loc + 3;
    // slot words first: their rows are read as 16-byte vectors
    uint32_t *s_slot = reinterpret_cast<uint32_t *>(sm);  // [R][KS]
    int32_t *s_row = reinterpret_cast<int32_t *>(s_slot + R * KS);  // [R][RP]
    const int LT = (a.n_loc + 2) & ~1;
    double *s_tri = reinterpret_cast<double *>(s_row + ((R * RP + 3) & ~3));  // [NT][3][LT], 16-byte aligned rows
    double *s_dict = s_tri + NT * 3 * LT;  // [n_codes][RP][NT], or the group's explicit values [R][K][RP][n_mats]
    // s_w[k][r * RP + j][q], q = t + 1: the space-factor results z_k[row][t], t = -1 .. n_loc
    double *s_w = s_dict + (DICT ? a.n_codes * RP * NT : R * K * RP * NT);

    const int tid = threadIdx.x;
    const int r = tid / W;
    const int p = tid - r * W;
    const bool in_row = r < R;
    const bool ghost_lane = GHOST && p == a.P;
    const int t0 = 2 * p;  // own lanes: first time step of the pair
    const bool has1 = t0 + 1 < a.n_loc;
    // what distinguishes the lanes of a row: where their 16 bytes of a column start
    const char *base_lane = ghost_lane ? reinterpret_cast<const char *>(a.gh)
                                       : reinterpret_cast<const char *>(a.x) + (size_t)t0 * 8;
    const uint32_t stride_lane = ghost_lane ? 16u : (uint32_t)a.ld * 8u;
    const uint32_t col_mask = (1u << a.col_bits) - 1u;
    // where a lane leaves its two sums in s_w (index q = t + 1)
    const int wq0 = ghost_lane ? 0 : t0 + 1;
    const int wdq = ghost_lane ? a.n_loc + 1 : 1;
    const bool wr1 = ghost_lane || has1;

    if constexpr (DICT) {
        for (int i = tid; i < a.n_codes * RP * NT; i += BS) {
            const int c = i / NT, k = i - c * NT;  // c = code * RP + row of the pair
            s_dict[i] = a.dict[k][c];
        }
    }
    if (a.any_tri) {
        for (int i = tid; i < NT * 3 * LT; i += BS) {
            const int k = i / (3 * LT), rem = i - k * 3 * LT;
            const int d = rem / LT, t = rem - d * LT;
            s_tri[i] = (a.tri[k] != nullptr && t < a.n_loc) ? a.tri[k][d * a.n_loc + t] : 0.0;
        }
        if (!GHOST) {  // z[-1] and z[n_loc] of every row: zero, never rewritten
            for (int i = tid; i < NT * R * RP; i += BS) {
                s_w[i * SW] = 0.0;
                s_w[i * SW + a.n_loc + 1] = 0.0;
            }
        }
    }

    // MULTI: does term k's time factor multiply z_k at this lane's time steps at
    // all?  z[t] enters the output through sub[t + 1], dia[t] and super[t - 1].
    bool need[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) need[k] = true;
    if constexpr (MULTI) {
        if (a.any_tri) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (a.tri[k] == nullptr) continue;
                const double *c = s_tri + k * 3 * LT;
                bool used = false;
                for (int t = t0; t < min(t0 + 2, a.n_loc); ++t) {
                    used = used || c[LT + t] != 0.0;
                    if (t + 1 < a.n_loc) used = used || c[t + 1] != 0.0;
                    if (t >= 1) used = used || c[2 * LT + t - 1] != 0.0;
                }
                need[k] = used;
            }
        }
    }

    // groups of this workgroup: interleaved with the other workgroups of its XCD
    const int xcd = blockIdx.x & 7;
    const int step = gridDim.x >> 3;
    const int gend = min((xcd + 1) * a.chunk, a.ngroups);
    int g = xcd * a.chunk + (int)(blockIdx.x >> 3);

    uint32_t pslot[NPF];
    double pval[NPV];
    int32_t prow = 0;
#pragma unroll
    for (int q = 0; q < NPF; ++q) pslot[q] = 0;
#pragma unroll
    for (int q = 0; q < NPV; ++q) pval[q] = 0.0;
    constexpr int vper = K * RP * NT;  // explicit values of one slot row (the terms' matrices, in term order)
    auto fetch = [&](int gq) {
        const int rows = min(R, a.n_units - gq * R);
        const uint32_t *src = a.slots + (size_t)gq * R * K;
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            if (i < rows * K) pslot[q] = (a.flags & 2) ? __builtin_nontemporal_load(src + i) : src[i];
        }
        if constexpr (!DICT) {
            const double *vsrc = a.vals + (size_t)gq * R * vper;
#pragma unroll
            for (int q = 0; q < NPV; ++q) {
                const int i = tid + q * BS;
                if (i < rows * vper) pval[q] = (a.flags & 2) ? __builtin_nontemporal_load(vsrc + i) : vsrc[i];
            }
        }
        if (tid < rows * RP) prow = a.row_ids ? a.row_ids[(size_t)gq * R * RP + tid] : gq * R + tid;
    };
    if (g < gend) fetch(g);

    unsigned long long seg[4] = {0, 0, 0, 0}, ts = 0;
    for (; g < gend; g += step) {
        if (DIAG) ts = stamp();
        const int rows = min(R, a.n_units - g * R);
        // ---- publish this group's entries ----------------------------------
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int i = tid + q * BS;
            // element i of the group's flat [rows][K] chunk -> LDS [row][KS]
            if (i < rows * K) s_slot[i + (i / K) * (KS - K)] = pslot[q];
        }
        if constexpr (!DICT) {
#pragma unroll
            for (int q = 0; q < NPV; ++q) {
                const int i = tid + q * BS;
                if (i < rows * vper) s_dict[i] = pval[q];
            }
        }
        if (tid < rows * RP) s_row[tid] = prow;
        __syncthreads();
        if (DIAG) {
            const unsigned long long t = stamp();
            seg[0] += t - ts, ts = t;
        }
        if (g + step < gend) fetch(g + step);  // in flight behind the gathers

        const bool active = in_row && r < rows;
        // read now: s_row is rewritten at the top of the next iteration, which a
        // fast wave reaches while a slow one is still storing
        int32_t yrow[RP];
#pragma unroll
        for (int j = 0; j < RP; ++j) yrow[j] = active ? s_row[r * RP + j] : -1;
        double acc0[RP][NT], acc1[RP][NT];
#pragma unroll
        for (int j = 0; j < RP; ++j)
#pragma unroll
            for (int k = 0; k < NT; ++k) acc0[j][k] = acc1[j][k] = 0.0;

        if (MULTI && active) {
            // the terms take turns: K gathers from term k's own slab, then its sums
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (!need[k]) continue;
                const char *base_k = reinterpret_cast<const char *>(a.xk[k]) + (size_t)t0 * 8;
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
                    for (int u = 0; u < K; ++u) xv[u] = load2(base_k + (size_t)(sl[u] & col_mask) * stride_lane);
                }
                asm volatile("" : "+v"(ro));  // the codes are read again: see below
                uint32_t sl[KS];
                const uint4 *so = reinterpret_cast<const uint4 *>(s_slot + ro);
#pragma unroll
                for (int u = 0; u < KS / 4; ++u) {
                    const uint4 v = so[u];
                    sl[4 * u] = v.x, sl[4 * u + 1] = v.y, sl[4 * u + 2] = v.z, sl[4 * u + 3] = v.w;
                }
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    const double *dv = s_dict + (sl[u] >> a.col_bits) * (RP * NT) + k;
#pragma unroll
                    for (int j = 0; j < RP; ++j, dv += NT) {
                        const double v = dv[0];
                        acc0[j][k] = fma(v, xv[u].x, acc0[j][k]);
                        acc1[j][k] = fma(v, xv[u].y, acc1[j][k]);
                    }
                }
            }
        }
        if (!MULTI && active) {
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
                if constexpr (DIAG) {
                    // ablations of the diagnostic build (wrong results, timing only):
                    // 16: every slot gathers the row's own column (one line set per row,
                    //     same instruction count); 32: one gather per lane instead of K
                    if (a.flags & 16) {
#pragma unroll
                        for (int u = 0; u < K; ++u) sl[u] = (uint32_t)yrow[0];
                    }
                }
                if (DIAG && (a.flags & 32)) {
                    xv[0] = load2(base_lane + (size_t)(sl[0] & col_mask) * stride_lane);
#pragma unroll
                    for (int u = 1; u < K; ++u) xv[u] = xv[0];
                } else {
#pragma unroll
                    for (int u = 0; u < K; ++u)
                        xv[u] = load2(base_lane + (size_t)(sl[u] & col_mask) * stride_lane);
                }
            }
            // The slot words are read a second time for their codes rather than
            // kept in registers across the gathers (the offset is made opaque so
            // that the compiler does not merge the two reads): 8 VGPRs less while
            // all K loads are in flight.
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
                const double *dv = DICT ? s_dict + (sl[u] >> a.col_bits) * (RP * NT)
                                        : s_dict + ((r * K + u) * RP) * NT;
#pragma unroll
                for (int j = 0; j < RP; ++j, dv += NT) {
                    double v[NT];
                    if constexpr (NT == 2) {  // one 16-byte read (s_dict is 16-byte aligned)
                        const double2 vv = *reinterpret_cast<const double2 *>(dv);
                        v[0] = vv.x, v[1] = vv.y;
                    } else {
#pragma unroll
                        for (int k = 0; k < NT; ++k) v[k] = dv[k];
                    }
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        acc0[j][k] = fma(v[k], xv[u].x, acc0[j][k]);
                        acc1[j][k] = fma(v[k], xv[u].y, acc1[j][k]);
                    }
                }
            }
        }

        if (DIAG) {
            // the sums must exist before the stamp: make them opaque to the scheduler
#pragma unroll
            for (int j = 0; j < RP; ++j)
#pragma unroll
                for (int k = 0; k < NT; ++k) asm volatile("" : "+v"(acc0[j][k]), "+v"(acc1[j][k]));
            const unsigned long long t = stamp();
            seg[1] += t - ts, ts = t;
        }
        // ---- time stencil through LDS, store ---------------------------------
        double y0[RP], y1[RP];
#pragma unroll
        for (int j = 0; j < RP; ++j) y0[j] = y1[j] = 0.0;
        if (a.any_tri) {
            if (active) {
                // own lanes: z[t0], z[t0 + 1]; the ghost lane: z[-1], z[n_loc]
#pragma unroll
                for (int j = 0; j < RP; ++j) {
                    double *w = s_w + (r * RP + j) * SW + wq0;
#pragma unroll
                    for (int k = 0; k < NT; ++k, w += R * RP * SW) {
                        w[0] = acc0[j][k];
                        if (wr1) w[wdq] = acc1[j][k];
                    }
                }
            }
            __syncthreads();
            if (DIAG) {
                const unsigned long long t = stamp();
                seg[2] += t - ts, ts = t;
            }
            if (active && !ghost_lane) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    if (a.tri[k] != nullptr) {
                        const double *c = s_tri + k * 3 * LT + t0;
                        const double2 sub = *reinterpret_cast<const double2 *>(c);
                        const double2 dia = *reinterpret_cast<const double2 *>(c + LT);
                        const double2 sup = *reinterpret_cast<const double2 *>(c + 2 * LT);
#pragma unroll
                        for (int j = 0; j < RP; ++j) {
                            const double *w = s_w + ((k * R + r) * RP + j) * SW + t0;  // w[q]: z at step t0 - 1 + q
                            double v0 = dia.x * acc0[j][k];
                            v0 = fma(sub.x, w[0], v0);
                            v0 = fma(sup.x, has1 ? acc1[j][k] : w[2], v0);
                            y0[j] += v0;
                            if (has1) {
                                double v1 = dia.y * acc1[j][k];
                                v1 = fma(sub.y, acc0[j][k], v1);
                                v1 = fma(sup.y, w[3], v1);
                                y1[j] += v1;
                            }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < RP; ++j) {
                            y0[j] += acc0[j][k];
                            y1[j] += acc1[j][k];
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < RP; ++j)
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    y0[j] += acc0[j][k];
                    y1[j] += acc1[j][k];
                }
            __syncthreads();  // the LDS entries are rewritten at the top of the loop
        }
        if (active && !ghost_lane) {
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                if (RP > 1 && yrow[j] < 0) continue;  // a slot row that serves one matrix row only
                if (!has1) y1[j] = 0.0;              // padding slot stays zero
                double2 *dst = reinterpret_cast<double2 *>(
                    reinterpret_cast<char *>(a.y) + (size_t)(uint32_t)yrow[j] * ((size_t)a.ld * 8) + (size_t)t0 * 8);
                if (a.beta != 0.0) {
                    const double2 old = *dst;
                    y0[j] = fma(a.beta, old.x, y0[j]);
                    if (has1) y1[j] = fma(a.beta, old.y, y1[j]);
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
        if (DIAG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // charge the store to this segment
            const unsigned long long t = stamp();
            seg[3] += t - ts;
        }
    }
    if (DIAG && a.diag != nullptr && (tid & 63) == 0) {
        unsigned long long *d = a.diag + ((size_t)blockIdx.x * (BS / 64) + (tid >> 6)) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = seg[q];
    }
}

// The first and last local time step once the halo is there, after the main pass
// ran WITHOUT the ghost steps (GHOST = false, beta = 0) while the exchange was still
// in flight -- the reference overlaps the exchange with the rows that do not need
// it the same way, mpi_kron.py:193-200.  The two boundary steps are RECOMPUTED in
// the arithmetic order of the one-pass kernel above,
//     y[i][t] = sum_k fma(super_k[t], z_k[i][t+1], fma(sub_k[t], z_k[i][t-1], dia_k[t] * z_k[i][t])),
//     z_k[i][s] = the slot-ordered fma chain of X_k's row i over time column s
// (s = -1: x_lo, s = n_loc: x_hi), and overwrite what the pass left there, so that
// a slab boundary leaves no trace in the result: the apply is bit for bit the
// one-rank apply wherever the slabs are cut.  (Adding the ghost steps' share to the
// pass's value afterwards -- rounds 1-5 -- put the neighbour's term last in the
// sum, an order no interior row has.)  One lane per slot row and side on the
// packed stream: three 8-byte gathers per slot (the step's own column, its inner
// neighbour column, the received row).
template <int NT, int K, int RP>
__global__ __launch_bounds__(256) void kron_pack_ghost_kernel(const PackArgs<NT> a, const double *__restrict__ lo,
                                                              const double *__restrict__ hi, int side0)
{
    extern __shared__ double sm[];
    double *s_dict = sm;  // [n_codes][RP][NT]
    if (a.vals == nullptr) {
        for (int i = threadIdx.x; i < a.n_codes * RP * NT; i += 256) {
            const int c = i / NT, k = i - c * NT;
            s_dict[i] = a.dict[k][c];
        }
    }
    __syncthreads();
    const int side = side0 + (int)blockIdx.y;         // 0: first local step, 1: last
    const int t = side ? a.n_loc - 1 : 0;
    // where z[t-1], z[t], z[t+1] gather from: a received row (stride 8) or a time column of the slab
    const size_t col_stride = (size_t)a.ld * 8;
    const char *xb = reinterpret_cast<const char *>(a.x);
    const char *src_m = t > 0 ? xb + (size_t)(t - 1) * 8 : reinterpret_cast<const char *>(lo);
    const size_t str_m = t > 0 ? col_stride : 8;
    const char *src_c = xb + (size_t)t * 8;
    const char *src_p = t + 1 < a.n_loc ? xb + (size_t)(t + 1) * 8 : reinterpret_cast<const char *>(hi);
    const size_t str_p = t + 1 < a.n_loc ? col_stride : 8;
    double sub[NT], dia[NT], sup[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        sub[k] = a.tri[k] ? a.tri[k][t] : 0.0;
        dia[k] = a.tri[k] ? a.tri[k][a.n_loc + t] : 1.0;
        sup[k] = a.tri[k] ? a.tri[k][2 * a.n_loc + t] : 0.0;
    }
    const uint32_t col_mask = (1u << a.col_bits) - 1u;
    const int stride = gridDim.x * 256;
    for (int u = blockIdx.x * 256 + threadIdx.x; u < a.n_units; u += stride) {
        uint32_t sl[K];
#pragma unroll
        for (int q = 0; q < K; ++q) sl[q] = a.slots[(size_t)u * K + q];
        double xm[K], xc[K], xp[K];
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const size_t col = sl[q] & col_mask;
            xm[q] = src_m ? *reinterpret_cast<const double *>(src_m + col * str_m) : 0.0;
            xc[q] = *reinterpret_cast<const double *>(src_c + col * col_stride);
            xp[q] = src_p ? *reinterpret_cast<const double *>(src_p + col * str_p) : 0.0;
        }
        double zm[RP][NT], zc[RP][NT], zp[RP][NT];
#pragma unroll
        for (int j = 0; j < RP; ++j)
#pragma unroll
            for (int k = 0; k < NT; ++k) zm[j][k] = zc[j][k] = zp[j][k] = 0.0;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const double *dv = a.vals ? a.vals + ((size_t)u * K + q) * RP * NT
                                      : s_dict + (sl[q] >> a.col_bits) * (RP * NT);
#pragma unroll
            for (int j = 0; j < RP; ++j)
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const double v = dv[j * NT + k];
                    zm[j][k] = fma(v, xm[q], zm[j][k]);
                    zc[j][k] = fma(v, xc[q], zc[j][k]);
                    zp[j][k] = fma(v, xp[q], zp[j][k]);
                }
        }
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            const int row = a.row_ids ? a.row_ids[(size_t)u * RP + j] : u;
            if (row < 0) continue;
            double yv = 0.0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (a.tri[k] != nullptr) {
                    double v = dia[k] * zc[j][k];
                    v = fma(sub[k], zm[j][k], v);
                    v = fma(sup[k], zp[j][k], v);
                    yv += v;
                } else {
                    yv += zc[j][k];
                }
            }
            a.y[(size_t)row * a.ld + t] = yv;
        }
    }
}

// The same two boundary steps from COMPACT operands (round 6): the pack kernel that
// extracts the rows a rank sends (stk_halo_pack_records) also leaves, per spatial dof j,
// rec[j] = (x[j][0], x[j][1], x[j][n_loc-2], x[j][n_loc-1]) -- 32 bytes -- and the received
// rows arrive interleaved, gh[j] = (x_lo[j], x_hi[j]).  A lane then serves BOTH sides of a
// slot row with three 16-byte loads per slot from two lines, where the kernel above issues
// six 8-byte gathers per slot (two lanes) over four to six lines of the slab: the gathers
// of this kernel are bound by their number, not by their bytes (0.069 / 0.130 / 0.079 ms at 9
// / 17 / 33 steps of 1 046 529 rows against the times in DESIGN.md section 4).  Same sums,
// same order: bit for bit the kernel above and the one-pass form.
template <int NT, int K, int RP>
__global__ __launch_bounds__(256) void kron_pack_boundary_kernel(const PackArgs<NT> a, const double4 *__restrict__ rec,
                                                                 const double2 *__restrict__ gh, int has_lo,
                                                                 int has_hi)
{
    extern __shared__ double sm[];
    double *s_dict = sm;  // [n_codes][RP][NT]
    if (a.vals == nullptr) {
        for (int i = threadIdx.x; i < a.n_codes * RP * NT; i += 256) {
            const int c = i / NT, k = i - c * NT;
            s_dict[i] = a.dict[k][c];
        }
    }
    __syncthreads();
    const int n = a.n_loc;
    double sub0[NT], dia0[NT], sup0[NT], sub1[NT], dia1[NT], sup1[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const double *t = a.tri[k];
        sub0[k] = t ? t[0] : 0.0, dia0[k] = t ? t[n] : 1.0, sup0[k] = t ? t[2 * n] : 0.0;
        sub1[k] = t ? t[n - 1] : 0.0, dia1[k] = t ? t[n + n - 1] : 1.0, sup1[k] = t ? t[2 * n + n - 1] : 0.0;
    }
    const uint32_t col_mask = (1u << a.col_bits) - 1u;
    const int stride = gridDim.x * 256;
    for (int u = blockIdx.x * 256 + threadIdx.x; u < a.n_units; u += stride) {
        uint32_t sl[K];
#pragma unroll
        for (int q = 0; q < K; ++q) sl[q] = a.slots[(size_t)u * K + q];
        // z[side-step][row of the pair][term]: 0 = lo, 1 = x0, 2 = x1, 3 = x_{n-2}, 4 = x_{n-1}, 5 = hi
        double z[6][RP][NT];
#pragma unroll
        for (int w = 0; w < 6; ++w)
#pragma unroll
            for (int j = 0; j < RP; ++j)
#pragma unroll
                for (int k = 0; k < NT; ++k) z[w][j][k] = 0.0;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const size_t col = sl[q] & col_mask;
            const double4 r = rec[col];
            const double2 g = gh[col];
            const double xs[6] = {g.x, r.x, r.y, r.z, r.w, g.y};
            const double *dv = a.vals ? a.vals + ((size_t)u * K + q) * RP * NT
                                      : s_dict + (sl[q] >> a.col_bits) * (RP * NT);
#pragma unroll
            for (int j = 0; j < RP; ++j)
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const double v = dv[j * NT + k];
#pragma unroll
                    for (int w = 0; w < 6; ++w) z[w][j][k] = fma(v, xs[w], z[w][j][k]);
                }
        }
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            const int row = a.row_ids ? a.row_ids[(size_t)u * RP + j] : u;
            if (row < 0) continue;
            double y_lo = 0.0, y_hi = 0.0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                if (a.tri[k] != nullptr) {
                    // one step: both received rows meet in it (z[5] is its upper neighbour)
                    double v = dia0[k] * z[1][j][k];
                    v = fma(sub0[k], z[0][j][k], v);
                    v = fma(sup0[k], n > 1 ? z[2][j][k] : z[5][j][k], v);
                    y_lo += v;
                    double w = dia1[k] * z[4][j][k];
                    w = fma(sub1[k], z[3][j][k], w);
                    w = fma(sup1[k], z[5][j][k], w);
                    y_hi += w;
                } else {
                    y_lo += z[1][j][k];
                    y_hi += z[4][j][k];
                }
            }
            double *yr = a.y + (size_t)row * a.ld;
            if (has_lo || n == 1) yr[0] = y_lo;
            if (has_hi && n > 1) yr[n - 1] = y_hi;
        }
    }
}

template <int NT, int RP>
int launch_boundary(hipStream_t st, const PackArgs<NT> &a, int K, const double *rec, const double *gh, int has_lo,
                    int has_hi)
{
    const size_t lds = sizeof(double) * (a.vals ? 0 : (size_t)a.n_codes * RP * NT) + 16;
    const unsigned grid = stk_flat_grid(a.n_units, 256);
#define STK_BOUNDARY_CASE(KK)                                                                                     \
    case KK:                                                                                                      \
        hipLaunchKernelGGL((kron_pack_boundary_kernel<NT, KK, RP>), dim3(grid), dim3(256), lds, st, a,            \
                           reinterpret_cast<const double4 *>(rec), reinterpret_cast<const double2 *>(gh), has_lo, \
                           has_hi);                                                                               \
        break;
    if constexpr (RP == 1) {
        switch (K) {
            STK_BOUNDARY_CASE(5)
            STK_BOUNDARY_CASE(7)
            STK_BOUNDARY_CASE(9)
            STK_BOUNDARY_CASE(12)
            STK_BOUNDARY_CASE(16)
            default: stk_set_error("stk_kron_pack_boundary_apply: K=%d is not one of 5, 7, 9, 12, 16", K); return 2;
        }
    } else {
        switch (K) {
            STK_BOUNDARY_CASE(8)
            STK_BOUNDARY_CASE(10)
            STK_BOUNDARY_CASE(12)
            default: stk_set_error("stk_kron_pack_boundary_apply: K=%d is not one of 8, 10, 12 (row pairs)", K); return 2;
        }
    }
#undef STK_BOUNDARY_CASE
    STK_LAUNCH_CHECK();
    return 0;
}

template <int NT, int RP>
int launch_ghost_only(hipStream_t st, const PackArgs<NT> &a, int K, const double *lo, const double *hi)
{
    const size_t lds = sizeof(double) * (a.vals ? 0 : (size_t)a.n_codes * RP * NT) + 16;
    const unsigned grid = stk_flat_grid(a.n_units, 256);
    // one step: both received rows meet in it; otherwise a side per received row
    const int side0 = (a.n_loc == 1 || lo) ? 0 : 1;
    const unsigned sides = (a.n_loc > 1 && lo && hi) ? 2u : 1u;
#define STK_GHOST_CASE(KK)                                                                                  \
    case KK:                                                                                                \
        hipLaunchKernelGGL((kron_pack_ghost_kernel<NT, KK, RP>), dim3(grid, sides), dim3(256), lds, st, a, lo, hi, \
                           side0);                                                                          \
        break;
    if constexpr (RP == 1) {
        switch (K) {
            STK_GHOST_CASE(5)
            STK_GHOST_CASE(7)
            STK_GHOST_CASE(9)
            STK_GHOST_CASE(12)
            STK_GHOST_CASE(16)
            default: stk_set_error("stk_kron_pack_ghost_apply: K=%d is not one of 5, 7, 9, 12, 16", K); return 2;
        }
    } else {
        switch (K) {
            STK_GHOST_CASE(8)
            STK_GHOST_CASE(10)
            STK_GHOST_CASE(12)
            default: stk_set_error("stk_kron_pack_ghost_apply: K=%d is not one of 8, 10, 12 (row pairs)", K); return 2;
        }
    }
#undef STK_GHOST_CASE
    STK_LAUNCH_CHECK();
    return 0;
}

template <int NT>
int dispatch_ghost_only(hipStream_t st, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld,
                        const stk_kron_pack_term *t, const double *x, const double *lo, const double *hi, double *y)
{
    PackArgs<NT> a;
    std::memset(&a, 0, sizeof(a));
    a.slots = pat->slots;
    a.row_ids = pat->row_ids;
    a.x = x;
    a.y = y;
    a.M = pat->M;
    a.n_units = pat->n_units;
    a.n_loc = n_loc;
    a.ld = ld;
    a.col_bits = pat->col_bits;
    a.n_codes = pat->n_codes;
    a.vals = pat->vals;
    a.n_mats = pat->n_mats;
    for (int k = 0; k < NT; ++k) {
        a.dict[k] = pat->vals ? nullptr : pat->dict + (size_t)t[k].mat * pat->n_codes * pat->rows_per_unit;
        a.tri[k] = t[k].tri;
        a.mat[k] = t[k].mat;
    }
    return pat->rows_per_unit == 2 ? launch_ghost_only<NT, 2>(st, a, pat->K, lo, hi)
                                   : launch_ghost_only<NT, 1>(st, a, pat->K, lo, hi);
}

template <int NT>
int dispatch_boundary(hipStream_t st, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld,
                      const stk_kron_pack_term *t, const double *rec, const double *gh, int has_lo, int has_hi,
                      double *y)
{
    PackArgs<NT> a;
    std::memset(&a, 0, sizeof(a));
    a.slots = pat->slots;
    a.row_ids = pat->row_ids;
    a.y = y;
    a.M = pat->M;
    a.n_units = pat->n_units;
    a.n_loc = n_loc;
    a.ld = ld;
    a.col_bits = pat->col_bits;
    a.n_codes = pat->n_codes;
    a.vals = pat->vals;
    a.n_mats = pat->n_mats;
    for (int k = 0; k < NT; ++k) {
        a.dict[k] = pat->vals ? nullptr : pat->dict + (size_t)t[k].mat * pat->n_codes * pat->rows_per_unit;
        a.tri[k] = t[k].tri;
        a.mat[k] = t[k].mat;
    }
    return pat->rows_per_unit == 2 ? launch_boundary<NT, 2>(st, a, pat->K, rec, gh, has_lo, has_hi)
                                   : launch_boundary<NT, 1>(st, a, pat->K, rec, gh, has_lo, has_hi);
}

__global__ __launch_bounds__(256) void interleave_ghosts_kernel(int32_t M, const double *__restrict__ lo,
                                                                const double *__restrict__ hi,
                                                                double2 *__restrict__ gh)
{
    const int stride = gridDim.x * 256;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += stride)
        gh[j] = make_double2(lo ? lo[j] : 0.0, hi ? hi[j] : 0.0);
}

int g_pack_wg_per_cu = 0;
int g_pack_flags = 3;  // non-temporal y stores and slot loads: measured 2-3 % faster at J_time = 6 / J_space = 9
int g_pack_block = 512;                   // threads per workgroup: 512 or 256
int g_pack_multi_lanes = 1;               // inputs per term: 0 = the terms take turns in one lane (round 4)
int g_pack_multi_wg_per_cu = 0;           // inputs per term: workgroups per CU (0: as the one-input form)
int g_pack_multi_r = 0;                   // inputs per term: cap on the slot rows of a group (0: none)
int g_pack_check_steps = 0;               // inputs per term: verify the stated time steps against the factors (tests)
unsigned long long *g_pack_diag = nullptr;  // set: the next headline-shape launch runs the DIAG instantiation

template <int NT, int K, bool GHOST, int BS, int RP>
int launch_npf(hipStream_t st, const PackArgs<NT> &a, unsigned grid, size_t lds)
{
    const int npf = (a.R * K + BS - 1) / BS;
    if constexpr (RP == 2) {
        if (a.vals != nullptr) {  // explicit values: pairs only (the one-row plain form is kron_ell.hip)
            if (npf <= 1)
                hipLaunchKernelGGL((kron_pack_kernel<NT, K, 1, GHOST, BS, false, RP, false>), dim3(grid), dim3(BS), lds,
                                   st, a);
            else if (npf <= 2)
                hipLaunchKernelGGL((kron_pack_kernel<NT, K, 2, GHOST, BS, false, RP, false>), dim3(grid), dim3(BS), lds,
                                   st, a);
            else
                hipLaunchKernelGGL((kron_pack_kernel<NT, K, 4, GHOST, BS, false, RP, false>), dim3(grid), dim3(BS), lds,
                                   st, a);
            STK_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr (!GHOST && NT >= 2 && BS == 512) {
        if (a.xk[0] != nullptr) {  // inputs per term (dictionary form only: checked by the entry point)
            if (npf <= 1)
                hipLaunchKernelGGL((kron_pack_kernel<NT, K, 1, false, BS, false, RP, true, true>), dim3(grid), dim3(BS),
                                   lds, st, a);
            else if (npf <= 2)
                hipLaunchKernelGGL((kron_pack_kernel<NT, K, 2, false, BS, false, RP, true, true>), dim3(grid), dim3(BS),
                                   lds, st, a);
            else
                hipLaunchKernelGGL((kron_pack_kernel<NT, K, 4, false, BS, false, RP, true, true>), dim3(grid), dim3(BS),
                                   lds, st, a);
            STK_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr (NT == 2 && K == 7 && !GHOST && BS == 512 && RP == 1) {
        if (a.diag != nullptr && npf <= 1) {
            hipLaunchKernelGGL((kron_pack_kernel<NT, K, 1, GHOST, BS, true, RP>), dim3(grid), dim3(BS), lds, st, a);
            STK_LAUNCH_CHECK();
            return 0;
        }
    }
    if (npf <= 1)
        hipLaunchKernelGGL((kron_pack_kernel<NT, K, 1, GHOST, BS, false, RP>), dim3(grid), dim3(BS), lds, st, a);
    else if (npf <= 2)
        hipLaunchKernelGGL((kron_pack_kernel<NT, K, 2, GHOST, BS, false, RP>), dim3(grid), dim3(BS), lds, st, a);
    else
        hipLaunchKernelGGL((kron_pack_kernel<NT, K, 4, GHOST, BS, false, RP>), dim3(grid), dim3(BS), lds, st, a);
    STK_LAUNCH_CHECK();
    return 0;
}

template <int NT, int BS, int RP>
int launch(hipStream_t st, PackArgs<NT> a, int K)
{
    const bool ghost = a.gh != nullptr;
    a.W = a.P + (ghost ? 1 : 0);
    a.R = BS / a.W;
    const bool multi = a.xk[0] != nullptr;
    if (multi && g_pack_multi_r > 0 && a.R > g_pack_multi_r) a.R = g_pack_multi_r;
    if (a.R * K > 4 * BS) a.R = 4 * BS / K;  // at most 4 prefetched words per thread
    if (a.vals && a.R * K * RP * NT > 2 * BS) a.R = 2 * BS / (K * RP * NT);  // ... and 2 values
    STK_REQUIRE(a.R >= 1, "stk_kron_pack_apply: a slot row of %d x %d x %d values is too wide", K, RP, NT);
    const int KS = (K + 3) & ~3;
    auto lds_of = [&](int R) {
        const size_t values = a.vals ? (size_t)R * K * RP * NT : (size_t)a.n_codes * RP * NT;
        return sizeof(double) * ((a.any_tri ? (size_t)NT * R * RP * (a.n_loc + 3) : 0) + values +
                                 (size_t)NT * 3 * (a.n_loc + 2)) +
               sizeof(uint32_t) * ((size_t)R * KS + (size_t)R * RP + 4) + 32;
    };
    while (a.R > 1 && lds_of(a.R) > 64 * 1024) --a.R;  // short slabs: many units per group
    a.ngroups = (a.n_units + a.R - 1) / a.R;
    a.chunk = (a.ngroups + 7) / 8;
    a.flags = g_pack_flags;
    a.diag = g_pack_diag;
    const size_t lds = lds_of(a.R);
    STK_REQUIRE(lds <= 64 * 1024, "stk_kron_pack_apply: %zu bytes of LDS per workgroup (dictionary too large?)", lds);
    const int n_cu = stk_cu_count();
    int per_cu = g_pack_wg_per_cu > 0 ? g_pack_wg_per_cu : ((K >= 12 || RP > 1) ? 2 : 3) * (512 / BS);
    if (multi && g_pack_multi_wg_per_cu > 0) per_cu = g_pack_multi_wg_per_cu;
    const int by_lds = (int)(160 * 1024 / (lds + 256));
    if (per_cu > by_lds) per_cu = by_lds > 0 ? by_lds : 1;
    int per_xcd = (n_cu / 8) * per_cu;
    if (per_xcd > a.chunk) per_xcd = a.chunk;
    if (per_xcd < 1) per_xcd = 1;
    const unsigned grid = (unsigned)per_xcd * 8;
#define STK_PACK_CASE(KK)                                                   \
    case KK:                                                                \
        return ghost ? launch_npf<NT, KK, true, BS, RP>(st, a, grid, lds)   \
                     : launch_npf<NT, KK, false, BS, RP>(st, a, grid, lds);
    if constexpr (RP == 1) {
        switch (K) {
            STK_PACK_CASE(5)
            STK_PACK_CASE(7)
            STK_PACK_CASE(9)
            STK_PACK_CASE(12)
            STK_PACK_CASE(16)
        }
        stk_set_error("stk_kron_pack_apply: K=%d is not one of 5, 7, 9, 12, 16", K);
    } else if constexpr (RP == 2) {
        switch (K) {
            STK_PACK_CASE(8)
            STK_PACK_CASE(10)
            STK_PACK_CASE(12)
        }
        stk_set_error("stk_kron_pack_apply: K=%d is not one of 8, 10, 12 (row pairs)", K);
    }
#undef STK_PACK_CASE
    return 2;
}

template <int NT>
int dispatch(hipStream_t st, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld, const stk_kron_pack_term *t,
             const double *x, const double *gh, double beta, double *y, const double *const *xs = nullptr)
{
    PackArgs<NT> a;
    for (int k = 0; k < NT; ++k) a.xk[k] = xs ? xs[k] : nullptr;
    a.slots = pat->slots;
    a.row_ids = pat->row_ids;
    a.x = x;
    a.gh = gh;
    a.y = y;
    a.beta = beta;
    a.M = pat->M;
    a.n_units = pat->n_units;
    a.n_loc = n_loc;
    a.ld = ld;
    a.col_bits = pat->col_bits;
    a.n_codes = pat->n_codes;
    a.any_tri = 0;
    a.vals = pat->vals;
    a.n_mats = pat->n_mats;
    for (int k = 0; k < NT; ++k) {
        a.dict[k] = pat->vals ? nullptr : pat->dict + (size_t)t[k].mat * pat->n_codes * pat->rows_per_unit;
        a.tri[k] = t[k].tri;
        a.mat[k] = t[k].mat;
        if (t[k].tri) a.any_tri = 1;
    }
    a.P = (n_loc + 1) / 2;
    if (pat->rows_per_unit == 2) return launch<NT, 512, 2>(st, a, pat->K);
    // 256-thread workgroups: only where a row still fits comfortably
    if (g_pack_block == 256 && a.P + 1 <= 64 && !xs) return launch<NT, 256, 1>(st, a, pat->K);
    return launch<NT, 512, 1>(st, a, pat->K);
}

}  // namespace

int stk_kron_pack_set_tuning(const char *key, int32_t value)
{
    if (std::strcmp(key, "pack_wg_per_cu") == 0) {
        g_pack_wg_per_cu = value;
        return 0;
    }
    if (std::strcmp(key, "pack_flags") == 0) {
        g_pack_flags = value;
        return 0;
    }
    if (std::strcmp(key, "pack_multi_wg_per_cu") == 0) {
        g_pack_multi_wg_per_cu = value;
        return 0;
    }
    if (std::strcmp(key, "pack_multi_r") == 0) {
        g_pack_multi_r = value;
        return 0;
    }
    if (std::strcmp(key, "pack_multi_lanes") == 0) {
        g_pack_multi_lanes = value;
        return 0;
    }
    if (std::strcmp(key, "pack_check_steps") == 0) {
        g_pack_check_steps = value;
        return 0;
    }
    if (std::strcmp(key, "pack_block") == 0) {
        g_pack_block = value == 256 ? 256 : 512;
        return 0;
    }
    return 1;
}

extern "C" int stk_kron_pack_apply(void *stream, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld,
                                   int32_t n_terms, const stk_kron_pack_term *t, const double *x,
                                   const double *ghosts, double beta, double *y)
{
    const stk_timed timed_(STK_OP_KRON, stream);
    STK_REQUIRE(pat && t && x && y, "stk_kron_pack_apply: null pointer");
    STK_REQUIRE(pat->M > 0 && pat->K >= 1 && pat->slots && (pat->dict || pat->vals), "stk_kron_pack_apply: bad pattern");
    STK_REQUIRE(!pat->vals || pat->rows_per_unit == 2, "stk_kron_pack_apply: explicit values need row pairs");
    if (pat->vals) {
        STK_REQUIRE(pat->n_mats == n_terms, "stk_kron_pack_apply: explicit values list %d matrices for %d terms",
                    pat->n_mats, n_terms);
        for (int k = 0; k < n_terms; ++k)
            STK_REQUIRE(t[k].mat == k, "stk_kron_pack_apply: with explicit values term %d must name matrix %d", k, k);
    }
    STK_REQUIRE(pat->rows_per_unit == 1 || pat->rows_per_unit == 2,
                "stk_kron_pack_apply: rows_per_unit=%d is not 1 or 2", pat->rows_per_unit);
    STK_REQUIRE(pat->n_units > 0 && (int64_t)pat->n_units * pat->rows_per_unit >= pat->M &&
                    (pat->rows_per_unit == 1 ? pat->n_units == pat->M : pat->row_ids != nullptr),
                "stk_kron_pack_apply: %d slot rows of %d matrix rows each do not cover M=%d (row_ids are required for pairs)",
                pat->n_units, pat->rows_per_unit, pat->M);
    STK_REQUIRE(pat->col_bits >= 1 && pat->col_bits <= 31 && ((int64_t)1 << pat->col_bits) >= pat->M,
                "stk_kron_pack_apply: col_bits=%d cannot address %d columns", pat->col_bits, pat->M);
    STK_REQUIRE(pat->vals || (pat->n_codes >= 1 && (int64_t)pat->n_codes <= ((int64_t)1 << (32 - pat->col_bits))),
                "stk_kron_pack_apply: %d codes do not fit %d bits", pat->n_codes, 32 - pat->col_bits);
    STK_REQUIRE(n_loc > 0 && ld >= n_loc && (ld & 1) == 0,
                "stk_kron_pack_apply: bad sizes n_loc=%d ld=%d (ld must be even)", n_loc, ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_pack_apply: n_terms=%d not in 1..3", n_terms);
    STK_REQUIRE((n_loc + 1) / 2 + 2 <= 512, "stk_kron_pack_apply: n_loc=%d too large", n_loc);
    STK_REQUIRE(x != y, "stk_kron_pack_apply: input aliases output");
    STK_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)ghosts) & 15) == 0,
                "stk_kron_pack_apply: x, y and ghosts must be 16-byte aligned");
    for (int k = 0; k < n_terms; ++k)
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < pat->n_mats, "stk_kron_pack_apply: term %d names matrix %d of %d", k,
                    t[k].mat, pat->n_mats);
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return dispatch<1>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
        case 2: return dispatch<2>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
        default: return dispatch<3>(st, pat, n_loc, ld, t, x, ghosts, beta, y);
    }
}

// kron_pack_multi.hip: a lane group per term (-1: a slot row would need more than 512 lanes)
int stk_kron_pack_terms_launch(hipStream_t st, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld, int32_t n_terms,
                               const stk_kron_pack_term *t, const double *const *xs, const int32_t *t_begin,
                               const int32_t *t_end, double beta, double *y);

extern "C" int stk_kron_pack_apply_multi_steps(void *stream, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld,
                                               int32_t n_terms, const stk_kron_pack_term *t,
                                               const double *const *xs_host, const int32_t *t_begin_host,
                                               const int32_t *t_end_host, double beta, double *y)
{
    const stk_timed timed_(STK_OP_KRON, stream);
    STK_REQUIRE(pat && t && xs_host && y, "stk_kron_pack_apply_multi: null pointer");
    STK_REQUIRE(pat->M > 0 && pat->K >= 1 && pat->slots && pat->dict && !pat->vals,
                "stk_kron_pack_apply_multi: needs a pattern with a dictionary");
    STK_REQUIRE(pat->rows_per_unit == 1 || pat->rows_per_unit == 2,
                "stk_kron_pack_apply_multi: rows_per_unit=%d is not 1 or 2", pat->rows_per_unit);
    STK_REQUIRE(pat->n_units > 0 && (int64_t)pat->n_units * pat->rows_per_unit >= pat->M &&
                    (pat->rows_per_unit == 1 ? pat->n_units == pat->M : pat->row_ids != nullptr),
                "stk_kron_pack_apply_multi: %d slot rows of %d matrix rows each do not cover M=%d", pat->n_units,
                pat->rows_per_unit, pat->M);
    STK_REQUIRE(pat->col_bits >= 1 && pat->col_bits <= 31 && ((int64_t)1 << pat->col_bits) >= pat->M,
                "stk_kron_pack_apply_multi: col_bits=%d cannot address %d columns", pat->col_bits, pat->M);
    STK_REQUIRE(pat->n_codes >= 1 && (int64_t)pat->n_codes <= ((int64_t)1 << (32 - pat->col_bits)),
                "stk_kron_pack_apply_multi: %d codes do not fit %d bits", pat->n_codes, 32 - pat->col_bits);
    STK_REQUIRE(n_loc > 0 && ld >= n_loc && (ld & 1) == 0,
                "stk_kron_pack_apply_multi: bad sizes n_loc=%d ld=%d (ld must be even)", n_loc, ld);
    STK_REQUIRE(n_terms >= 2 && n_terms <= 3, "stk_kron_pack_apply_multi: n_terms=%d not 2 or 3", n_terms);
    STK_REQUIRE((n_loc + 1) / 2 + 2 <= 512, "stk_kron_pack_apply_multi: n_loc=%d too large", n_loc);
    STK_REQUIRE((t_begin_host == nullptr) == (t_end_host == nullptr),
                "stk_kron_pack_apply_multi_steps: t_begin and t_end come together");
    for (int k = 0; k < n_terms; ++k) {
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < pat->n_mats, "stk_kron_pack_apply_multi: term %d names matrix %d of %d",
                    k, t[k].mat, pat->n_mats);
        STK_REQUIRE(xs_host[k] && xs_host[k] != y && ((uintptr_t)xs_host[k] & 15) == 0,
                    "stk_kron_pack_apply_multi: input %d missing, aliasing the output or not 16-byte aligned", k);
        STK_REQUIRE(!t_begin_host || (t_begin_host[k] >= 0 && t_begin_host[k] <= t_end_host[k] && t_end_host[k] <= n_loc),
                    "stk_kron_pack_apply_multi_steps: term %d states the time steps [%d, %d) of %d", k,
                    t_begin_host ? t_begin_host[k] : 0, t_end_host ? t_end_host[k] : 0, n_loc);
    }
    STK_REQUIRE(((uintptr_t)y & 15) == 0, "stk_kron_pack_apply_multi: y must be 16-byte aligned");
    hipStream_t st = stk_stream(stream);
    if (g_pack_check_steps && t_begin_host) {
        // Tuning key "pack_check_steps" (tests, debugging): a stated range that omits a
        // time step the factor reads would silently drop its contribution (the lanes of
        // that step are never launched).  Column s of a factor is read through sub[s + 1],
        // dia[s] and super[s - 1]: fetch the three diagonals and look (a stream
        // synchronisation per call: not for production runs).
        std::vector<double> tri(3 * (size_t)n_loc);
        for (int k = 0; k < n_terms; ++k) {
            if (!t[k].tri) {
                STK_REQUIRE(t_begin_host[k] == 0 && t_end_host[k] == n_loc,
                            "stk_kron_pack_apply_multi_steps: term %d has an identity time factor but states the "
                            "steps [%d, %d) of %d", k, t_begin_host[k], t_end_host[k], n_loc);
                continue;
            }
            STK_HIP(hipMemcpyAsync(tri.data(), t[k].tri, sizeof(double) * tri.size(), hipMemcpyDeviceToHost, st));
            STK_HIP(hipStreamSynchronize(st));
            for (int s_ = 0; s_ < n_loc; ++s_) {
                const bool read = tri[n_loc + s_] != 0.0 || (s_ + 1 < n_loc && tri[s_ + 1] != 0.0) ||
                                  (s_ >= 1 && tri[2 * (size_t)n_loc + s_ - 1] != 0.0);
                STK_REQUIRE(!read || (s_ >= t_begin_host[k] && s_ < t_end_host[k]),
                            "stk_kron_pack_apply_multi_steps: term %d reads time step %d, outside the stated [%d, %d)",
                            k, s_, t_begin_host[k], t_end_host[k]);
            }
        }
    }
    // A lane group per term pays where the slab is long or does not fit the Infinity
    // Cache; on short slabs of small problems the turn-taking lanes are level or ahead in
    // a loop of launches and level inside S (profiles/r05_multi_lanes_vs_turns.log).
    const bool lanes = g_pack_multi_lanes == 2 ||
                       (g_pack_multi_lanes == 1 && (n_loc >= 24 || (int64_t)pat->M * ld * 8 > ((int64_t)256 << 20)));
    if (lanes) {
        const int rc = stk_kron_pack_terms_launch(st, pat, n_loc, ld, n_terms, t, xs_host, t_begin_host, t_end_host,
                                                  beta, y);
        if (rc >= 0) return rc;
    }
    if (n_terms == 2) return dispatch<2>(st, pat, n_loc, ld, t, xs_host[0], nullptr, beta, y, xs_host);
    return dispatch<3>(st, pat, n_loc, ld, t, xs_host[0], nullptr, beta, y, xs_host);
}

extern "C" int stk_kron_pack_apply_multi(void *stream, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld,
                                         int32_t n_terms, const stk_kron_pack_term *t, const double *const *xs_host,
                                         double beta, double *y)
{
    return stk_kron_pack_apply_multi_steps(stream, pat, n_loc, ld, n_terms, t, xs_host, nullptr, nullptr, beta, y);
}

extern "C" int stk_interleave_ghosts(void *stream, int32_t M, const double *lo, const double *hi, double *ghosts)
{
    STK_REQUIRE(M > 0 && ghosts, "stk_interleave_ghosts: bad arguments");
    STK_REQUIRE(((uintptr_t)ghosts & 15) == 0, "stk_interleave_ghosts: ghosts must be 16-byte aligned");
    hipLaunchKernelGGL(interleave_ghosts_kernel, dim3(stk_flat_grid(M, 256)), dim3(256), 0, stk_stream(stream), M, lo,
                       hi, reinterpret_cast<double2 *>(ghosts));
    STK_LAUNCH_CHECK();
    return 0;
}

/* Diagnostic (tools/kron_ab.py --phases): while `buf` is set, launches of the headline
 * instantiation (2 terms, K = 7, no ghosts) run the stamped build and leave
 * per-wave cycle sums of the four segments of an iteration in buf[wave][4]. */
extern "C" int stk_kron_pack_set_diag(unsigned long long *buf)
{
    g_pack_diag = buf;
    return 0;
}

extern "C" int stk_kron_pack_ghost_apply(void *stream, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld,
                                         int32_t n_terms, const stk_kron_pack_term *t, const double *x,
                                         const double *x_lo, const double *x_hi, double *y)
{
    const stk_timed timed_(STK_OP_KRON, stream);
    STK_REQUIRE(pat && t && x && y && x != y, "stk_kron_pack_ghost_apply: null pointer or input aliases output");
    if (!x_lo && !x_hi) return 0;
    STK_REQUIRE(pat->M > 0 && pat->K >= 1 && pat->slots && (pat->dict || pat->vals) && pat->n_units > 0,
                "stk_kron_pack_ghost_apply: bad pattern");
    STK_REQUIRE(pat->rows_per_unit == 1 || (pat->rows_per_unit == 2 && pat->row_ids),
                "stk_kron_pack_ghost_apply: rows_per_unit=%d", pat->rows_per_unit);
    STK_REQUIRE(n_loc > 0 && ld >= n_loc, "stk_kron_pack_ghost_apply: bad sizes n_loc=%d ld=%d", n_loc, ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_pack_ghost_apply: n_terms=%d not in 1..3", n_terms);
    STK_REQUIRE(pat->vals || sizeof(double) * (size_t)pat->n_codes * pat->rows_per_unit * n_terms <= 60 * 1024,
                "stk_kron_pack_ghost_apply: dictionary too large");
    if (pat->vals) {
        STK_REQUIRE(pat->n_mats == n_terms, "stk_kron_pack_ghost_apply: explicit values list %d matrices for %d terms",
                    pat->n_mats, n_terms);
        for (int k = 0; k < n_terms; ++k)
            STK_REQUIRE(t[k].mat == k, "stk_kron_pack_ghost_apply: with explicit values term %d must name matrix %d", k,
                        k);
    }
    for (int k = 0; k < n_terms; ++k)
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < pat->n_mats, "stk_kron_pack_ghost_apply: term %d names matrix %d of %d",
                    k, t[k].mat, pat->n_mats);
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return dispatch_ghost_only<1>(st, pat, n_loc, ld, t, x, x_lo, x_hi, y);
        case 2: return dispatch_ghost_only<2>(st, pat, n_loc, ld, t, x, x_lo, x_hi, y);
        default: return dispatch_ghost_only<3>(st, pat, n_loc, ld, t, x, x_lo, x_hi, y);
    }
}

extern "C" int stk_kron_pack_boundary_apply(void *stream, const stk_pack_pattern *pat, int32_t n_loc, int32_t ld,
                                            int32_t n_terms, const stk_kron_pack_term *t, const double *records,
                                            const double *ghosts, int32_t has_lo, int32_t has_hi, double *y)
{
    const stk_timed timed_(STK_OP_KRON, stream);
    STK_REQUIRE(pat && t && records && ghosts && y, "stk_kron_pack_boundary_apply: null pointer");
    if (!has_lo && !has_hi) return 0;
    STK_REQUIRE(pat->M > 0 && pat->K >= 1 && pat->slots && (pat->dict || pat->vals) && pat->n_units > 0,
                "stk_kron_pack_boundary_apply: bad pattern");
    STK_REQUIRE(pat->rows_per_unit == 1 || (pat->rows_per_unit == 2 && pat->row_ids),
                "stk_kron_pack_boundary_apply: rows_per_unit=%d", pat->rows_per_unit);
    STK_REQUIRE(n_loc > 0 && ld >= n_loc, "stk_kron_pack_boundary_apply: bad sizes n_loc=%d ld=%d", n_loc, ld);
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_pack_boundary_apply: n_terms=%d not in 1..3", n_terms);
    STK_REQUIRE((((uintptr_t)records & 31) | ((uintptr_t)ghosts & 15)) == 0,
                "stk_kron_pack_boundary_apply: records must be 32-byte, ghosts 16-byte aligned");
    STK_REQUIRE(pat->vals || sizeof(double) * (size_t)pat->n_codes * pat->rows_per_unit * n_terms <= 60 * 1024,
                "stk_kron_pack_boundary_apply: dictionary too large");
    if (pat->vals) {
        STK_REQUIRE(pat->n_mats == n_terms, "stk_kron_pack_boundary_apply: explicit values list %d matrices for %d terms",
                    pat->n_mats, n_terms);
        for (int k = 0; k < n_terms; ++k)
            STK_REQUIRE(t[k].mat == k, "stk_kron_pack_boundary_apply: with explicit values term %d must name matrix %d",
                        k, k);
    }
    for (int k = 0; k < n_terms; ++k)
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < pat->n_mats,
                    "stk_kron_pack_boundary_apply: term %d names matrix %d of %d", k, t[k].mat, pat->n_mats);
    hipStream_t st = stk_stream(stream);
    switch (n_terms) {
        case 1: return dispatch_boundary<1>(st, pat, n_loc, ld, t, records, ghosts, has_lo, has_hi, y);
        case 2: return dispatch_boundary<2>(st, pat, n_loc, ld, t, records, ghosts, has_lo, has_hi, y);
        default: return dispatch_boundary<3>(st, pat, n_loc, ld, t, records, ghosts, has_lo, has_hi, y);
    }
}
