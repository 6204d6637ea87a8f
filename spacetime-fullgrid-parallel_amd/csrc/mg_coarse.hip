// The coarse end of a V-cycle in one launch.
//
// On the coarse levels of the hierarchy a Gauss-Seidel group or a transfer is
// a few hundred rows: each of the ~27 launches per level costs more in launch
// latency than in work.  This kernel runs the whole sub-V-cycle below a cut
// level Lc -- MGM(Lc, u, f) of reference source/multigrid.py:168-182, i.e.
// smooth / residual / restrict / ... / exact coarse solve / ... / correct /
// smooth on levels Lc..0 -- as a list of jobs executed in order.  Time slices
// never interact, so every workgroup owns a few time pairs of ALL rows and only
// workgroup barriers separate the jobs: no inter-workgroup synchronisation.
//
// The arithmetic of a job is that of rows_ell.hip (same sliced-ELL matrices,
// same summation order), so results are identical to the level-by-level path.
//
// Round 4, the UNIFORM form (mg_coarse_uniform_kernel): a job costs ~1.5 us whatever
// its rows (220 us for the 141 jobs of the preconditioner family's sub-V-cycle, 124 us
// for K's 81) -- two dependent trips to the L2 per job, the descriptor and then the
// matrix entries of the rows.  All matrices of the job list are therefore copied once,
// at plan construction, to ONE slot count KU (8 or 12; zero-valued slots appended: an
// added slot contributes fma(0, x, s) = s, results unchanged bit for bit), the job
// descriptors travel to LDS when the kernel starts, and every thread fetches the
// entries of its first row of job n + 1 BEFORE the barrier that ends job n.
#include <algorithm>
#include <map>
#include <tuple>
#include <vector>

#include "stk_common.h"

namespace {

constexpr int CBS = 512;
enum { JOB_SPMM = 0, JOB_GS = 1, JOB_ZERO = 2, JOB_COARSE = 3 };

struct Job {
    int32_t kind, K;
    int32_t pos_begin, pos_end;   // ELL rows (JOB_ZERO / JOB_COARSE: row count in pos_end)
    const int32_t *idx;
    const double *va, *vm;
    const int32_t *row_ids;
    const double *dia_a, *dia_m;
    const double *x;   // gather source (GS: u)
    const double *z;   // SPMM: beta operand, GS: right-hand side
    double *y;
    double alpha, beta;
    int32_t use_m;     // entries are ca*va + cm[t]*vm (level matrices) or plain va (transfers)
    int32_t diag_free; // GS: slots hold the off-diagonal entries only, u_i = (f_i - s) / a_ii
    int32_t lx, lz, ly;  // the same three vectors as row offsets into the LDS arena (LDS variant)
    int32_t level;       // level of the hierarchy the job's matrix belongs to
};

struct CoarseArgs {
    const Job *jobs;
    int32_t n_jobs;
    // LDS variant: arena of lds_rows double2 rows; f of the cut level comes from /
    // u of the cut level goes to global memory, everything else stays in LDS
    int32_t lds_rows, top_n, top_lu, top_lf;
    const double *top_f;
    double *top_u;
    int32_t n_loc, ld;
    int32_t pairs_per_wg;
    double ca;
    const double *cm;          // [n_loc] or NULL
    const int32_t *kind;       // coarse-inverse index per time slice or NULL
    const double *coarse_inv;  // [n_kinds][n0][n0]
    double coarse_scale;       // 1/ca when cm == NULL
};

template <int K, bool HAS_M>
__device__ inline void run_rows_job(const Job &j, const CoarseArgs &a, int p0, int W)
{
    const int nrows = j.pos_end - j.pos_begin;
    const bool use_m = HAS_M && j.use_m;
    const double ca = j.use_m ? a.ca : 1.0;
    for (int item = threadIdx.x; item < nrows * W; item += CBS) {
        const int r = item / W, p = p0 + (item - r * W);
        const int t0 = 2 * p;
        const bool has1 = t0 + 1 < a.n_loc;
        const int pos = j.pos_begin + r;
        const size_t e0 = (size_t)pos * K;
        const int row = j.row_ids ? j.row_ids[pos] : pos;
        double cm0 = 0.0, cm1 = 0.0;
        if (use_m) {
            cm0 = a.cm[t0];
            if (has1) cm1 = a.cm[t0 + 1];
        }
        int col[K];
        double2 xv[K];
#pragma unroll
        for (int u = 0; u < K; ++u) col[u] = j.idx[e0 + u];
#pragma unroll
        for (int u = 0; u < K; ++u)
            xv[u] = *reinterpret_cast<const double2 *>(j.x + (size_t)col[u] * a.ld + t0);
        const size_t yo = (size_t)row * a.ld + t0;
        double2 zv = make_double2(0.0, 0.0), own = make_double2(0.0, 0.0);
        if (j.kind == JOB_GS) {
            zv = *reinterpret_cast<const double2 *>(j.z + yo);
            if (!j.diag_free) own = *reinterpret_cast<const double2 *>(j.x + yo);
        } else if (j.beta != 0.0) {
            zv = *reinterpret_cast<const double2 *>(j.z + yo);
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int u = 0; u < K; ++u) {
            double v0 = ca * j.va[e0 + u], v1 = v0;
            if (use_m) {
                const double m = j.vm[e0 + u];
                v0 = fma(cm0, m, v0);
                v1 = fma(cm1, m, v1);
            }
            s0 = fma(v0, xv[u].x, s0);
            s1 = fma(v1, xv[u].y, s1);
        }
        double o0, o1;
        if (j.kind == JOB_GS) {
            double d0 = ca * j.dia_a[pos], d1 = d0;
            if (use_m) {
                const double m = j.dia_m[pos];
                d0 = fma(cm0, m, d0);
                d1 = fma(cm1, m, d1);
            }
            o0 = own.x + (1.0 / d0) * (zv.x - s0);
            o1 = own.y + (1.0 / d1) * (zv.y - s1);
        } else {
            o0 = j.alpha * s0;
            o1 = j.alpha * s1;
            if (j.beta != 0.0) {
                o0 = fma(j.beta, zv.x, o0);
                o1 = fma(j.beta, zv.y, o1);
            }
        }
        if (!has1) o1 = 0.0;  // padding slot stays zero
        *reinterpret_cast<double2 *>(j.y + yo) = make_double2(o0, o1);
    }
}

// LDS variant of run_rows_job: every vector of the sub-V-cycle lives in the LDS
// arena sv, one element per row.  PAIR: a workgroup owns a pair of time steps
// (double2 elements); otherwise a single time step (double elements), which
// halves the arena and lets one more level fit.
template <bool PAIR>
struct TimeElem;
template <>
struct TimeElem<true> {
    typedef double2 type;
    __device__ static inline double2 make(double a, double b) { return make_double2(a, b); }
    __device__ static inline double lo(double2 v) { return v.x; }
    __device__ static inline double hi(double2 v) { return v.y; }
};
template <>
struct TimeElem<false> {
    typedef double type;
    __device__ static inline double make(double a, double) { return a; }
    __device__ static inline double lo(double v) { return v; }
    __device__ static inline double hi(double) { return 0.0; }
};

template <int K, bool HAS_M, bool PAIR>
__device__ inline void run_rows_job_lds(const Job &j, const CoarseArgs &a, typename TimeElem<PAIR>::type *sv,
                                        double cm0, double cm1, bool has1)
{
    typedef TimeElem<PAIR> E;
    typedef typename E::type V;
    const int nrows = j.pos_end - j.pos_begin;
    const bool use_m = HAS_M && j.use_m;
    const double ca = j.use_m ? a.ca : 1.0;
    const V *vx = sv + j.lx;
    for (int r = threadIdx.x; r < nrows; r += CBS) {
        const int pos = j.pos_begin + r;
        const size_t e0 = (size_t)pos * K;
        const int row = j.row_ids ? j.row_ids[pos] : pos;
        int col[K];
        double va[K], vm[K];
#pragma unroll
        for (int u = 0; u < K; ++u) col[u] = j.idx[e0 + u];
#pragma unroll
        for (int u = 0; u < K; ++u) va[u] = j.va[e0 + u];
        if (use_m) {
#pragma unroll
            for (int u = 0; u < K; ++u) vm[u] = j.vm[e0 + u];
        }
        double da = 0.0, dm = 0.0;
        if (j.kind == JOB_GS) {
            da = j.dia_a[pos];
            if (use_m) dm = j.dia_m[pos];
        }
        V xv[K];
#pragma unroll
        for (int u = 0; u < K; ++u) xv[u] = vx[col[u]];
        V zv = E::make(0.0, 0.0), own = E::make(0.0, 0.0);
        if (j.kind == JOB_GS) {
            zv = sv[j.lz + row];
            if (!j.diag_free) own = vx[row];
        } else if (j.beta != 0.0) {
            zv = sv[j.lz + row];
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int u = 0; u < K; ++u) {
            double v0 = ca * va[u], v1 = v0;
            if (use_m) {
                v0 = fma(cm0, vm[u], v0);
                if (PAIR) v1 = fma(cm1, vm[u], v1);
            }
            s0 = fma(v0, E::lo(xv[u]), s0);
            if (PAIR) s1 = fma(v1, E::hi(xv[u]), s1);
        }
        double o0, o1 = 0.0;
        if (j.kind == JOB_GS) {
            double d0 = ca * da, d1 = d0;
            if (use_m) {
                d0 = fma(cm0, dm, d0);
                if (PAIR) d1 = fma(cm1, dm, d1);
            }
            o0 = E::lo(own) + (1.0 / d0) * (E::lo(zv) - s0);
            if (PAIR) o1 = E::hi(own) + (1.0 / d1) * (E::hi(zv) - s1);
        } else {
            o0 = j.alpha * s0;
            if (PAIR) o1 = j.alpha * s1;
            if (j.beta != 0.0) {
                o0 = fma(j.beta, E::lo(zv), o0);
                if (PAIR) o1 = fma(j.beta, E::hi(zv), o1);
            }
        }
        if (!has1) o1 = 0.0;  // padding slot stays zero
        sv[j.ly + row] = E::make(o0, o1);
    }
}

template <bool HAS_M, bool PAIR>
__global__ __launch_bounds__(CBS) void mg_coarse_lds_kernel(const CoarseArgs a)
{
    typedef TimeElem<PAIR> E;
    typedef typename E::type V;
    extern __shared__ double sm_raw[];
    V *sv = reinterpret_cast<V *>(sm_raw);
    const int t0 = (PAIR ? 2 : 1) * (int)blockIdx.x;
    const bool has1 = PAIR && t0 + 1 < a.n_loc;
    double cm0 = 0.0, cm1 = 0.0;
    if (HAS_M) {
        cm0 = a.cm[t0];
        if (has1) cm1 = a.cm[t0 + 1];
    }
    auto gload = [&](const double *p) -> V {
        if constexpr (PAIR)
            return *reinterpret_cast<const double2 *>(p);
        else
            return *p;
    };
    for (int r = threadIdx.x; r < a.top_n; r += CBS) {
        sv[a.top_lf + r] = gload(a.top_f + (size_t)r * a.ld + t0);
        sv[a.top_lu + r] = E::make(0.0, 0.0);  // MGM starts from zero (multigrid.py:176)
    }
    __syncthreads();
    for (int jn = 0; jn < a.n_jobs; ++jn) {
        const Job j = a.jobs[jn];
        if (j.kind == JOB_ZERO) {
            for (int r = threadIdx.x; r < j.pos_end; r += CBS) sv[j.ly + r] = E::make(0.0, 0.0);
        } else if (j.kind == JOB_COARSE) {
            const int n0 = j.pos_end;
            for (int i = threadIdx.x; i < n0; i += CBS) {
                const double *A0 = a.coarse_inv + (size_t)(a.kind ? a.kind[t0] : 0) * n0 * n0 + (size_t)i * n0;
                const double *A1 =
                    a.coarse_inv + (size_t)((a.kind && has1) ? a.kind[t0 + 1] : 0) * n0 * n0 + (size_t)i * n0;
                double s0 = 0.0, s1 = 0.0;
                for (int c = 0; c < n0; ++c) {
                    const V fv = sv[j.lz + c];
                    s0 = fma(A0[c], E::lo(fv), s0);
                    if (PAIR) s1 = fma(A1[c], E::hi(fv), s1);
                }
                sv[j.ly + i] = E::make(a.coarse_scale * s0, has1 ? a.coarse_scale * s1 : 0.0);
            }
        } else {
            switch (j.K) {
                case 2: run_rows_job_lds<2, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
                case 4: run_rows_job_lds<4, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
                case 5: run_rows_job_lds<5, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
                case 6: run_rows_job_lds<6, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
                case 7: run_rows_job_lds<7, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
                case 9: run_rows_job_lds<9, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
                case 12: run_rows_job_lds<12, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
                case 16: run_rows_job_lds<16, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
                default: run_rows_job_lds<20, HAS_M, PAIR>(j, a, sv, cm0, cm1, has1); break;
            }
        }
        __syncthreads();
    }
    for (int r = threadIdx.x; r < a.top_n; r += CBS) {
        double *dst = a.top_u + (size_t)r * a.ld + t0;
        if constexpr (PAIR)
            *reinterpret_cast<double2 *>(dst) = sv[a.top_lu + r];
        else
            *dst = sv[a.top_lu + r];
    }
    // single steps: the padding column of an odd slab stays zero
    if (!PAIR && blockIdx.x == 0 && (a.n_loc & 1))
        for (int r = threadIdx.x; r < a.top_n; r += CBS) a.top_u[(size_t)r * a.ld + a.n_loc] = 0.0;
}

template <bool HAS_M>
__global__ __launch_bounds__(CBS) void mg_coarse_kernel(const CoarseArgs a)
{
    const int all_pairs = (a.n_loc + 1) / 2;
    const int p0 = blockIdx.x * a.pairs_per_wg;
    const int W = min(a.pairs_per_wg, all_pairs - p0);
    if (W <= 0) return;
    for (int jn = 0; jn < a.n_jobs; ++jn) {
        const Job j = a.jobs[jn];
        if (j.kind == JOB_ZERO) {
            const int n = j.pos_end;
            for (int item = threadIdx.x; item < n * W; item += CBS) {
                const int r = item / W, p = p0 + (item - r * W);
                *reinterpret_cast<double2 *>(j.y + (size_t)r * a.ld + 2 * p) = make_double2(0.0, 0.0);
            }
        } else if (j.kind == JOB_COARSE) {
            // exact solve on level 0 with the dense inverse (multigrid.py:161-170)
            const int n0 = j.pos_end;
            for (int item = threadIdx.x; item < n0 * W; item += CBS) {
                const int i = item / W, p = p0 + (item - i * W);
                const int t0 = 2 * p;
                const bool has1 = t0 + 1 < a.n_loc;
                const double *A0 = a.coarse_inv + (size_t)(a.kind ? a.kind[t0] : 0) * n0 * n0 + (size_t)i * n0;
                const double *A1 =
                    a.coarse_inv + (size_t)((a.kind && has1) ? a.kind[t0 + 1] : 0) * n0 * n0 + (size_t)i * n0;
                double s0 = 0.0, s1 = 0.0;
                for (int c = 0; c < n0; ++c) {
                    const double2 fv = *reinterpret_cast<const double2 *>(j.z + (size_t)c * a.ld + t0);
                    s0 = fma(A0[c], fv.x, s0);
                    s1 = fma(A1[c], fv.y, s1);
                }
                *reinterpret_cast<double2 *>(j.y + (size_t)i * a.ld + t0) =
                    make_double2(a.coarse_scale * s0, has1 ? a.coarse_scale * s1 : 0.0);
            }
        } else {
            switch (j.K) {
                case 2: run_rows_job<2, HAS_M>(j, a, p0, W); break;
                case 4: run_rows_job<4, HAS_M>(j, a, p0, W); break;
                case 5: run_rows_job<5, HAS_M>(j, a, p0, W); break;
                case 6: run_rows_job<6, HAS_M>(j, a, p0, W); break;
                case 7: run_rows_job<7, HAS_M>(j, a, p0, W); break;
                case 9: run_rows_job<9, HAS_M>(j, a, p0, W); break;
                case 12: run_rows_job<12, HAS_M>(j, a, p0, W); break;
                case 16: run_rows_job<16, HAS_M>(j, a, p0, W); break;
                default: run_rows_job<20, HAS_M>(j, a, p0, W); break;
            }
        }
        // the next job reads what this one wrote (same workgroup, same time pairs)
        __syncthreads();
    }
}

// ---- the uniform form ---------------------------------------------------------
struct CJob {  // 48 bytes, copied to LDS
    int32_t kind, n_rows;
    int32_t lx, lz, ly;
    int32_t flags;  // bit 0: level matrix (ca*va + cm*vm), bit 1: diagonal-free Gauss-Seidel rows
    uint32_t mat_off;  // first row of the job in the uniform arrays
    int32_t pad;
    double alpha, beta;
};

struct UniArgs {
    const CJob *jobs;
    const int32_t *idx;    // [rows][KU]
    const double *va, *vm; // [rows][KU]
    const double *dia_a, *dia_m;  // [rows]
    const int32_t *row;    // [rows] output row of every listed row
    // MEMBER matrices (stk_mg_set_member_matrices): the level matrices of time slice t are
    // the entries of matrix kind[t] itself instead of ca*va + cm[t]*vm; NULL: none
    const double *vmem;    // [n_kinds][rows][KU]
    const double *dmem;    // [n_kinds][rows]
    size_t mem_rows;       // rows of the uniform arrays
};

__global__ void repack_uniform_kernel(int n_rows, int K_src, int KU, const int32_t *__restrict__ idx,
                                      const double *__restrict__ va, const double *__restrict__ vm,
                                      const int32_t *__restrict__ row_ids, const double *__restrict__ dia_a,
                                      const double *__restrict__ dia_m, int pos_begin, uint32_t mat_off, int32_t *o_idx,
                                      double *o_va, double *o_vm, double *o_da, double *o_dm, int32_t *o_row)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * KU) return;
    const int r = t / KU, u = t - r * KU;
    const size_t src = (size_t)(pos_begin + r) * K_src, dst = (size_t)(mat_off + r) * KU + u;
    // appended slots: the column of slot 0 (read anyway), value zero
    o_idx[dst] = idx[src + (u < K_src ? u : 0)];
    o_va[dst] = u < K_src ? va[src + u] : 0.0;
    if (o_vm) o_vm[dst] = (vm && u < K_src) ? vm[src + u] : 0.0;
    if (u == 0) {
        o_row[mat_off + r] = row_ids ? row_ids[pos_begin + r] : pos_begin + r;
        o_da[mat_off + r] = dia_a ? dia_a[pos_begin + r] : 0.0;
        if (o_dm) o_dm[mat_off + r] = dia_m ? dia_m[pos_begin + r] : 0.0;
    }
}

template <int KU, bool HAS_M, bool PAIR, int UBS, bool SF>
__global__ __launch_bounds__(UBS) void mg_coarse_uniform_kernel(const CoarseArgs a, const UniArgs m)
{
    typedef TimeElem<PAIR> E;
    typedef typename E::type V;
    extern __shared__ double sm_raw[];
    V *sv = reinterpret_cast<V *>(sm_raw);
    // the descriptors are copied with 16-byte accesses: their region starts on a 16-byte
    // boundary also where V = double and lds_rows is odd
    CJob *s_jobs = reinterpret_cast<CJob *>(sm_raw + (PAIR ? 2 * a.lds_rows : ((a.lds_rows + 1) & ~1)));
    const int tid = threadIdx.x;
    const int t0 = (PAIR ? 2 : 1) * (int)blockIdx.x;
    const bool has1 = PAIR && t0 + 1 < a.n_loc;
    double cm0 = 0.0, cm1 = 0.0;
    if (HAS_M) {
        cm0 = a.cm[t0];
        if (has1) cm1 = a.cm[t0 + 1];
    }
    // member matrices: this workgroup's time steps read the entries of their own matrix
    const bool members = HAS_M && m.vmem != nullptr && a.kind != nullptr;
    const size_t mem0 = members ? (size_t)a.kind[t0] * m.mem_rows : 0;
    const size_t mem1 = (members && has1) ? (size_t)a.kind[t0 + 1] * m.mem_rows : mem0;
    {
        const int4 *src = reinterpret_cast<const int4 *>(m.jobs);
        int4 *dst = reinterpret_cast<int4 *>(s_jobs);
        for (int i = tid; i < a.n_jobs * (int)(sizeof(CJob) / 16); i += UBS) dst[i] = src[i];
    }
    for (int r = tid; r < a.top_n; r += UBS) {
        const double *p = a.top_f + (size_t)r * a.ld + t0;
        if constexpr (PAIR)
            sv[a.top_lf + r] = *reinterpret_cast<const double2 *>(p);
        else
            sv[a.top_lf + r] = *p;
        sv[a.top_lu + r] = E::make(0.0, 0.0);  // MGM starts from zero (multigrid.py:176)
    }
    // Entries of one listed row, in registers.  TWO sets take turns: set A serves the
    // even jobs, set B the odd ones, and a set is refilled for job n + 2 as soon as job
    // n is through with it -- the entries then have a whole job and two barriers to
    // arrive (one job ahead they would be needed right behind the barrier: the first
    // thing a job does is gather at its columns).
    struct RowRegs {
        int32_t col[KU];
        double va[KU], vm[KU];
        double da, dm;
        int32_t orow;
    };
    auto load_row = [&](const CJob &j, int r, RowRegs &S) {
        const size_t row = (size_t)j.mat_off + r, e0 = row * KU;
        const int4 *pi = reinterpret_cast<const int4 *>(m.idx + e0);
#pragma unroll
        for (int q = 0; q < KU / 4; ++q) {
            const int4 v = pi[q];
            S.col[4 * q] = v.x, S.col[4 * q + 1] = v.y, S.col[4 * q + 2] = v.z, S.col[4 * q + 3] = v.w;
        }
        const bool mem = members && (j.flags & 1);
        // (member matrices: S.va / S.vm hold the entries for the first / second time step)
        const double2 *pa = reinterpret_cast<const double2 *>((mem ? m.vmem + mem0 * KU : m.va) + e0);
#pragma unroll
        for (int q = 0; q < KU / 2; ++q) {
            const double2 v = pa[q];
            S.va[2 * q] = v.x, S.va[2 * q + 1] = v.y;
        }
        // Every array is read whatever the job needs of it (the second value array and the
        // diagonals of a transfer are zeros in the uniform copies): the SAME loads for every
        // row, so that the compiler's wait counters know how many loads lie between a set of
        // registers and the next one.  With loads behind branches it waited for vmcnt(0)
        // before the first use of a prefetched row -- i.e. also for the set it had just issued
        // for the job after next -- and a job took one L2 round trip, 1.35 us, whatever its
        // rows (round 6).
        if (HAS_M && (SF || (j.flags & 1))) {
            const double2 *pm = reinterpret_cast<const double2 *>((mem ? m.vmem + mem1 * KU : m.vm) + e0);
#pragma unroll
            for (int q = 0; q < KU / 2; ++q) {
                const double2 v = pm[q];
                S.vm[2 * q] = v.x, S.vm[2 * q + 1] = v.y;
            }
        }
        if (SF || j.kind == JOB_GS) {
            S.da = (mem ? m.dmem + mem0 : m.dia_a)[row];
            if (HAS_M && (SF || (j.flags & 1))) S.dm = (mem ? m.dmem + mem1 : m.dia_m)[row];
        }
        S.orow = m.row[row];
    };
    // the first row of a thread in job jn -- or, where there is none (no row job, a thread
    // beyond the job's rows, past the last job), row 0 of the uniform arrays: same loads
    // (SF = false: loads only where there is a row -- round 4's form, kept for the A/B)
    auto fetch = [&](int jn, RowRegs &S) {
        if constexpr (SF) {
            CJob nj = s_jobs[min(jn, a.n_jobs - 1)];
            const bool rows = jn < a.n_jobs && (nj.kind == JOB_SPMM || nj.kind == JOB_GS) && tid < nj.n_rows;
            if (!rows) nj.mat_off = 0, nj.flags = 0;
            load_row(nj, rows ? tid : 0, S);
        } else if (jn < a.n_jobs) {
            const CJob nj = s_jobs[jn];
            if ((nj.kind == JOB_SPMM || nj.kind == JOB_GS) && tid < nj.n_rows) load_row(nj, tid, S);
        }
    };
    auto row_update = [&](const CJob &j, const RowRegs &S) {
        const bool use_m = HAS_M && (j.flags & 1);
        const bool mem = members && (j.flags & 1);
        const bool diag_free = (j.flags & 2) != 0;
        const double ca = (j.flags & 1) ? a.ca : 1.0;
        const V *vx = sv + j.lx;
        V xv[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) xv[u] = vx[S.col[u]];
        V zv = E::make(0.0, 0.0), own = E::make(0.0, 0.0);
        if (j.kind == JOB_GS) {
            zv = sv[j.lz + S.orow];
            if (!diag_free) own = vx[S.orow];
        } else if (j.beta != 0.0) {
            zv = sv[j.lz + S.orow];
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            double v0 = ca * S.va[u], v1 = v0;
            if (mem) {
                v0 = S.va[u], v1 = S.vm[u];
            } else if (use_m) {
                v0 = fma(cm0, S.vm[u], v0);
                if (PAIR) v1 = fma(cm1, S.vm[u], v1);
            }
            s0 = fma(v0, E::lo(xv[u]), s0);
            if (PAIR) s1 = fma(v1, E::hi(xv[u]), s1);
        }
        double o0, o1 = 0.0;
        if (j.kind == JOB_GS) {
            double d0 = ca * S.da, d1 = d0;
            if (mem) {
                d0 = S.da, d1 = S.dm;
            } else if (use_m) {
                d0 = fma(cm0, S.dm, d0);
                if (PAIR) d1 = fma(cm1, S.dm, d1);
            }
            o0 = E::lo(own) + (1.0 / d0) * (E::lo(zv) - s0);
            if (PAIR) o1 = E::hi(own) + (1.0 / d1) * (E::hi(zv) - s1);
        } else {
            o0 = j.alpha * s0;
            if (PAIR) o1 = j.alpha * s1;
            if (j.beta != 0.0) {
                o0 = fma(j.beta, E::lo(zv), o0);
                if (PAIR) o1 = fma(j.beta, E::hi(zv), o1);
            }
        }
        if (!has1) o1 = 0.0;  // padding slot stays zero
        sv[j.ly + S.orow] = E::make(o0, o1);
    };
    // One job; S holds this thread's first row of it and is refilled for job jn + 2.
    // The barrier is LDS-only: __syncthreads() would wait for the loads in flight (its
    // fence covers global memory, vmcnt(0)), which is why a prefetch in front of it
    // gains nothing; the jobs exchange data through LDS alone -- global memory is
    // read-only until the final store of the top level.
    auto run_job = [&](int jn, RowRegs &S) {
        const CJob j = s_jobs[jn];
        if (j.kind == JOB_ZERO) {
            for (int r = tid; r < j.n_rows; r += UBS) sv[j.ly + r] = E::make(0.0, 0.0);
        } else if (j.kind == JOB_COARSE) {
            const int n0 = j.n_rows;
            for (int i = tid; i < n0; i += UBS) {
                const double *A0 = a.coarse_inv + (size_t)(a.kind ? a.kind[t0] : 0) * n0 * n0 + (size_t)i * n0;
                const double *A1 =
                    a.coarse_inv + (size_t)((a.kind && has1) ? a.kind[t0 + 1] : 0) * n0 * n0 + (size_t)i * n0;
                double s0 = 0.0, s1 = 0.0;
                for (int c = 0; c < n0; ++c) {
                    const V fv = sv[j.lz + c];
                    s0 = fma(A0[c], E::lo(fv), s0);
                    if (PAIR) s1 = fma(A1[c], E::hi(fv), s1);
                }
                sv[j.ly + i] = E::make(a.coarse_scale * s0, has1 ? a.coarse_scale * s1 : 0.0);
            }
        } else {
            if (tid < j.n_rows) row_update(j, S);
            for (int r = tid + UBS; r < j.n_rows; r += UBS) {  // further rows: fetched on the spot
                RowRegs T;
                load_row(j, r, T);
                row_update(j, T);
            }
        }
        fetch(jn + 2, S);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    __syncthreads();
    RowRegs A, B;
    fetch(0, A);
    fetch(1, B);
    for (int jn = 0; jn < a.n_jobs; jn += 2) {
        run_job(jn, A);
        if (jn + 1 < a.n_jobs) run_job(jn + 1, B);
    }
    for (int r = tid; r < a.top_n; r += UBS) {
        double *dst = a.top_u + (size_t)r * a.ld + t0;
        if constexpr (PAIR)
            *reinterpret_cast<double2 *>(dst) = sv[a.top_lu + r];
        else
            *dst = sv[a.top_lu + r];
    }
    if (!PAIR && blockIdx.x == 0 && (a.n_loc & 1))
        for (int r = tid; r < a.top_n; r += UBS) a.top_u[(size_t)r * a.ld + a.n_loc] = 0.0;
}

}  // namespace

int g_mg_coarse_pairs = 0;  // time pairs per workgroup (0 = default: 1, vectors resident in LDS)
int g_mg_coarse_lds = 1;    // 0: keep the level vectors in global memory
int g_mg_coarse_uniform = 1;  // 0: the LDS variant on the levels' own ELL copies (per-job slot counts)

// Host side: the job list of MGM(Lc, u_Lc, f_Lc) and its launcher (used by mg.hip).
struct stk_coarse_plan {
    std::vector<Job> host_jobs;
    Job *dev_jobs = nullptr;
    int n_jobs = 0;
    // LDS arena: rows of u, f, res of every level 0..Lc
    int lds_rows = 0, top_n = 0, top_lu = 0, top_lf = 0;
    const double *top_f = nullptr;
    double *top_u = nullptr;
    // the uniform form: every matrix of the job list at KU slots per row (0: none)
    int KU = 0;
    bool uni_has_m = false;
    CJob *dev_cjobs = nullptr;
    int32_t *u_idx = nullptr, *u_row = nullptr;
    double *u_va = nullptr, *u_vm = nullptr, *u_da = nullptr, *u_dm = nullptr;
    // member matrices (stk_coarse_plan_set_members)
    int Lc = 0;
    uint32_t u_rows = 0;             // rows of the uniform arrays
    int32_t *u_level = nullptr;      // [u_rows] level of a level-matrix row, -1 for transfers
    int n_kinds = 0;
    double *u_vmem = nullptr, *u_dmem = nullptr;  // [n_kinds][u_rows][KU], [n_kinds][u_rows]
    std::vector<char> member_level;  // [Lc + 1] members given for the level?
    bool members_ready = false;      // ... for all of 1..Lc
};

void stk_coarse_plan_free(stk_coarse_plan *p)
{
    if (!p) return;
    if (p->dev_jobs) (void)hipFree(p->dev_jobs);
    for (void *q : {(void *)p->dev_cjobs, (void *)p->u_idx, (void *)p->u_row, (void *)p->u_va, (void *)p->u_vm,
                    (void *)p->u_da, (void *)p->u_dm, (void *)p->u_level, (void *)p->u_vmem, (void *)p->u_dmem})
        if (q) (void)hipFree(q);
    delete p;
}

static Job rows_job(int kind, const stk_ell_rows &e, int pos_begin, int pos_end, const double *x, const double *z,
                    double *y, double alpha, double beta, bool level_matrix, int level)
{
    Job j;
    j.level = level;
    j.kind = kind;
    j.K = e.K;
    j.pos_begin = pos_begin;
    j.pos_end = pos_end;
    j.idx = e.idx;
    j.va = e.va;
    j.vm = e.vm;
    j.row_ids = e.row_ids;
    j.dia_a = e.dia_a;
    j.dia_m = e.dia_m;
    j.diag_free = e.diag_free;
    j.x = x;
    j.z = z;
    j.y = y;
    j.alpha = alpha;
    j.beta = beta;
    j.use_m = level_matrix ? 1 : 0;
    j.lx = j.lz = j.ly = -1;
    return j;
}

// The uniform copies of the job list's matrices (see the head of the file).  Leaves
// p->KU = 0 when a row is longer than 12 slots or an allocation fails.
static void build_uniform(stk_coarse_plan *p)
{
    const std::vector<Job> &J = p->host_jobs;
    int kmax = 0;
    bool has_m = false;
    for (const Job &j : J)
        if (j.kind == JOB_SPMM || j.kind == JOB_GS) {
            kmax = std::max(kmax, (int)j.K);
            has_m = has_m || (j.use_m && j.vm != nullptr);
        }
    const int KU = kmax <= 8 ? 8 : (kmax <= 12 ? 12 : 0);
    if (KU == 0 || sizeof(CJob) != 48) return;
    // one copy per distinct (matrix, position range)
    std::map<std::tuple<const int32_t *, int, int>, uint32_t> where;
    std::vector<CJob> cj(J.size());
    uint32_t total = 0;
    for (size_t n = 0; n < J.size(); ++n) {
        const Job &j = J[n];
        CJob &c = cj[n];
        c.kind = j.kind;
        c.lx = j.lx, c.lz = j.lz, c.ly = j.ly;
        c.flags = (j.use_m ? 1 : 0) | (j.diag_free ? 2 : 0);
        c.alpha = j.alpha, c.beta = j.beta;
        c.pad = 0;
        c.mat_off = 0;
        if (j.kind == JOB_SPMM || j.kind == JOB_GS) {
            c.n_rows = j.pos_end - j.pos_begin;
            const auto key = std::make_tuple(j.idx, (int)j.pos_begin, (int)j.pos_end);
            auto it = where.find(key);
            if (it == where.end()) {
                it = where.emplace(key, total).first;
                total += (uint32_t)c.n_rows;
            }
            c.mat_off = it->second;
        } else {
            c.n_rows = j.pos_end;
        }
    }
    if (total == 0) return;
    // level of every uniform row that belongs to a level matrix (member matrices)
    std::vector<int32_t> row_level(total, -1);
    for (size_t n = 0; n < J.size(); ++n)
        if ((J[n].kind == JOB_SPMM || J[n].kind == JOB_GS) && J[n].use_m)
            for (int r = 0; r < cj[n].n_rows; ++r) row_level[cj[n].mat_off + r] = J[n].level;
    bool ok = hipMalloc((void **)&p->u_level, sizeof(int32_t) * (size_t)total) == hipSuccess &&
              hipMemcpy(p->u_level, row_level.data(), sizeof(int32_t) * (size_t)total, hipMemcpyHostToDevice) ==
                  hipSuccess &&
              hipMalloc((void **)&p->u_idx, sizeof(int32_t) * (size_t)total * KU) == hipSuccess &&
              hipMalloc((void **)&p->u_va, sizeof(double) * (size_t)total * KU) == hipSuccess &&
              hipMalloc((void **)&p->u_da, sizeof(double) * (size_t)total) == hipSuccess &&
              hipMalloc((void **)&p->u_row, sizeof(int32_t) * (size_t)total) == hipSuccess &&
              hipMalloc((void **)&p->dev_cjobs, sizeof(CJob) * cj.size()) == hipSuccess;
    if (ok && has_m)
        ok = hipMalloc((void **)&p->u_vm, sizeof(double) * (size_t)total * KU) == hipSuccess &&
             hipMalloc((void **)&p->u_dm, sizeof(double) * (size_t)total) == hipSuccess;
    if (ok) ok = hipMemcpy(p->dev_cjobs, cj.data(), sizeof(CJob) * cj.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        std::map<std::tuple<const int32_t *, int, int>, bool> done;
        for (const Job &j : J) {
            if (j.kind != JOB_SPMM && j.kind != JOB_GS) continue;
            const auto key = std::make_tuple(j.idx, (int)j.pos_begin, (int)j.pos_end);
            if (done[key]) continue;
            done[key] = true;
            const int rows = j.pos_end - j.pos_begin;
            if (rows <= 0) continue;
            const int threads = rows * KU;
            hipLaunchKernelGGL(repack_uniform_kernel, dim3((threads + 255) / 256), dim3(256), 0, 0, rows, (int)j.K, KU,
                               j.idx, j.va, j.vm, j.row_ids, j.dia_a, j.dia_m, (int)j.pos_begin, where[key], p->u_idx,
                               p->u_va, p->u_vm, p->u_da, p->u_dm, p->u_row);
        }
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        return;  // KU stays 0: the per-job form runs
    }
    p->KU = KU;
    p->uni_has_m = has_m;
    p->u_rows = total;
}

// Entries of one member matrix (CSR on the device, sorted columns) at the slots of the
// uniform rows of `level`: slot (row i, column c) gets C[i, c] -- zero where the matrix
// has no such entry or the slot is padding (both value arrays zero there) -- and the
// row's diagonal entry goes to dmem.
__global__ __launch_bounds__(256) void member_fill_kernel(uint32_t rows, int KU, int level,
                                                          const int32_t *__restrict__ u_level,
                                                          const int32_t *__restrict__ u_idx,
                                                          const int32_t *__restrict__ u_row,
                                                          const double *__restrict__ u_va,
                                                          const double *__restrict__ u_vm,
                                                          const int32_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const double *__restrict__ data, double *vmem, double *dmem,
                                                          int32_t *missing)
{
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    if (item >= rows * (uint32_t)(KU + 1)) return;
    const uint32_t r = item / (KU + 1);
    const int u = (int)(item - r * (KU + 1));
    if (u_level[r] != level) return;
    const int i = u_row[r];
    const bool is_diag = u == KU;
    const size_t e = (size_t)r * KU + (is_diag ? 0 : u);
    const bool real = is_diag || u_va[e] != 0.0 || (u_vm != nullptr && u_vm[e] != 0.0);
    double v = 0.0;
    if (real) {
        const int c = is_diag ? i : u_idx[e];
        int lo = indptr[i], hi = indptr[i + 1];
        while (lo < hi) {  // first entry with column >= c
            const int mid = (lo + hi) >> 1;
            if (indices[mid] < c)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < indptr[i + 1] && indices[lo] == c)
            v = data[lo];
        else if (is_diag)
            atomicAdd(missing, 1);  // a member matrix without a diagonal entry
    }
    if (is_diag)
        dmem[r] = v;
    else
        vmem[e] = v;
}

// Levels 1..Lc of the plan take member matrices (0: the plan has no uniform form).
int stk_coarse_plan_levels(const stk_coarse_plan *p) { return (p && p->KU != 0 && p->uni_has_m) ? p->Lc : 0; }

int stk_coarse_plan_set_members(stk_coarse_plan *p, int level, int n_kinds, int32_t n, const int32_t *const *indptr,
                                const int32_t *const *indices, const double *const *data)
{
    STK_REQUIRE(p && p->KU != 0 && p->uni_has_m, "stk_mg_set_member_matrices: the plan has no uniform coarse form");
    STK_REQUIRE(level >= 1 && level <= p->Lc, "stk_mg_set_member_matrices: level %d is not one of 1..%d", level, p->Lc);
    STK_REQUIRE(n_kinds >= 1 && (p->n_kinds == 0 || p->n_kinds == n_kinds),
                "stk_mg_set_member_matrices: %d matrices, %d at the first call", n_kinds, p->n_kinds);
    if (p->u_vmem == nullptr) {
        const size_t nv = (size_t)n_kinds * p->u_rows * p->KU, nd = (size_t)n_kinds * p->u_rows;
        STK_HIP(hipMalloc((void **)&p->u_vmem, sizeof(double) * nv));
        STK_HIP(hipMalloc((void **)&p->u_dmem, sizeof(double) * nd));
        STK_HIP(hipMemset(p->u_vmem, 0, sizeof(double) * nv));
        STK_HIP(hipMemset(p->u_dmem, 0, sizeof(double) * nd));
        p->n_kinds = n_kinds;
        p->member_level.assign(p->Lc + 1, 0);
    }
    int32_t *missing = nullptr;
    STK_HIP(hipMalloc((void **)&missing, sizeof(int32_t)));
    STK_HIP(hipMemset(missing, 0, sizeof(int32_t)));
    int rc = 0;
    for (int k = 0; k < n_kinds && rc == 0; ++k) {
        if (indptr[k] == nullptr) continue;  // a kind no time slice names (the family's kind 0)
        const int64_t nnz = indptr[k][n];
        int32_t *d_ptr = nullptr, *d_idx = nullptr;
        double *d_val = nullptr;
        if (hipMalloc((void **)&d_ptr, sizeof(int32_t) * (size_t)(n + 1)) != hipSuccess ||
            hipMalloc((void **)&d_idx, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)) != hipSuccess ||
            hipMalloc((void **)&d_val, sizeof(double) * (size_t)std::max<int64_t>(nnz, 1)) != hipSuccess ||
            hipMemcpy(d_ptr, indptr[k], sizeof(int32_t) * (size_t)(n + 1), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_idx, indices[k], sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_val, data[k], sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice) != hipSuccess) {
            stk_set_error("stk_mg_set_member_matrices: device copy of matrix %d failed", k);
            rc = 1;
        } else {
            const uint32_t items = p->u_rows * (uint32_t)(p->KU + 1);
            hipLaunchKernelGGL(member_fill_kernel, dim3((items + 255) / 256), dim3(256), 0, 0, p->u_rows, p->KU, level,
                               p->u_level, p->u_idx, p->u_row, p->u_va, p->u_vm, d_ptr, d_idx, d_val,
                               p->u_vmem + (size_t)k * p->u_rows * p->KU, p->u_dmem + (size_t)k * p->u_rows, missing);
            if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                stk_set_error("stk_mg_set_member_matrices: fill kernel failed");
                rc = 1;
            }
        }
        (void)hipFree(d_ptr), (void)hipFree(d_idx), (void)hipFree(d_val);
    }
    int32_t n_missing = 0;
    if (rc == 0 && hipMemcpy(&n_missing, missing, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    (void)hipFree(missing);
    if (rc) return rc;
    STK_REQUIRE(n_missing == 0, "stk_mg_set_member_matrices: %d rows of level %d lack their diagonal entry", n_missing,
                level);
    p->member_level[level] = 1;
    p->members_ready = true;
    for (int l = 1; l <= p->Lc; ++l) p->members_ready = p->members_ready && p->member_level[l];
    return 0;
}

stk_coarse_plan *stk_coarse_plan_build(const stk_coarse_level *lv, int Lc, int smoothsteps)
{
    stk_coarse_plan *p = new stk_coarse_plan();
    std::vector<Job> &J = p->host_jobs;
    auto smooth = [&](int j, bool backward) {
        const stk_coarse_level &L = lv[j];
        const stk_ell_rows &e = backward ? L.bwd : L.fwd;
        const int32_t *pos = backward ? L.bwd_pos : L.fwd_pos;
        const int ng = backward ? L.n_bwd : L.n_fwd;
        for (int it = 0; it < smoothsteps; ++it)
            for (int g = 0; g < ng; ++g)
                if (pos[g + 1] > pos[g])
                    J.push_back(rows_job(JOB_GS, e, pos[g], pos[g + 1], L.u, L.f, L.u, 0.0, 0.0, true, j));
    };
    for (int j = Lc; j >= 1; --j) {
        const stk_coarse_level &L = lv[j], &C = lv[j - 1];
        smooth(j, false);
        J.push_back(rows_job(JOB_SPMM, L.a, 0, L.a.n_pos, L.u, L.f, L.res, 1.0, -1.0, true, j));   // r = A u - f
        J.push_back(rows_job(JOB_SPMM, L.r, 0, L.r.n_pos, L.res, nullptr, C.f, 1.0, 0.0, false, j));  // d = R r
        Job z;
        z.kind = JOB_ZERO;
        z.K = 0;
        z.pos_begin = 0;
        z.pos_end = C.n;
        z.idx = nullptr;
        z.va = z.vm = z.dia_a = z.dia_m = z.x = z.z = nullptr;
        z.row_ids = nullptr;
        z.y = C.u;
        z.alpha = z.beta = 0.0;
        z.use_m = z.diag_free = 0;
        z.lx = z.lz = z.ly = -1;
        z.level = j - 1;
        if (j - 1 >= 1) J.push_back(z);  // level 0 is overwritten by the exact solve
    }
    {
        Job c;
        c.kind = JOB_COARSE;
        c.K = 0;
        c.pos_begin = 0;
        c.pos_end = lv[0].n;
        c.idx = nullptr;
        c.va = c.vm = c.dia_a = c.dia_m = c.x = nullptr;
        c.row_ids = nullptr;
        c.z = lv[0].f;
        c.y = lv[0].u;
        c.alpha = c.beta = 0.0;
        c.use_m = c.diag_free = 0;
        c.lx = c.lz = c.ly = -1;
        c.level = 0;
        J.push_back(c);
    }
    for (int j = 1; j <= Lc; ++j) {
        const stk_coarse_level &L = lv[j], &C = lv[j - 1];
        J.push_back(rows_job(JOB_SPMM, L.p, 0, L.p.n_pos, C.u, L.u, L.u, -1.0, 1.0, false, j));  // u -= P u_c
        smooth(j, true);
    }
    // LDS arena offsets of every global workspace the jobs name
    {
        std::vector<std::pair<const double *, int>> where;
        int rows = 0;
        for (int j = 0; j <= Lc; ++j)
            for (const double *v : {(const double *)lv[j].u, (const double *)lv[j].f, (const double *)lv[j].res}) {
                where.push_back({v, rows});
                rows += lv[j].n;
            }
        auto off = [&](const double *v) {
            for (auto &w : where)
                if (w.first == v) return w.second;
            return -1;
        };
        for (Job &j : J) {
            j.lx = off(j.x);
            j.lz = off(j.z);
            j.ly = off(j.y);
        }
        p->lds_rows = rows;
        p->top_n = lv[Lc].n;
        p->top_lu = off(lv[Lc].u);
        p->top_lf = off(lv[Lc].f);
        p->top_f = lv[Lc].f;
        p->top_u = lv[Lc].u;
    }
    p->n_jobs = (int)J.size();
    p->Lc = Lc;
    if (hipMalloc((void **)&p->dev_jobs, sizeof(Job) * J.size()) != hipSuccess ||
        hipMemcpy(p->dev_jobs, J.data(), sizeof(Job) * J.size(), hipMemcpyHostToDevice) != hipSuccess) {
        stk_coarse_plan_free(p);
        return nullptr;
    }
    build_uniform(p);  // optional: the plan works without it
    return p;
}

// Does the job list run with its level vectors in LDS?  That variant starts the cut
// level's u from zero itself and writes every entry of it at the end: the caller
// need not zero it in memory.
bool stk_coarse_plan_in_lds(const stk_coarse_plan *p)
{
    return g_mg_coarse_lds && g_mg_coarse_pairs <= 1 && sizeof(double) * (size_t)p->lds_rows <= 144 * 1024;
}

int g_mg_coarse_static_fetch = 0;  // tuning key "mg_coarse_static_fetch": the same loads for every row (A/B)

template <int KU, bool HAS_M, bool PAIR, int UBS, bool SF>
static int launch_uniform_sf(dim3 grid, size_t lds, hipStream_t st, const CoarseArgs &a, const UniArgs &m)
{
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        STK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mg_coarse_uniform_kernel<KU, HAS_M, PAIR, UBS, SF>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((mg_coarse_uniform_kernel<KU, HAS_M, PAIR, UBS, SF>), grid, dim3(UBS), lds, st, a, m);
    STK_LAUNCH_CHECK();
    return 0;
}

template <int KU, bool HAS_M, bool PAIR, int UBS>
static int launch_uniform_one(dim3 grid, size_t lds, hipStream_t st, const CoarseArgs &a, const UniArgs &m)
{
    return g_mg_coarse_static_fetch ? launch_uniform_sf<KU, HAS_M, PAIR, UBS, true>(grid, lds, st, a, m)
                                    : launch_uniform_sf<KU, HAS_M, PAIR, UBS, false>(grid, lds, st, a, m);
}

// 1024 threads per time step where the registers allow: a level-5 job is then one or
// four rows per thread instead of two or eight.
extern int g_mg_coarse_uniform;
static int launch_uniform(int KU, bool has_m, bool pair, dim3 grid, size_t lds, hipStream_t st, const CoarseArgs &a,
                          const UniArgs &m)
{
    if (KU == 8) {
        if (g_mg_coarse_uniform == 2) {  // A/B: 512 threads throughout
            if (has_m) return pair ? launch_uniform_one<8, true, true, 512>(grid, lds, st, a, m)
                                   : launch_uniform_one<8, true, false, 512>(grid, lds, st, a, m);
            return pair ? launch_uniform_one<8, false, true, 512>(grid, lds, st, a, m)
                        : launch_uniform_one<8, false, false, 512>(grid, lds, st, a, m);
        }
        // (with the second value array two register sets of 8 slots do not fit the 128
        // registers a 1024-thread workgroup leaves a thread)
        if (has_m) return pair ? launch_uniform_one<8, true, true, 512>(grid, lds, st, a, m)
                               : launch_uniform_one<8, true, false, 512>(grid, lds, st, a, m);
        return pair ? launch_uniform_one<8, false, true, 512>(grid, lds, st, a, m)
                    : launch_uniform_one<8, false, false, 1024>(grid, lds, st, a, m);
    }
    if (has_m) return pair ? launch_uniform_one<12, true, true, 512>(grid, lds, st, a, m)
                           : launch_uniform_one<12, true, false, 512>(grid, lds, st, a, m);
    return pair ? launch_uniform_one<12, false, true, 512>(grid, lds, st, a, m)
                : launch_uniform_one<12, false, false, 512>(grid, lds, st, a, m);
}

int stk_coarse_plan_run(const stk_coarse_plan *p, hipStream_t st, int n_loc, int ld, double ca, const double *cm,
                        const int32_t *kind, const double *coarse_inv)
{
    CoarseArgs a;
    a.jobs = p->dev_jobs;
    a.n_jobs = p->n_jobs;
    a.n_loc = n_loc;
    a.ld = ld;
    a.pairs_per_wg = g_mg_coarse_pairs > 0 ? g_mg_coarse_pairs : 1;
    a.ca = ca;
    a.cm = cm;
    a.kind = cm ? kind : nullptr;
    a.coarse_inv = coarse_inv;
    a.coarse_scale = cm ? 1.0 : 1.0 / ca;
    const int all_pairs = (n_loc + 1) / 2;
    a.lds_rows = p->lds_rows;
    a.top_n = p->top_n;
    a.top_lu = p->top_lu;
    a.top_lf = p->top_lf;
    a.top_f = p->top_f;
    a.top_u = p->top_u;
    // level vectors in LDS: pairs of time steps per workgroup if the arena fits,
    // else single time steps (half the arena)
    const size_t lds_pair = sizeof(double2) * (size_t)p->lds_rows, lds_one = sizeof(double) * (size_t)p->lds_rows;
    const size_t lds_max = 144 * 1024;
    if (stk_coarse_plan_in_lds(p)) {
        static bool attr_set = false;
        if (!attr_set) {
            STK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mg_coarse_lds_kernel<true, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            STK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mg_coarse_lds_kernel<false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            STK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mg_coarse_lds_kernel<true, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            STK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mg_coarse_lds_kernel<false, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            attr_set = true;
        }
        const bool pair = lds_pair <= lds_max;
        const dim3 grid(pair ? all_pairs : n_loc);
        const size_t lds = pair ? lds_pair : lds_one;
        if (g_mg_coarse_uniform && p->KU != 0 && (!cm || p->uni_has_m)) {
            UniArgs m;
            m.jobs = p->dev_cjobs;
            m.idx = p->u_idx;
            m.va = p->u_va;
            m.vm = p->u_vm;
            m.dia_a = p->u_da;
            m.dia_m = p->u_dm;
            m.row = p->u_row;
            m.vmem = p->members_ready ? p->u_vmem : nullptr;
            m.dmem = p->members_ready ? p->u_dmem : nullptr;
            m.mem_rows = p->u_rows;
            const size_t lds_u = lds + (pair ? 0 : sizeof(double) * (p->lds_rows & 1)) + sizeof(CJob) * (size_t)p->n_jobs;
            if (lds_u <= 160 * 1024 - 512) return launch_uniform(p->KU, cm != nullptr, pair, grid, lds_u, st, a, m);
        }
        if (cm && pair)
            hipLaunchKernelGGL((mg_coarse_lds_kernel<true, true>), grid, dim3(CBS), lds, st, a);
        else if (cm)
            hipLaunchKernelGGL((mg_coarse_lds_kernel<true, false>), grid, dim3(CBS), lds, st, a);
        else if (pair)
            hipLaunchKernelGGL((mg_coarse_lds_kernel<false, true>), grid, dim3(CBS), lds, st, a);
        else
            hipLaunchKernelGGL((mg_coarse_lds_kernel<false, false>), grid, dim3(CBS), lds, st, a);
        STK_LAUNCH_CHECK();
        return 0;
    }
    const unsigned grid = (unsigned)((all_pairs + a.pairs_per_wg - 1) / a.pairs_per_wg);
    if (cm)
        hipLaunchKernelGGL(mg_coarse_kernel<true>, dim3(grid), dim3(CBS), 0, st, a);
    else
        hipLaunchKernelGGL(mg_coarse_kernel<false>, dim3(grid), dim3(CBS), 0, st, a);
    STK_LAUNCH_CHECK();
    return 0;
}
