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

}  // namespace

int g_mg_coarse_pairs = 0;  // time pairs per workgroup (0 = default: 1, vectors resident in LDS)
int g_mg_coarse_lds = 1;    // 0: keep the level vectors in global memory

// Host side: the job list of MGM(Lc, u_Lc, f_Lc) and its launcher (used by mg.hip).
struct stk_coarse_plan {
    std::vector<Job> host_jobs;
    Job *dev_jobs = nullptr;
    int n_jobs = 0;
    // LDS arena: rows of u, f, res of every level 0..Lc
    int lds_rows = 0, top_n = 0, top_lu = 0, top_lf = 0;
    const double *top_f = nullptr;
    double *top_u = nullptr;
};

void stk_coarse_plan_free(stk_coarse_plan *p)
{
    if (!p) return;
    if (p->dev_jobs) (void)hipFree(p->dev_jobs);
    delete p;
}

static Job rows_job(int kind, const stk_ell_rows &e, int pos_begin, int pos_end, const double *x, const double *z,
                    double *y, double alpha, double beta, bool level_matrix)
{
    Job j;
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
                    J.push_back(rows_job(JOB_GS, e, pos[g], pos[g + 1], L.u, L.f, L.u, 0.0, 0.0, true));
    };
    for (int j = Lc; j >= 1; --j) {
        const stk_coarse_level &L = lv[j], &C = lv[j - 1];
        smooth(j, false);
        J.push_back(rows_job(JOB_SPMM, L.a, 0, L.a.n_pos, L.u, L.f, L.res, 1.0, -1.0, true));   // r = A u - f
        J.push_back(rows_job(JOB_SPMM, L.r, 0, L.r.n_pos, L.res, nullptr, C.f, 1.0, 0.0, false));  // d = R r
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
        J.push_back(c);
    }
    for (int j = 1; j <= Lc; ++j) {
        const stk_coarse_level &L = lv[j], &C = lv[j - 1];
        J.push_back(rows_job(JOB_SPMM, L.p, 0, L.p.n_pos, C.u, L.u, L.u, -1.0, 1.0, false));  // u -= P u_c
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
    if (hipMalloc((void **)&p->dev_jobs, sizeof(Job) * J.size()) != hipSuccess ||
        hipMemcpy(p->dev_jobs, J.data(), sizeof(Job) * J.size(), hipMemcpyHostToDevice) != hipSuccess) {
        stk_coarse_plan_free(p);
        return nullptr;
    }
    return p;
}

// Does the job list run with its level vectors in LDS?  That variant starts the cut
// level's u from zero itself and writes every entry of it at the end: the caller
// need not zero it in memory.
bool stk_coarse_plan_in_lds(const stk_coarse_plan *p)
{
    return g_mg_coarse_lds && g_mg_coarse_pairs <= 1 && sizeof(double) * (size_t)p->lds_rows <= 144 * 1024;
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
