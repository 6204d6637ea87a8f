// Sparse direct solve on the device:  x = A^-1 b  for all time steps of a slab at once,
// from the factors Pr A Pc = L U of scipy.sparse.linalg.splu (SuperLU) -- the
// reference's InvLinOp (source/linop.py:18-26: `self.inv = splu(mat)`, `_matmat` =
// `self.inv.solve`), which heateq_mpi.py:155-157 uses for precond='direct'.  The
// factorisation stays where the reference has it (SuperLU on the host, at set-up);
// what runs per apply -- the two triangular solves and the two permutations -- runs
// here, so a direct preconditioner needs no host round trip of the slab.
//
// Level scheduling: row k of L y = c depends on the rows j < k of its entries; rows
// of equal depth in that graph are independent (the dependency depths the
// Gauss-Seidel plans use, csrc/ell_build.hip, on a triangular matrix).  SuperLU's
// column ordering gives the elimination tree of a 2D mesh a few WIDE levels at the
// leaves (32 levels hold 11 081 of the 16 129 rows of A_x at J_space = 6) and a long
// tail of narrow ones towards the root (1 425 levels with 3.5 rows on average, rows
// of up to 1 141 entries).  A wide level is one launch over the whole chip; a RUN of
// narrow levels is one launch of ONE workgroup that walks the levels with a
// workgroup barrier between them (a launch per narrow level would cost 5 us each).
//
// A (row, time step) item is served by SP lanes: lane s adds the entries e = s, s +
// SP, .. of the row with fused multiply-adds, the SP partial sums meet in a fixed
// shuffle tree.  SP and the tree do not depend on the slab length, so a column of the
// result is the same doubles on any partition of the time axis.
#include <algorithm>
#include <vector>

#include "stk_common.h"

namespace {

constexpr int SP = 16;    // lanes per (row, time step)
constexpr int TBS = 1024; // threads of a workgroup
constexpr int WIDE = 64;  // a level with at least this many rows is a launch of its own

struct TriDev {
    int32_t n = 0, n_levels = 0;
    int32_t *lvl_ptr = nullptr;  // [n_levels + 1] into the level-sorted row list
    int32_t *row = nullptr;      // [n] row index, sorted by level
    int32_t *ptr = nullptr;      // [n + 1] entries of the sorted rows (off-diagonal part)
    int32_t *col = nullptr;
    double *val = nullptr;
    double *dinv = nullptr;  // [n] 1 / diagonal of the sorted rows
    std::vector<int32_t> lvl_rows;  // host: rows per level
    // segments: consecutive levels [a, b) per launch; wide levels stand alone
    std::vector<std::pair<int32_t, int32_t>> segments;
};

struct TriArgs {
    const int32_t *lvl_ptr, *row, *ptr, *col;
    const double *val, *dinv;
    const int32_t *src_perm;  // rhs row of solution row k (NULL: k), read from `rhs`
    const int32_t *dst_perm;  // ALSO write solution row k to out[dst_perm[k]] (NULL: no second copy)
    const double *rhs;        // right-hand side slab (may be the solution slab itself)
    double *u;                // solution slab, in place
    double *out;
    int32_t n_loc, ld;
    int32_t lvl_begin, lvl_end;
};

// One launch = the levels [lvl_begin, lvl_end).  More than one level: a single
// workgroup (gridDim.x == 1), __syncthreads between levels -- the rows a level reads
// were written by waves of this workgroup, through this CU's L1.
__global__ __launch_bounds__(TBS) void sptrsv_kernel(const TriArgs a)
{
    const int lane_s = threadIdx.x & (SP - 1);
    const int item0 = (int)((blockIdx.x * (unsigned)TBS + threadIdx.x) / SP);
    const int item_stride = (int)(gridDim.x * (unsigned)TBS / SP);
    for (int lvl = a.lvl_begin; lvl < a.lvl_end; ++lvl) {
        const int r0 = a.lvl_ptr[lvl], r1 = a.lvl_ptr[lvl + 1];
        const int items = (r1 - r0) * a.n_loc;
        // all lanes of a wavefront take part in the shuffles: round the trip count up
        const int trips = (items + item_stride - 1) / item_stride;
        for (int trip = 0, item = item0; trip < trips; ++trip, item += item_stride) {
            const bool live = item < items;
            const int q = live ? r0 + item / a.n_loc : r0;
            const int t = live ? item - (q - r0) * a.n_loc : 0;
            const int e0 = a.ptr[q], e1 = live ? a.ptr[q + 1] : e0;
            const double *ut = a.u + t;
            double acc = 0.0;
            int e = e0 + lane_s;
            for (; e + 3 * SP < e1; e += 4 * SP) {  // four independent gathers in flight
                const double v0 = a.val[e], v1 = a.val[e + SP], v2 = a.val[e + 2 * SP], v3 = a.val[e + 3 * SP];
                const double x0 = ut[(size_t)a.col[e] * a.ld], x1 = ut[(size_t)a.col[e + SP] * a.ld];
                const double x2 = ut[(size_t)a.col[e + 2 * SP] * a.ld], x3 = ut[(size_t)a.col[e + 3 * SP] * a.ld];
                acc = fma(v0, x0, acc);
                acc = fma(v1, x1, acc);
                acc = fma(v2, x2, acc);
                acc = fma(v3, x3, acc);
            }
            for (; e < e1; e += SP) acc = fma(a.val[e], ut[(size_t)a.col[e] * a.ld], acc);
#pragma unroll
            for (int off = SP / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, SP);
            if (live && lane_s == 0) {
                const int k = a.row[q];
                const int src = a.src_perm ? a.src_perm[k] : k;
                const double x = (a.rhs[(size_t)src * a.ld + t] - acc) * a.dinv[q];
                a.u[(size_t)k * a.ld + t] = x;
                if (a.dst_perm) {
                    double *o = a.out + (size_t)a.dst_perm[k] * a.ld;
                    o[t] = x;
                    if (t == a.n_loc - 1)  // the padding of a slab row stays zero
                        for (int tt = a.n_loc; tt < a.ld; ++tt) o[tt] = 0.0;
                }
            }
        }
        if (lvl + 1 < a.lvl_end) __syncthreads();
    }
}

template <class T>
int upload(const std::vector<T> &h, T **d)
{
    STK_HIP(hipMalloc((void **)d, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) STK_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

void release(TriDev &t)
{
    for (void *p : {(void *)t.lvl_ptr, (void *)t.row, (void *)t.ptr, (void *)t.col, (void *)t.val, (void *)t.dinv})
        (void)hipFree(p);
    t = TriDev();
}

// Level-sorted device copy of a triangular CSR matrix (sorted column indices; the
// diagonal entry present in every row, or `unit` for an implicit / explicit 1).
int build(int32_t n, const int32_t *indptr, const int32_t *indices, const double *data, bool lower, bool unit,
          TriDev *out)
{
    std::vector<int32_t> depth(n, 0);
    std::vector<double> diag(n, unit ? 1.0 : 0.0);
    auto visit = [&](int i) {
        int d = 0;
        for (int e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int j = indices[e];
            if (j == i) {
                if (!unit) diag[i] = data[e];
            } else if (lower ? j < i : j > i) {
                d = std::max(d, depth[j] + 1);
            } else {
                return -1;  // an entry on the wrong side of the diagonal
            }
        }
        depth[i] = d;
        return 0;
    };
    if (lower) {
        for (int i = 0; i < n; ++i)
            if (visit(i)) goto bad;
    } else {
        for (int i = n - 1; i >= 0; --i)
            if (visit(i)) goto bad;
    }
    {
        int n_levels = 0;
        for (int i = 0; i < n; ++i) {
            STK_REQUIRE(diag[i] != 0.0, "stk_lu_create: zero diagonal in row %d of %s", i, lower ? "L" : "U");
            n_levels = std::max(n_levels, depth[i] + 1);
        }
        std::vector<int32_t> lvl_ptr(n_levels + 1, 0);
        for (int i = 0; i < n; ++i) ++lvl_ptr[depth[i] + 1];
        for (int l = 0; l < n_levels; ++l) lvl_ptr[l + 1] += lvl_ptr[l];
        std::vector<int32_t> fill(lvl_ptr.begin(), lvl_ptr.end() - 1), row(n), ptr(n + 1, 0);
        for (int i = 0; i < n; ++i) row[fill[depth[i]]++] = i;  // increasing row index inside a level
        std::vector<int32_t> col;
        std::vector<double> val, dinv(n);
        col.reserve(indptr[n]), val.reserve(indptr[n]);
        for (int q = 0; q < n; ++q) {
            const int i = row[q];
            for (int e = indptr[i]; e < indptr[i + 1]; ++e)
                if (indices[e] != i) col.push_back(indices[e]), val.push_back(data[e]);
            ptr[q + 1] = (int32_t)col.size();
            dinv[q] = 1.0 / diag[i];
        }
        TriDev t;
        t.n = n, t.n_levels = n_levels;
        t.lvl_rows.resize(n_levels);
        for (int l = 0; l < n_levels; ++l) t.lvl_rows[l] = lvl_ptr[l + 1] - lvl_ptr[l];
        for (int l = 0; l < n_levels;) {
            if (t.lvl_rows[l] >= WIDE) {
                t.segments.push_back({l, l + 1});
                ++l;
                continue;
            }
            int b = l;
            while (b < n_levels && t.lvl_rows[b] < WIDE) ++b;
            t.segments.push_back({l, b});
            l = b;
        }
        if (upload(lvl_ptr, &t.lvl_ptr) || upload(row, &t.row) || upload(ptr, &t.ptr) || upload(col, &t.col) ||
            upload(val, &t.val) || upload(dinv, &t.dinv)) {
            release(t);
            return 1;
        }
        *out = t;
        return 0;
    }
bad:
    stk_set_error("stk_lu_create: %s has an entry on the wrong side of its diagonal", lower ? "L" : "U");
    return 1;
}

int solve(hipStream_t st, const TriDev &t, int32_t n_loc, int32_t ld, const int32_t *src_perm, const double *rhs,
          double *u, const int32_t *dst_perm, double *out)
{
    TriArgs a;
    a.lvl_ptr = t.lvl_ptr, a.row = t.row, a.ptr = t.ptr, a.col = t.col, a.val = t.val, a.dinv = t.dinv;
    a.src_perm = src_perm, a.dst_perm = dst_perm, a.rhs = rhs, a.u = u, a.out = out;
    a.n_loc = n_loc, a.ld = ld;
    for (const auto &seg : t.segments) {
        a.lvl_begin = seg.first, a.lvl_end = seg.second;
        unsigned grid = 1;
        if (seg.second - seg.first == 1) {
            const int64_t items = (int64_t)t.lvl_rows[seg.first] * n_loc;
            grid = (unsigned)std::min<int64_t>((items * SP + TBS - 1) / TBS, 4096);
            if (grid < 1) grid = 1;
        }
        hipLaunchKernelGGL(sptrsv_kernel, dim3(grid), dim3(TBS), 0, st, a);
        STK_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

struct stk_lu {
    int32_t n = 0;
    TriDev L, U;
    int32_t *src_perm = nullptr;  // row of b that row k of L y = Pr b reads
    int32_t *dst_perm = nullptr;  // row of x that receives row k of z = U^-1 y
};

extern "C" int stk_lu_destroy(stk_lu *lu)
{
    if (!lu) return 0;
    release(lu->L), release(lu->U);
    (void)hipFree(lu->src_perm);
    (void)hipFree(lu->dst_perm);
    delete lu;
    return 0;
}

extern "C" int stk_lu_create(int32_t n, const int32_t *L_indptr, const int32_t *L_indices, const double *L_data,
                             const int32_t *U_indptr, const int32_t *U_indices, const double *U_data,
                             const int32_t *perm_r, const int32_t *perm_c, stk_lu **out)
{
    STK_REQUIRE(n > 0 && L_indptr && L_indices && L_data && U_indptr && U_indices && U_data && out,
                "stk_lu_create: null argument or n=%d", n);
    // SciPy: Pr[perm_r[i], i] = 1, Pc[i, perm_c[i]] = 1, Pr A Pc = L U.  A x = b:
    //   (Pr b)[perm_r[i]] = b[i];  z = U^-1 L^-1 Pr b;  x[i] = z[perm_c[i]]
    std::vector<int32_t> src(n), dst(n);
    std::vector<char> seen_r(n, 0), seen_c(n, 0);
    for (int i = 0; i < n; ++i) {
        const int r = perm_r ? perm_r[i] : i, c = perm_c ? perm_c[i] : i;
        STK_REQUIRE(r >= 0 && r < n && c >= 0 && c < n && !seen_r[r] && !seen_c[c],
                    "stk_lu_create: perm_r / perm_c is not a permutation (entry %d)", i);
        seen_r[r] = seen_c[c] = 1;
        src[r] = i;  // row r of Pr b is b[i]
        dst[c] = i;  // row c of z goes to x[i]
    }
    stk_lu *lu = new stk_lu;
    lu->n = n;
    if (build(n, L_indptr, L_indices, L_data, true, true, &lu->L) ||
        build(n, U_indptr, U_indices, U_data, false, false, &lu->U) || upload(src, &lu->src_perm) ||
        upload(dst, &lu->dst_perm)) {
        stk_lu_destroy(lu);
        return 1;
    }
    *out = lu;
    return 0;
}

extern "C" int stk_lu_info(const stk_lu *lu, int32_t *levels_L, int32_t *levels_U, int32_t *launches)
{
    STK_REQUIRE(lu, "stk_lu_info: null plan");
    if (levels_L) *levels_L = lu->L.n_levels;
    if (levels_U) *levels_U = lu->U.n_levels;
    if (launches) *launches = (int32_t)(lu->L.segments.size() + lu->U.segments.size());
    return 0;
}

extern "C" int stk_lu_solve(stk_lu *lu, void *stream, int32_t n_loc, int32_t ld, const double *b, double *x,
                            double *work)
{
    const stk_timed timed_(STK_OP_SPACE, stream);
    STK_REQUIRE(lu && b && x && work, "stk_lu_solve: null argument");
    STK_REQUIRE(n_loc > 0 && ld >= n_loc, "stk_lu_solve: bad sizes n_loc=%d ld=%d", n_loc, ld);
    STK_REQUIRE(work != b && work != x, "stk_lu_solve: work aliases b or x");
    hipStream_t st = stk_stream(stream);
    // L y = Pr b into work; U z = y in place on work, every row also to its place in x
    int rc = solve(st, lu->L, n_loc, ld, lu->src_perm, b, work, nullptr, nullptr);
    if (rc) return rc;
    return solve(st, lu->U, n_loc, ld, nullptr, work, work, lu->dst_perm, x);
}
