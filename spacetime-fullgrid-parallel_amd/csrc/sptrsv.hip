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
    std::vector<int32_t> depth;     // host: level of every row of the matrix
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

// Level-sorted device copy of (a part of) a triangular CSR matrix with sorted column
// indices: the rows i with take(i), their entries j with keep(j); the dependency depth
// of a row counts its entries with deps(j) only.  one_level: all rows in one level
// (none of the kept entries is a dependency among them).  unit: diagonal 1, implicit
// or explicit.  no_diag: the rows divide by 1 (a product, not a solve).
template <class Take, class Keep, class Deps>
int build(int32_t n, const int32_t *indptr, const int32_t *indices, const double *data, bool lower, bool unit,
          Take take, Keep keep, Deps deps, bool one_level, bool no_diag, TriDev *out)
{
    const char *name = lower ? "L" : "U";
    std::vector<int32_t> depth(n, 0);
    std::vector<double> diag(n, unit ? 1.0 : 0.0);
    auto visit = [&](int i) {
        int d = 0;
        for (int e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int j = indices[e];
            if (j == i) {
                if (!unit) diag[i] = data[e];
            } else if (lower ? j < i : j > i) {
                if (deps(j)) d = std::max(d, depth[j] + 1);
            } else {
                return -1;  // an entry on the wrong side of the diagonal
            }
        }
        depth[i] = one_level ? 0 : d;
        return 0;
    };
    bool bad = false;
    if (lower) {
        for (int i = 0; i < n && !bad; ++i)
            if (take(i)) bad = visit(i) != 0;
    } else {
        for (int i = n - 1; i >= 0 && !bad; --i)
            if (take(i)) bad = visit(i) != 0;
    }
    STK_REQUIRE(!bad, "stk_lu_create: %s has an entry on the wrong side of its diagonal", name);
    int n_levels = 0, n_rows = 0;
    for (int i = 0; i < n; ++i) {
        if (!take(i)) continue;
        STK_REQUIRE(diag[i] != 0.0, "stk_lu_create: zero diagonal in row %d of %s", i, name);
        n_levels = std::max(n_levels, depth[i] + 1);
        ++n_rows;
    }
    std::vector<int32_t> lvl_ptr(n_levels + 1, 0);
    for (int i = 0; i < n; ++i)
        if (take(i)) ++lvl_ptr[depth[i] + 1];
    for (int l = 0; l < n_levels; ++l) lvl_ptr[l + 1] += lvl_ptr[l];
    std::vector<int32_t> fill(lvl_ptr.begin(), lvl_ptr.end() - 1), row(n_rows), ptr(n_rows + 1, 0);
    for (int i = 0; i < n; ++i)
        if (take(i)) row[fill[depth[i]]++] = i;  // increasing row index inside a level
    std::vector<int32_t> col;
    std::vector<double> val, dinv(n_rows);
    for (int q = 0; q < n_rows; ++q) {
        const int i = row[q];
        for (int e = indptr[i]; e < indptr[i + 1]; ++e)
            if (indices[e] != i && keep(indices[e])) col.push_back(indices[e]), val.push_back(data[e]);
        ptr[q + 1] = (int32_t)col.size();
        dinv[q] = no_diag ? 1.0 : 1.0 / diag[i];
    }
    TriDev t;
    t.n = n_rows, t.n_levels = n_levels;
    t.lvl_rows.resize(n_levels);
    for (int l = 0; l < n_levels; ++l) t.lvl_rows[l] = lvl_ptr[l + 1] - lvl_ptr[l];
    for (int l = 0; l < n_levels;) {
        if (t.lvl_rows[l] >= WIDE || one_level) {
            t.segments.push_back({l, l + 1});
            ++l;
            continue;
        }
        int b = l;
        while (b < n_levels && t.lvl_rows[b] < WIDE) ++b;
        t.segments.push_back({l, b});
        l = b;
    }
    t.depth = depth;
    if (upload(lvl_ptr, &t.lvl_ptr) || upload(row, &t.row) || upload(ptr, &t.ptr) || upload(col, &t.col) ||
        upload(val, &t.val) || upload(dinv, &t.dinv)) {
        release(t);
        return 1;
    }
    *out = t;
    return 0;
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

// ---- the top of the elimination tree as a dense block ------------------------------
// The narrow levels are the separators near the root: a few thousand rows that depend on
// each other almost densely, one row after the other.  S = the rows from the first narrow
// level of L on (ancestors of a row of S are in S).  The caller hands the blocks L[S, S] and
// U[S, S] over as dense matrices whose DIAGONAL BLOCKS of `block` rows are replaced by their
// inverses (stk_lu_set_top_inverse), and the hundreds of dependent levels become
//   forward:  head levels;  d_S = (Pr b)_S - L_SH y_H  (one launch);
//             per block k, ascending:   t = d_k - L[k, <k] y_<k;   y_k = inv(L_kk) t
//   backward: per block k, descending:  t = y_k - U[k, >k] z_>k;   z_k = inv(U_kk) t;
//             head levels (their rows read z_S like any column)
// two launches per block (the product with the solved part, the block's inverse), each spread
// over the chip.
// Blocks of 256 rows keep the accuracy of substitution (an explicit inverse of the whole
// 2 578-row block is 5-10 times less accurate, 1.4e-15 against 2e-16 -- enough to move the
// last entry of a converged r.Pr history by 6e-11).  SP lanes per
// row, lane s taking the columns s, s + SP, ...: one summation shape whatever the slab.
// phase 0: scratch[i - a][t] = u[S[i]][t] - sum_{j outside the block, on its solved side} mat[i][j] u[S[j]][t]
// phase 1: u[S[i]][t] = sum_{j in the block} inv_block[i][j] scratch[j - a][t]
// one (row, time step) item per SP lanes, the items of a launch spread over the chip.
__global__ __launch_bounds__(256) void block_top_kernel(int32_t n_top, const int32_t *__restrict__ rows,
                                                        const double *__restrict__ mat, int32_t a, int32_t e,
                                                        int lower, int phase, int32_t n_loc, int32_t ld, double *u,
                                                        double *scratch, const int32_t *__restrict__ dst_perm,
                                                        double *out)
{
    const int lane_s = threadIdx.x & (SP - 1);
    const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) / SP;  // whole lane groups are in or out
    if (item >= (int64_t)(e - a) * n_loc) return;
    const int i = a + (int)(item / n_loc), t = (int)(item - (int64_t)(i - a) * n_loc);
    const double *r = mat + (size_t)i * n_top;
    double acc = 0.0;
    if (phase == 0) {
        const int j0 = lower ? 0 : e, j1 = lower ? a : n_top;
        for (int j = j0 + lane_s; j < j1; j += SP) acc = fma(r[j], u[(size_t)rows[j] * ld + t], acc);
    } else {
        const int j0 = lower ? a : i, j1 = lower ? i + 1 : e;
        for (int j = j0 + lane_s; j < j1; j += SP) acc = fma(r[j], scratch[(size_t)(j - a) * n_loc + t], acc);
    }
#pragma unroll
    for (int off = SP / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, SP);
    if (lane_s != 0) return;
    const int k = rows[i];
    if (phase == 0) {
        scratch[(size_t)(i - a) * n_loc + t] = u[(size_t)k * ld + t] - acc;
        return;
    }
    u[(size_t)k * ld + t] = acc;
    if (dst_perm) {
        double *o = out + (size_t)dst_perm[k] * ld;
        o[t] = acc;
        if (t == n_loc - 1)
            for (int tt = n_loc; tt < ld; ++tt) o[tt] = 0.0;
    }
}

}  // namespace

struct stk_lu {
    int32_t n = 0;
    TriDev L, U;                  // all rows, level by level (no dense top)
    int32_t *src_perm = nullptr;  // row of b that row k of L y = Pr b reads
    int32_t *dst_perm = nullptr;  // row of x that receives row k of z = U^-1 y
    // dense top: S and the three pieces of each solve that are not the dense products
    int32_t n_top = 0;
    std::vector<int32_t> top_rows_host;
    int32_t *top_rows = nullptr;
    TriDev L_head, L_top, U_head;  // L_top: rows of S, entries outside S (one level, a product)
    double *L_inv = nullptr, *U_inv = nullptr;  // [n_top][n_top] row-major, set by the caller
    int32_t top_block = 0;                       // rows of a diagonal block (inverted in place)
    double *scratch = nullptr;                   // [top_block][n_loc] between the two launches of a block
    int64_t scratch_doubles = 0;
};

extern "C" int stk_lu_destroy(stk_lu *lu)
{
    if (!lu) return 0;
    release(lu->L), release(lu->U), release(lu->L_head), release(lu->L_top), release(lu->U_head);
    for (void *p : {(void *)lu->src_perm, (void *)lu->dst_perm, (void *)lu->top_rows, (void *)lu->L_inv,
                    (void *)lu->U_inv, (void *)lu->scratch})
        (void)hipFree(p);
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
    auto all = [](int) { return true; };
    if (build(n, L_indptr, L_indices, L_data, true, true, all, all, all, false, false, &lu->L) ||
        build(n, U_indptr, U_indices, U_data, false, false, all, all, all, false, false, &lu->U) ||
        upload(src, &lu->src_perm) || upload(dst, &lu->dst_perm)) {
        stk_lu_destroy(lu);
        return 1;
    }
    // S: the rows from the first narrow level of L on -- if there are enough levels to
    // save, the blocks stay affordable (two n_top^2 arrays), and the rows of S depend in U
    // on rows of S alone (true for the symmetric patterns of SymmetricMode; checked)
    {
        const TriDev &T = lu->L;
        int d0 = 0;
        while (d0 < T.n_levels && T.lvl_rows[d0] >= WIDE) ++d0;
        std::vector<char> in_top(n, 0);
        int n_top = 0;
        for (int i = 0; i < n; ++i)
            if (T.depth[i] >= d0) in_top[i] = 1, ++n_top;
        bool closed = true;
        for (int i = 0; i < n && closed; ++i)
            if (in_top[i])
                for (int e = U_indptr[i]; e < U_indptr[i + 1]; ++e)
                    if (!in_top[U_indices[e]]) closed = false;
        if (T.n_levels - d0 >= 16 && n_top >= 2 && n_top <= 40000 && closed) {
            auto top = [&](int i) { return in_top[i] != 0; };
            auto head = [&](int i) { return in_top[i] == 0; };
            if (build(n, L_indptr, L_indices, L_data, true, true, head, all, all, false, false, &lu->L_head) ||
                build(n, L_indptr, L_indices, L_data, true, true, top, head, head, true, true, &lu->L_top) ||
                build(n, U_indptr, U_indices, U_data, false, false, head, all, head, false, false, &lu->U_head)) {
                stk_lu_destroy(lu);
                return 1;
            }
            lu->n_top = n_top;
            for (int i = 0; i < n; ++i)
                if (in_top[i]) lu->top_rows_host.push_back(i);
            if (upload(lu->top_rows_host, &lu->top_rows)) {
                stk_lu_destroy(lu);
                return 1;
            }
        }
    }
    *out = lu;
    return 0;
}

extern "C" int stk_lu_top_rows(const stk_lu *lu, int32_t *n_top, int32_t *rows_host)
{
    STK_REQUIRE(lu && n_top, "stk_lu_top_rows: null argument");
    *n_top = lu->n_top;
    if (rows_host)
        for (int i = 0; i < lu->n_top; ++i) rows_host[i] = lu->top_rows_host[i];
    return 0;
}

extern "C" int stk_lu_set_top_inverse(stk_lu *lu, const double *L_inv_dev, const double *U_inv_dev, int32_t block)
{
    STK_REQUIRE(lu && lu->n_top > 0 && L_inv_dev && U_inv_dev, "stk_lu_set_top_inverse: the plan has no dense top");
    STK_REQUIRE(block >= 1 && block <= 8192, "stk_lu_set_top_inverse: blocks of %d rows (1 .. 8192)", block);
    lu->top_block = block < lu->n_top ? block : lu->n_top;
    const size_t bytes = sizeof(double) * (size_t)lu->n_top * lu->n_top;
    if (!lu->L_inv) STK_HIP(hipMalloc((void **)&lu->L_inv, bytes));
    if (!lu->U_inv) STK_HIP(hipMalloc((void **)&lu->U_inv, bytes));
    STK_HIP(hipMemcpy(lu->L_inv, L_inv_dev, bytes, hipMemcpyDeviceToDevice));
    STK_HIP(hipMemcpy(lu->U_inv, U_inv_dev, bytes, hipMemcpyDeviceToDevice));
    return 0;
}

extern "C" int stk_lu_info(const stk_lu *lu, int32_t *levels_L, int32_t *levels_U, int32_t *launches)
{
    STK_REQUIRE(lu, "stk_lu_info: null plan");
    if (levels_L) *levels_L = lu->L.n_levels;
    if (levels_U) *levels_U = lu->U.n_levels;
    if (launches) {
        if (lu->L_inv && lu->U_inv)
            *launches = (int32_t)(lu->L_head.segments.size() + lu->L_top.segments.size() +
                                  lu->U_head.segments.size() +
                                  4 * ((lu->n_top + lu->top_block - 1) / lu->top_block));
        else
            *launches = (int32_t)(lu->L.segments.size() + lu->U.segments.size());
    }
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
    if (!(lu->L_inv && lu->U_inv)) {
        // L y = Pr b into work; U z = y in place on work, every row also to its place in x
        int rc = solve(st, lu->L, n_loc, ld, lu->src_perm, b, work, nullptr, nullptr);
        if (rc) return rc;
        return solve(st, lu->U, n_loc, ld, nullptr, work, work, lu->dst_perm, x);
    }
    const int nb = lu->top_block, n_blocks = (lu->n_top + nb - 1) / nb;
    const int64_t need = (int64_t)nb * n_loc;
    if (lu->scratch_doubles < need) {
        if (lu->scratch) STK_HIP(hipFree(lu->scratch));
        lu->scratch = nullptr, lu->scratch_doubles = 0;
        STK_HIP(hipMalloc((void **)&lu->scratch, sizeof(double) * (size_t)need));
        lu->scratch_doubles = need;
    }
    auto block = [&](int k, int lower, const double *mat, const int32_t *dst, double *out) {
        const int a = k * nb, e = std::min((k + 1) * nb, (int)lu->n_top);
        const unsigned grid = (unsigned)(((int64_t)(e - a) * n_loc * SP + 255) / 256);
        for (int phase = 0; phase < 2; ++phase)
            hipLaunchKernelGGL(block_top_kernel, dim3(grid), dim3(256), 0, st, lu->n_top, lu->top_rows, mat, a, e, lower,
                               phase, n_loc, ld, work, lu->scratch, phase ? dst : nullptr, out);
    };
    // forward
    int rc = solve(st, lu->L_head, n_loc, ld, lu->src_perm, b, work, nullptr, nullptr);
    if (rc == 0) rc = solve(st, lu->L_top, n_loc, ld, lu->src_perm, b, work, nullptr, nullptr);  // d_S into work
    if (rc) return rc;
    for (int k = 0; k < n_blocks; ++k) block(k, 1, lu->L_inv, nullptr, nullptr);
    // backward
    for (int k = n_blocks - 1; k >= 0; --k) block(k, 0, lu->U_inv, lu->dst_perm, x);
    STK_LAUNCH_CHECK();
    return solve(st, lu->U_head, n_loc, ld, nullptr, work, work, lu->dst_perm, x);
}
