// Multigrid plan construction behind the C ABI: from the finest-level CSR
// matrix (optionally a second one, for families a(t) = ca*A + cm[t]*M) and the
// prolongation matrices of the hierarchy -- what a caller of the reference holds
// when it constructs MultiGrid(mat, hierarchy) (reference multigrid.py:130-166)
// -- to everything stk_mg_apply runs on: Galerkin products R A P
// (multigrid.py:142-145), the dependency schedule of the Gauss-Seidel sweeps,
// the sliced-ELL copies of every level (level matrix in locality order, the same
// rows group by group for the sweeps, transfers, the fused restricted-residual
// products R*A, the zero-start matrices of the first sweep), the bands of the
// strip-wise smoothing, and the exact inverse of the coarsest matrices.  Host
// arithmetic in C++, device uploads with hipMemcpy; no Python.  The Python
// planner (source/multigrid.py) builds the same plan with NumPy / SciPy; tests
// compare V-cycles from the two.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <future>
#include <map>
#include <mutex>
#include <numeric>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "stk_common.h"

int g_mg_band_merge = 0;  // tuning key "mg_band_merge": mesh rows per band of the strip-wise sweeps (0 = default)

namespace {

struct Csr {
    int rows = 0, cols = 0;
    std::vector<int32_t> ptr, idx;
    std::vector<double> val;
    int64_t nnz() const { return (int64_t)idx.size(); }
};

Csr from_host(const stk_csr_host &h)
{
    Csr m;
    m.rows = h.n_rows;
    m.cols = h.n_cols;
    m.ptr.assign(h.indptr, h.indptr + h.n_rows + 1);
    m.idx.assign(h.indices, h.indices + m.ptr.back());
    m.val.assign(h.data, h.data + m.ptr.back());
    return m;
}

void sort_rows(Csr &m)
{
    std::vector<std::pair<int32_t, double>> tmp;
    for (int i = 0; i < m.rows; ++i) {
        tmp.clear();
        for (int e = m.ptr[i]; e < m.ptr[i + 1]; ++e) tmp.emplace_back(m.idx[e], m.val[e]);
        std::sort(tmp.begin(), tmp.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        for (int e = m.ptr[i], k = 0; e < m.ptr[i + 1]; ++e, ++k) {
            m.idx[e] = tmp[k].first;
            m.val[e] = tmp[k].second;
        }
    }
}

Csr transpose(const Csr &a)
{
    Csr t;
    t.rows = a.cols;
    t.cols = a.rows;
    t.ptr.assign(t.rows + 1, 0);
    for (int32_t c : a.idx) ++t.ptr[c + 1];
    for (int i = 0; i < t.rows; ++i) t.ptr[i + 1] += t.ptr[i];
    t.idx.resize(a.idx.size());
    t.val.resize(a.idx.size());
    std::vector<int32_t> at(t.ptr.begin(), t.ptr.end() - 1);
    for (int i = 0; i < a.rows; ++i)
        for (int e = a.ptr[i]; e < a.ptr[i + 1]; ++e) {
            const int p = at[a.idx[e]]++;
            t.idx[p] = i;
            t.val[p] = a.val[e];
        }
    return t;  // rows come out sorted
}

// C = A B exactly as SciPy's csr_matmat forms it: each entry accumulated in the
// order of A's row entries and then B's row entries, exact zeros not emitted, and
// -- `emit_sorted` false -- a row's entries emitted in REVERSE order of first touch
// (csr_matmat's linked list), which is the order the next product of a chain
// traverses them in.  (R A) P formed this way is bit for bit SciPy's `R @ A @ P`,
// i.e. the reference's Galerkin matrix (multigrid.py:142-145), the Python
// planner's, and stk_csr_galerkin's.
// Rows [r0, r1) of the product; rows are independent, so the blocks of several
// threads concatenate to the product formed by one.
static void matmul_rows(const Csr &a, const Csr &b, bool emit_sorted, int r0, int r1, std::vector<int32_t> &counts,
                        std::vector<int32_t> &idx, std::vector<double> &val)
{
    std::vector<double> acc(b.cols, 0.0);
    std::vector<int32_t> mark(b.cols, -1), list;
    counts.assign(r1 - r0, 0);
    for (int i = r0; i < r1; ++i) {
        list.clear();
        for (int e = a.ptr[i]; e < a.ptr[i + 1]; ++e) {
            const int k = a.idx[e];
            const double v = a.val[e];
            for (int f = b.ptr[k]; f < b.ptr[k + 1]; ++f) {
                const int j = b.idx[f];
                if (mark[j] != i) {
                    mark[j] = i;
                    acc[j] = 0.0;
                    list.push_back(j);
                }
                const double prod = v * b.val[f];  // rounded on its own, as the SciPy build does
                acc[j] = acc[j] + prod;
            }
        }
        if (emit_sorted)
            std::sort(list.begin(), list.end());
        else
            std::reverse(list.begin(), list.end());
        const size_t before = idx.size();
        for (int j : list) {
            if (acc[j] == 0.0) continue;
            idx.push_back(j);
            val.push_back(acc[j]);
        }
        counts[i - r0] = (int32_t)(idx.size() - before);
    }
}

int host_threads()
{
    int T = (int)std::thread::hardware_concurrency();
    if (const char *env = getenv("STK_HOST_THREADS")) T = atoi(env);
    return std::max(1, std::min(T, 32));
}

Csr matmul(const Csr &a, const Csr &b, bool emit_sorted = true)
{
    Csr c;
    c.rows = a.rows;
    c.cols = b.cols;
    c.ptr.assign(a.rows + 1, 0);
    const int T = a.rows < 8192 ? 1 : host_threads();
    // blocks of rows with equal shares of a's entries
    std::vector<int> cut(T + 1, a.rows);
    cut[0] = 0;
    for (int k = 1; k < T; ++k)
        cut[k] = (int)(std::lower_bound(a.ptr.begin(), a.ptr.end(), (int32_t)((int64_t)a.ptr[a.rows] * k / T)) - a.ptr.begin());
    for (int k = 1; k <= T; ++k) cut[k] = std::min(a.rows, std::max(cut[k], cut[k - 1]));
    std::vector<std::vector<int32_t>> counts(T), idx(T);
    std::vector<std::vector<double>> val(T);
    {
        std::vector<std::thread> pool;
        for (int k = 1; k < T; ++k)
            pool.emplace_back([&, k] { matmul_rows(a, b, emit_sorted, cut[k], cut[k + 1], counts[k], idx[k], val[k]); });
        matmul_rows(a, b, emit_sorted, cut[0], cut[1], counts[0], idx[0], val[0]);
        for (auto &th : pool) th.join();
    }
    size_t total = 0;
    for (int k = 0; k < T; ++k) total += idx[k].size();
    c.idx.reserve(total);
    c.val.reserve(total);
    for (int k = 0; k < T; ++k) {
        for (int i = cut[k]; i < cut[k + 1]; ++i) c.ptr[i + 1] = counts[k][i - cut[k]];
        c.idx.insert(c.idx.end(), idx[k].begin(), idx[k].end());
        c.val.insert(c.val.end(), val[k].begin(), val[k].end());
    }
    for (int i = 0; i < a.rows; ++i) c.ptr[i + 1] += c.ptr[i];
    return c;
}

// entries that are pure rounding noise of a Galerkin product (|a| < rel * max|a|)
// are removed, as source/multigrid.py does (_drop_roundoff)
void drop_roundoff(Csr &m, double rel = 1e-14)
{
    double mx = 0.0;
    for (double v : m.val) mx = std::max(mx, std::fabs(v));
    Csr o;
    o.rows = m.rows;
    o.cols = m.cols;
    o.ptr.assign(m.rows + 1, 0);
    for (int i = 0; i < m.rows; ++i) {
        for (int e = m.ptr[i]; e < m.ptr[i + 1]; ++e)
            if (!(std::fabs(m.val[e]) < rel * mx) && m.val[e] != 0.0) {
                o.idx.push_back(m.idx[e]);
                o.val.push_back(m.val[e]);
            }
        o.ptr[i + 1] = (int32_t)o.idx.size();
    }
    m = std::move(o);
}

// union pattern of one or two matrices with the values of each on it
struct Union {
    int n = 0, cols = 0;
    std::vector<int32_t> ptr, idx;
    std::vector<double> va, vm;  // vm empty if one matrix
};

Union make_union(const Csr &a, const Csr *m)
{
    Union u;
    u.n = a.rows;
    u.cols = a.cols;
    u.ptr.assign(a.rows + 1, 0);
    std::vector<int32_t> cols;
    for (int i = 0; i < a.rows; ++i) {
        cols.assign(a.idx.begin() + a.ptr[i], a.idx.begin() + a.ptr[i + 1]);
        if (m) cols.insert(cols.end(), m->idx.begin() + m->ptr[i], m->idx.begin() + m->ptr[i + 1]);
        std::sort(cols.begin(), cols.end());
        cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
        const size_t base = u.idx.size();
        u.idx.insert(u.idx.end(), cols.begin(), cols.end());
        u.va.resize(u.idx.size(), 0.0);
        if (m) u.vm.resize(u.idx.size(), 0.0);
        for (int e = a.ptr[i]; e < a.ptr[i + 1]; ++e)
            u.va[base + (std::lower_bound(cols.begin(), cols.end(), a.idx[e]) - cols.begin())] += a.val[e];
        if (m)
            for (int e = m->ptr[i]; e < m->ptr[i + 1]; ++e)
                u.vm[base + (std::lower_bound(cols.begin(), cols.end(), m->idx[e]) - cols.begin())] += m->val[e];
        u.ptr[i + 1] = (int32_t)u.idx.size();
    }
    return u;
}

// ---- orders ------------------------------------------------------------------
// mesh-tile order of the first n points (tiles of about rows_per_tile points,
// visited lexicographically, points lexicographic inside): source/assembly.py
// tile_order_from_coords
std::vector<int32_t> tile_order(const double *coords, int dim, int n, int rows_per_tile = 2048)
{
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    if (!coords) return order;
    std::vector<double> lo(dim, 1e300), hi(dim, -1e300);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < dim; ++k) {
            lo[k] = std::min(lo[k], coords[(size_t)i * dim + k]);
            hi[k] = std::max(hi[k], coords[(size_t)i * dim + k]);
        }
    double vol = 1.0;
    for (int k = 0; k < dim; ++k) vol *= std::max(hi[k] - lo[k], 1e-30);
    const bool tiled = n > rows_per_tile;
    const double side = std::pow(vol / std::max(1.0, n / (double)rows_per_tile), 1.0 / dim);
    // sort keys formed once: the tile (slowest axis first), then the coordinates
    struct Key {
        int64_t tile[3];
        double c[3];
        int32_t id;
    };
    std::vector<Key> keys(n);
    for (int i = 0; i < n; ++i) {
        Key &q = keys[i];
        q.id = i;
        for (int k = 0; k < 3; ++k) q.tile[k] = 0, q.c[k] = 0.0;
        for (int k = 0; k < dim; ++k) {
            const double v = coords[(size_t)i * dim + (dim - 1 - k)];  // slowest axis first
            q.c[k] = v;
            q.tile[k] = tiled ? (int64_t)std::floor((v - lo[dim - 1 - k]) / side) : (int64_t)0;
        }
    }
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
        for (int k = 0; k < 3; ++k)
            if (a.tile[k] != b.tile[k]) return a.tile[k] < b.tile[k];
        for (int k = 0; k < 3; ++k)
            if (a.c[k] != b.c[k]) return a.c[k] < b.c[k];
        return a.id < b.id;
    });
    for (int i = 0; i < n; ++i) order[i] = keys[i].id;
    return order;
}

// bands with "coupled rows at most one band apart", verified on the pattern:
// slices along the last axis when coordinates are known (halved until the
// property holds), breadth-first levels otherwise (source/multigrid.py
// coupling_bands).  Empty result: no useful banding.
std::vector<int32_t> coupling_bands(const double *coords, int dim, const Union &u)
{
    const int n = u.n;
    std::vector<int32_t> band(n, 0);
    if (n < 2) return {};
    auto ok = [&](const std::vector<int32_t> &b) {
        for (int i = 0; i < n; ++i)
            for (int e = u.ptr[i]; e < u.ptr[i + 1]; ++e)
                if (std::abs(b[i] - b[u.idx[e]]) > 1) return false;
        return true;
    };
    if (coords) {
        std::vector<double> ys(n);
        for (int i = 0; i < n; ++i) ys[i] = std::round(coords[(size_t)i * dim + dim - 1] * 1e12) / 1e12;
        std::vector<double> uniq(ys);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        for (int i = 0; i < n; ++i) band[i] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), ys[i]) - uniq.begin());
        int mx = (int)uniq.size() - 1;
        while (mx > 0 && !ok(band)) {
            for (auto &b : band) b /= 2;
            mx /= 2;
        }
        if (mx < 1 || !ok(band)) return {};
        return band;
    }
    // breadth-first levels from row 0 on the symmetrised pattern
    std::vector<std::vector<int32_t>> adj(n);
    for (int i = 0; i < n; ++i)
        for (int e = u.ptr[i]; e < u.ptr[i + 1]; ++e) {
            adj[i].push_back(u.idx[e]);
            adj[u.idx[e]].push_back(i);
        }
    std::fill(band.begin(), band.end(), -1);
    std::queue<int> q;
    band[0] = 0;
    q.push(0);
    int mx = 0;
    while (!q.empty()) {
        const int i = q.front();
        q.pop();
        for (int j : adj[i])
            if (band[j] < 0) {
                band[j] = band[i] + 1;
                mx = std::max(mx, band[j]);
                q.push(j);
            }
    }
    for (int b : band)
        if (b < 0) return {};
    if (mx < 1 || !ok(band)) return {};
    return band;
}

// ---- the builder -------------------------------------------------------------
struct Builder {
    std::vector<void *> dev;  // every device allocation (adopted by the plan)
    std::atomic<bool> failed{false};
    std::mutex mu;  // the pieces of a level are built by several host threads

    template <typename T>
    const T *up(const std::vector<T> &h)
    {
        if (h.empty()) {  // keep pointers valid
            std::vector<T> one(1, T());
            return up(one);
        }
        T *d = nullptr;
        if (hipMalloc((void **)&d, h.size() * sizeof(T)) != hipSuccess ||
            hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
            failed = true;
            return nullptr;
        }
        {
            std::lock_guard<std::mutex> lock(mu);
            dev.push_back(d);
        }
        return d;
    }
};

static const int ROW_SLOTS[] = {2, 4, 5, 6, 7, 9, 12, 16, 20};

}  // namespace

// Tuning key "mg_gs_diag_free" (read when a plan is built by stk_mg_create_from_csr):
// 1 (default): the Gauss-Seidel copies hold the off-diagonal entries only and a row is
// updated as u_i = (f_i - sum_{j != i} a_ij u_j) / a_ii; 0: the diagonal stays among the
// slots and the update is the reference's u_i += (f_i - row_i u) / a_ii
// (multigrid.py:89-97); 2: the latter on the finest level, which also gets the
// diagonal-free copies as its alternative form (stk_mg_level.ell_fwd_alt / ell_bwd_alt),
// the former below -- with stk_mg_set_option(plan, "fuse_restrict_max_level", finest - 1),
// ("fast_until_cycle", vcycles - 1) and ("fast_parts", 1) the arithmetic of
// HeatEquationMPI(arithmetic='accurate'), DESIGN.md section 5.
int g_mg_gs_diag_free = 1;

namespace {

// sliced-ELL copy of the rows `order` of a union pattern (source/linop.py
// EllRowsMatrix): K = smallest slot count that holds the longest listed row,
// unused slots: column pad_col (pad_col < 0: the row's first listed column, or the
// row itself if it lists none), value 0.  ok = false if a row is too long.
// diag = true (Gauss-Seidel copies): the diagonal entries go to dia_a / dia_m and
// NOT into the slots (stk_ell_rows.diag_free).
bool ell_rows(Builder &B, const Union &u, const std::vector<int32_t> &order, bool diag, int pad_col,
              const std::vector<double> *dia_a, const std::vector<double> *dia_m, stk_ell_rows *out)
{
    int kmax = 1;
    for (int i : order) kmax = std::max(kmax, u.ptr[i + 1] - u.ptr[i] - (diag ? 1 : 0));
    int K = 0;
    for (int s : ROW_SLOTS)
        if (s >= kmax) {
            K = s;
            break;
        }
    if (K == 0) return false;
    const size_t np = order.size();
    const bool has_m = !u.vm.empty();
    std::vector<int32_t> idx(np * K, pad_col < 0 ? 0 : pad_col), rows(np);
    std::vector<double> va(np * K, 0.0), vm(has_m ? np * K : 0, 0.0), da, dm;
    if (diag || dia_a) {
        da.resize(np);
        if (has_m) dm.resize(np);
    }
    for (size_t p = 0; p < np; ++p) {
        const int i = order[p];
        rows[p] = i;
        for (int e = u.ptr[i], s = 0; e < u.ptr[i + 1]; ++e) {
            if (diag && u.idx[e] == i) {
                da[p] = u.va[e];
                if (has_m) dm[p] = u.vm[e];
                continue;
            }
            idx[p * K + s] = u.idx[e];
            va[p * K + s] = u.va[e];
            if (has_m) vm[p * K + s] = u.vm[e];
            ++s;
        }
        if (dia_a) {
            da[p] = (*dia_a)[i];
            if (has_m && dia_m) dm[p] = (*dia_m)[i];
        }
        if (pad_col < 0) {
            int s = u.ptr[i + 1] - u.ptr[i] - (diag ? 1 : 0);
            const int32_t pad = s > 0 ? idx[p * K] : i;
            for (; s < K; ++s) idx[p * K + s] = pad;
        }
    }
    out->n_pos = (int32_t)np;
    out->n_rows = u.n;
    out->K = K;
    out->idx = B.up(idx);
    out->va = B.up(va);
    out->vm = has_m ? B.up(vm) : nullptr;
    out->row_ids = B.up(rows);
    out->dia_a = (diag || dia_a) ? B.up(da) : nullptr;
    out->dia_m = ((diag || dia_a) && has_m) ? B.up(dm) : nullptr;
    out->diag_free = diag ? 1 : 0;
    return true;
}

Union union_of(const Csr &m) { return make_union(m, nullptr); }

// The caller's own dense inverse (stk_mg_set_coarse_inverse), or none; the mesh rows per
// band (tuning key "mg_band_merge").  Both are process-wide settings that a plan
// construction reads ONCE, under the mutex, when it starts (PlannerSettings): a setter
// running on another thread while plans are being built (this repository builds K's and
// the family's plan side by side) cannot change a construction half way.
stk_dense_inverse_fn g_inverse_fn = nullptr;
void *g_inverse_user = nullptr;
std::mutex g_planner_mutex;

struct PlannerSettings {
    stk_dense_inverse_fn inverse_fn;
    void *inverse_user;
    int band_merge;
};

PlannerSettings planner_settings()
{
    std::lock_guard<std::mutex> lock(g_planner_mutex);
    return PlannerSettings{g_inverse_fn, g_inverse_user, g_mg_band_merge};
}

// dense inverse by Gauss-Jordan elimination with partial pivoting (level 0 is tiny)
bool dense_inverse(std::vector<double> a, int n, double *out, const PlannerSettings &set)
{
    if (set.inverse_fn != nullptr) return set.inverse_fn(n, a.data(), out, set.inverse_user) == 0;
    std::vector<double> inv((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(a[(size_t)r * n + c]) > std::fabs(a[(size_t)piv * n + c])) piv = r;
        if (a[(size_t)piv * n + c] == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < n; ++k) {
                std::swap(a[(size_t)piv * n + k], a[(size_t)c * n + k]);
                std::swap(inv[(size_t)piv * n + k], inv[(size_t)c * n + k]);
            }
        const double d = 1.0 / a[(size_t)c * n + c];
        for (int k = 0; k < n; ++k) {
            a[(size_t)c * n + k] *= d;
            inv[(size_t)c * n + k] *= d;
        }
        for (int r = 0; r < n; ++r)
            if (r != c) {
                const double f = a[(size_t)r * n + c];
                if (f != 0.0)
                    for (int k = 0; k < n; ++k) {
                        a[(size_t)r * n + k] -= f * a[(size_t)c * n + k];
                        inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k];
                    }
            }
    }
    std::memcpy(out, inv.data(), sizeof(double) * (size_t)n * n);
    return true;
}

}  // namespace

extern "C" int stk_mg_set_coarse_inverse(stk_dense_inverse_fn fn, void *user)
{
    std::lock_guard<std::mutex> lock(g_planner_mutex);
    g_inverse_fn = fn;
    g_inverse_user = user;
    return 0;
}

extern "C" int stk_mg_create_from_csr(int32_t n_levels, const stk_csr_host *A_fine, const stk_csr_host *M_fine,
                                      const stk_csr_host *P_host, const double *coords_host, int32_t dim,
                                      int32_t smoothsteps, int32_t vcycles, double ca, int32_t n_cms,
                                      const double *cms_host, int32_t max_ld, stk_mg **out)
{
    STK_REQUIRE(n_levels >= 1 && A_fine && out && (n_levels == 1 || P_host), "stk_mg_create_from_csr: bad arguments");
    STK_REQUIRE(A_fine->n_rows == A_fine->n_cols && A_fine->indptr && A_fine->indices && A_fine->data,
                "stk_mg_create_from_csr: the fine matrix must be square CSR");
    STK_REQUIRE(!M_fine || (M_fine->n_rows == A_fine->n_rows && M_fine->n_cols == A_fine->n_cols),
                "stk_mg_create_from_csr: second matrix has another shape");
    STK_REQUIRE((n_cms == 0) == (M_fine == nullptr) || n_cms == 0,
                "stk_mg_create_from_csr: coefficients cms need the second matrix");
    STK_REQUIRE(!coords_host || (dim >= 1 && dim <= 3), "stk_mg_create_from_csr: dim=%d not in 1..3", dim);
    const int J = n_levels - 1;
    const PlannerSettings settings = planner_settings();  // one consistent view for this construction
    // STK_PLAN_TIMING=1: seconds per stage of the planner on stderr
    static const bool timing = getenv("STK_PLAN_TIMING") != nullptr;
    std::map<std::string, double> spent;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_last = now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto t = now();
        spent[what] += std::chrono::duration<double>(t - t_last).count();
        t_last = t;
    };
    // ---- Galerkin hierarchies, coarse to fine (multigrid.py:142-145) -------------
    std::vector<Csr> A(n_levels), Mm(M_fine ? n_levels : 0), P(J), R(J);
    A[J] = from_host(*A_fine);
    sort_rows(A[J]);
    if (M_fine) {
        Mm[J] = from_host(*M_fine);
        sort_rows(Mm[J]);
    }
    for (int j = J - 1; j >= 0; --j) {
        const int fine_rows = j + 1 == J ? A_fine->n_rows : P_host[j + 1].n_cols;
        STK_REQUIRE(P_host[j].n_rows == fine_rows, "stk_mg_create_from_csr: P[%d] has %d rows, level %d has %d", j,
                    P_host[j].n_rows, j + 1, fine_rows);
        P[j] = from_host(P_host[j]);
        sort_rows(P[j]);
        R[j] = transpose(P[j]);
    }
    lap("copy in, sort rows");
    // The chain of Galerkin products runs beside the plan of the finest level, which
    // needs nothing of it.  Worker threads work on the caller's device.
    int device = 0;
    (void)hipGetDevice(&device);
    std::thread chain([&] {
        for (int j = J - 1; j >= 0; --j) {
            A[j] = matmul(matmul(R[j], A[j + 1], /*emit_sorted=*/false), P[j]);
            drop_roundoff(A[j]);
            if (M_fine) {
                Mm[j] = matmul(matmul(R[j], Mm[j + 1], false), P[j]);
                drop_roundoff(Mm[j]);
            }
        }
    });
    struct Joiner {  // no return path leaves the thread running
        std::thread &t;
        ~Joiner()
        {
            if (t.joinable()) t.join();
        }
    } chain_guard{chain};
    Builder B;
    std::vector<stk_mg_level> lv(n_levels);
    std::memset(lv.data(), 0, sizeof(stk_mg_level) * n_levels);
    // host arrays the level structs point into while stk_mg_create copies them
    struct Keep {
        std::vector<int32_t> fwd_ptr, bwd_ptr, fwd_trow, bwd_trow;
        stk_ell_rows a, fwd, bwd, p, r, ra, fwd_alt, bwd_alt;
        std::vector<stk_ell_rows> fwd0;
    };
    std::vector<Keep> keep(n_levels);
    // processing order of a level = of the first n dofs (levels are nested); formed once per size
    std::map<int, std::vector<int32_t>> tile_orders;
    auto tile_of = [&](int n) -> const std::vector<int32_t> & {
        auto it = tile_orders.find(n);
        if (it == tile_orders.end()) it = tile_orders.emplace(n, tile_order(coords_host, dim, n)).first;
        return it->second;
    };
    auto build_level = [&](int j) -> int {
        const Union u = make_union(A[j], M_fine ? &Mm[j] : nullptr);
        const int n = u.n;
        stk_mg_level &L = lv[j];
        Keep &K = keep[j];
        L.n = n;
        std::vector<int32_t> diag(n, -1);
        for (int i = 0; i < n; ++i)
            for (int e = u.ptr[i]; e < u.ptr[i + 1]; ++e)
                if (u.idx[e] == i) diag[i] = e;
        for (int i = 0; i < n; ++i)
            STK_REQUIRE(diag[i] >= 0, "stk_mg_create_from_csr: level %d row %d lacks a diagonal entry", j, i);
        L.indptr = B.up(u.ptr);
        L.indices = B.up(u.idx);
        L.vals_a = B.up(u.va);
        L.vals_m = M_fine ? B.up(u.vm) : nullptr;
        L.diag = B.up(diag);
        std::vector<double> dia_a_of_row(n), dia_m_of_row(M_fine ? n : 0);
        for (int i = 0; i < n; ++i) {
            dia_a_of_row[i] = u.va[diag[i]];
            if (M_fine) dia_m_of_row[i] = u.vm[diag[i]];
        }
        lap("union pattern, CSR upload");
        if (j == 0) return 0;
        // processing order (mesh tiles if coordinates are known) and bands
        const std::vector<int32_t> &tile = tile_of(n);
        const std::vector<int32_t> &tile_c = tile_of(P[j - 1].cols);
        // the transfer operators and the products R A need neither the union pattern
        // nor the schedules: two tasks beside the rest of the level
        std::future<bool> transfers = std::async(std::launch::async, [&, j]() -> bool {
            (void)hipSetDevice(device);
            const Union up_ = union_of(P[j - 1]), ur = union_of(R[j - 1]);
            L.p_indptr = B.up(up_.ptr);
            L.p_indices = B.up(up_.idx);
            L.p_vals = B.up(up_.va);
            L.r_indptr = B.up(ur.ptr);
            L.r_indices = B.up(ur.idx);
            L.r_vals = B.up(ur.va);
            bool ok = ell_rows(B, up_, tile, false, 0, nullptr, nullptr, &K.p);
            return ell_rows(B, ur, tile_c, false, 0, nullptr, nullptr, &K.r) && ok;
        });
        // restricted residual in one step: d = (R A) u - R f
        std::future<bool> products = std::async(std::launch::async, [&, j]() -> bool {
            (void)hipSetDevice(device);
            Csr ra = matmul(R[j - 1], A[j]);
            drop_roundoff(ra);
            Csr rm;
            if (M_fine) {
                rm = matmul(R[j - 1], Mm[j]);
                drop_roundoff(rm);
            }
            const Union ura = make_union(ra, M_fine ? &rm : nullptr);
            return ell_rows(B, ura, tile_c, false, 0, nullptr, nullptr, &K.ra);
        });
        std::vector<int64_t> rank(n);
        for (int p = 0; p < n; ++p) rank[tile[p]] = p;
        lap("tile order");
        std::vector<int32_t> band = coupling_bands(coords_host, dim, u);
        // several mesh rows per band for the strip-wise sweeps (tuning key "mg_band_merge" as
        // it stands when the plan is built; 0 = the default: 6 for families a(t) = ca A +
        // cm[t] M, 1 otherwise -- source/multigrid.py BAND_MERGE has the measurements):
        // coarser bands keep "coupled rows at most one band apart", results do not change
        if (coords_host && !band.empty()) {
            const int merge = settings.band_merge > 0 ? settings.band_merge : (M_fine ? 6 : 1);
            if (merge > 1)
                for (auto &b : band) b /= merge;
        }
        lap("bands");
        auto key = [&](int i) { return (band.empty() ? (int64_t)0 : (int64_t)band[i] * n) + rank[i]; };
        bool ells_ok = ell_rows(B, u, tile, false, 0, nullptr, nullptr, &K.a);
        lap("ELL copy of the level matrix");
        std::vector<std::vector<int32_t>> fwd_groups;
        for (int bw = 0; bw < 2; ++bw) {
            // depth in the dependency DAG of the sweep in dof order (row i waits for
            // its neighbours j < i, or j > i backwards); one pass suffices
            std::vector<int32_t> depth(n, 0);
            int dmax = 0;
            for (int s = 0; s < n; ++s) {
                const int i = bw ? n - 1 - s : s;
                int d = 0;
                for (int e = u.ptr[i]; e < u.ptr[i + 1]; ++e) {
                    const int c = u.idx[e];
                    if (bw ? c > i : c < i) d = std::max(d, depth[c] + 1);
                }
                depth[i] = d;
                dmax = std::max(dmax, d);
            }
            const int ng = dmax + 1;
            std::vector<std::vector<int32_t>> groups(ng);
            // CSR fallback lists: ascending (descending for the backward sweep) row index
            for (int s = 0; s < n; ++s) {
                const int i = bw ? n - 1 - s : s;
                groups[depth[i]].push_back(i);
            }
            std::vector<int32_t> ptr(1, 0), rows;
            for (auto &g : groups) {
                rows.insert(rows.end(), g.begin(), g.end());
                ptr.push_back((int32_t)rows.size());
            }
            lap("sweep schedules");
            // ELL copy: each group band by band, tile order inside a band
            std::vector<int32_t> listed, trow;
            for (auto &g : groups) {
                std::sort(g.begin(), g.end(), [&](int a, int b) { return key(a) < key(b); });
                listed.insert(listed.end(), g.begin(), g.end());
            }
            if (!band.empty())
                for (int i : listed) trow.push_back(band[i]);
            lap("sweep orders");
            stk_ell_rows *dst = bw ? &K.bwd : &K.fwd;
            // row form of this level's sweeps: diagonal-free (key 1, and the levels below
            // the finest with key 2) or with the diagonal (key 0; the finest level with
            // key 2, which also gets the diagonal-free copies as its alternative form)
            const bool full_rows = g_mg_gs_diag_free == 0 || (g_mg_gs_diag_free == 2 && j == J);
            if (!full_rows) {
                ells_ok = ell_rows(B, u, listed, true, 0, nullptr, nullptr, dst) && ells_ok;
            } else {
                ells_ok = ell_rows(B, u, listed, false, 0, &dia_a_of_row, M_fine ? &dia_m_of_row : nullptr, dst) && ells_ok;
                if (g_mg_gs_diag_free == 2) {
                    stk_ell_rows *alt = bw ? &K.bwd_alt : &K.fwd_alt;
                    if (ell_rows(B, u, listed, true, 0, nullptr, nullptr, alt)) (bw ? L.ell_bwd_alt : L.ell_fwd_alt) = alt;
                }
            }
            lap("ELL copies of the sweeps");
            if (bw) {
                K.bwd_ptr = ptr;
                K.bwd_trow = trow;
                L.n_bwd = ng;
                L.bwd_rows = B.up(rows);
            } else {
                K.fwd_ptr = ptr;
                K.fwd_trow = trow;
                L.n_fwd = ng;
                L.fwd_rows = B.up(rows);
                fwd_groups = groups;
            }
        }
        L.fwd_ptr_host = K.fwd_ptr.data();
        L.bwd_ptr_host = K.bwd_ptr.data();
        ells_ok = transfers.get() && ells_ok;
        const bool ra_ok = products.get();
        lap("waiting for transfers and R A products");
        if (!ells_ok) return 0;
        L.ell_a = &K.a;
        L.ell_fwd = &K.fwd;
        L.ell_bwd = &K.bwd;
        L.ell_p = &K.p;
        L.ell_r = &K.r;
        L.fwd_pos_host = K.fwd_ptr.data();
        L.bwd_pos_host = K.bwd_ptr.data();
        if (!band.empty()) {
            L.n_tile_rows = *std::max_element(band.begin(), band.end()) + 1;
            L.fwd_tile_row_host = K.fwd_trow.data();
            L.bwd_tile_row_host = K.bwd_trow.data();
        }
        if (ra_ok) L.ell_ra = &K.ra;
        // first sweep from u = 0: per forward group, only the entries towards earlier groups
        {
            bool nonempty = !fwd_groups.empty();
            for (auto &g : fwd_groups) nonempty = nonempty && !g.empty();
            if (nonempty) {
                std::vector<int32_t> grp(n);
                for (size_t g = 0; g < fwd_groups.size(); ++g)
                    for (int i : fwd_groups[g]) grp[i] = (int32_t)g;
                Union f;
                f.n = n;
                f.cols = n;
                f.ptr.assign(n + 1, 0);
                for (int i = 0; i < n; ++i) {
                    for (int e = u.ptr[i]; e < u.ptr[i + 1]; ++e)
                        if (grp[u.idx[e]] < grp[i]) {
                            f.idx.push_back(u.idx[e]);
                            f.va.push_back(u.va[e]);
                            if (M_fine) f.vm.push_back(u.vm[e]);
                        }
                    f.ptr[i + 1] = (int32_t)f.idx.size();
                }
                const std::vector<double> &da = dia_a_of_row, &dm = dia_m_of_row;
                // unused slots repeat the row's first kept column (a row of an earlier
                // group, written before this row in every order of the sweeps)
                const int safe = -1;
                K.fwd0.resize(fwd_groups.size());
                bool ok0 = true;
                // (has_m must follow the plan, also for an empty value array)
                if (M_fine && f.vm.size() != f.idx.size()) f.vm.assign(f.idx.size(), 0.0);
                for (size_t g = 0; g < fwd_groups.size(); ++g) {
                    ok0 = ell_rows(B, f, fwd_groups[g], false, safe, &da, M_fine ? &dm : nullptr, &K.fwd0[g]) && ok0;
                    if (M_fine && K.fwd0[g].vm == nullptr) {  // group without entries: still needs a vm array
                        std::vector<double> z((size_t)K.fwd0[g].n_pos * K.fwd0[g].K, 0.0);
                        K.fwd0[g].vm = B.up(z);
                    }
                }
                if (ok0) L.ell_fwd0 = K.fwd0.data();
            }
        }
        lap("zero-start copies");
        return 0;
    };
    {
        int rc = build_level(J);
        chain.join();
        lap("waiting for the Galerkin products");
        for (int j = J - 1; j >= 0 && rc == 0; --j) rc = build_level(j);
        if (rc) {
            for (void *d : B.dev) (void)hipFree(d);
            return rc;
        }
    }
    // ---- exact coarse inverses: kind 0 = A_0 alone, kind 1 + k = ca*A_0 + cms[k]*M_0 --
    const int n0 = A[0].rows;
    const int n_kinds = 1 + (M_fine ? n_cms : 0);
    std::vector<double> inv((size_t)n_kinds * n0 * n0);
    {
        std::vector<double> a0((size_t)n0 * n0, 0.0), m0((size_t)n0 * n0, 0.0);
        for (int i = 0; i < n0; ++i) {
            for (int e = A[0].ptr[i]; e < A[0].ptr[i + 1]; ++e) a0[(size_t)i * n0 + A[0].idx[e]] += A[0].val[e];
            if (M_fine)
                for (int e = Mm[0].ptr[i]; e < Mm[0].ptr[i + 1]; ++e) m0[(size_t)i * n0 + Mm[0].idx[e]] += Mm[0].val[e];
        }
        STK_REQUIRE(dense_inverse(a0, n0, inv.data(), settings), "stk_mg_create_from_csr: coarsest matrix is singular");
        for (int k = 0; k < n_kinds - 1; ++k) {
            std::vector<double> c((size_t)n0 * n0);
            for (size_t q = 0; q < c.size(); ++q) c[q] = ca * a0[q] + cms_host[k] * m0[q];
            STK_REQUIRE(dense_inverse(c, n0, inv.data() + (size_t)(k + 1) * n0 * n0, settings),
                        "stk_mg_create_from_csr: coarsest matrix of kind %d is singular", k + 1);
        }
    }
    const double *d_inv = B.up(inv);
    if (B.failed) {
        for (void *d : B.dev) (void)hipFree(d);
        stk_set_error("stk_mg_create_from_csr: device allocation or copy failed");
        return 1;
    }
    stk_mg *mg = nullptr;
    const int rc = stk_mg_create(n_levels, lv.data(), smoothsteps, vcycles, n_kinds, d_inv, max_ld, &mg);
    if (rc) {
        for (void *d : B.dev) (void)hipFree(d);
        return rc;
    }
    stk_mg_adopt(mg, B.dev.data(), (int)B.dev.size());
    *out = mg;
    lap("coarse inverses, stk_mg_create");
    if (timing) {
        double total = 0.0;
        for (auto &kv : spent) total += kv.second;
        fprintf(stderr, "stk_mg_create_from_csr: %.3f s\n", total);
        for (auto &kv : spent) fprintf(stderr, "   %-36s %.3f s\n", kv.first.c_str(), kv.second);
    }
    return 0;
}
