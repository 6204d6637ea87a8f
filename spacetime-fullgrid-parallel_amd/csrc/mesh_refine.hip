// One uniform (red) refinement of a triangulation with hierarchical vertex
// numbering, on the host threads of libstk (problem set-up, SURVEY section 8 row
// f1; the reference gets its meshes from Netgen, source/problem.py:7-41, and the
// parent-vertex table from NGSolve's GetParentVertices, source/multigrid.py:20-21).
// Host code only, integers and one halving per coordinate: the result is the one of
// source/mesh.py:TriangleMesh.refine's NumPy form, entry for entry.
//
//   edges      unique vertex pairs (lo < hi) in ascending (lo, hi) -- a counting
//              sort by `lo`, the handful of entries of a bucket sorted by `hi`;
//   midpoints  0.5 * (p[lo] + p[hi]);
//   numbering  new vertices ordered by (colour of the bisected edge, y, x): a
//              sample sort over the host threads, every key distinct;
//   children   [v0 m2 m1 | v1 m0 m2 | v2 m1 m0 | m0 m1 m2] in four blocks of nt
//              triangles, edge colours carried along as mesh.py does.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "stk_common.h"

#pragma clang fp contract(off)

namespace {

struct EdgeKey {  // sort key of a new vertex
    int64_t col;
    double y, x;
    int64_t e;
};

inline bool before(const EdgeKey &a, const EdgeKey &b)
{
    if (a.col != b.col) return a.col < b.col;
    if (a.y != b.y) return a.y < b.y;
    if (a.x != b.x) return a.x < b.x;
    return a.e < b.e;  // lexsort is stable
}

int host_threads(int64_t items)
{
    int T = (int)std::thread::hardware_concurrency();
    if (const char *env = getenv("STK_HOST_THREADS")) T = atoi(env);
    T = std::max(1, std::min(T, 32));
    if (items < 16384) T = 1;
    return T;
}

template <class F>
void on_threads(int T, F body)
{
    if (T == 1) {
        body(0);
        return;
    }
    std::vector<std::thread> pool;
    for (int k = 0; k < T; ++k) pool.emplace_back(body, k);
    for (auto &th : pool) th.join();
}

inline int64_t share(int64_t n, int T, int k) { return n * k / T; }

// keys[0 .. n) into ascending order: samples pick T - 1 splitters, every thread
// scatters its share into the buckets and then sorts one bucket
template <class Key, class Less>
void sample_sort(std::vector<Key> &keys, int T, Less before)
{
    const int64_t n = (int64_t)keys.size();
    if (T == 1 || n < 65536) {
        std::sort(keys.begin(), keys.end(), before);
        return;
    }
    const int per = 64;
    std::vector<Key> sample;
    sample.reserve((size_t)T * per);
    for (int64_t s = 0; s < (int64_t)T * per; ++s) sample.push_back(keys[(size_t)((2 * s + 1) * n / (2 * (int64_t)T * per))]);
    std::sort(sample.begin(), sample.end(), before);
    std::vector<Key> split;
    for (int k = 1; k < T; ++k) split.push_back(sample[(size_t)k * per]);
    auto bucket_of = [&](const Key &q) {
        return (int)(std::upper_bound(split.begin(), split.end(), q, before) - split.begin());
    };
    std::vector<int32_t> where((size_t)n);
    std::vector<int64_t> count((size_t)T * T, 0);  // [thread][bucket]
    on_threads(T, [&](int k) {
        for (int64_t i = share(n, T, k); i < share(n, T, k + 1); ++i) {
            where[i] = bucket_of(keys[i]);
            ++count[(size_t)k * T + where[i]];
        }
    });
    // bucket b holds thread 0's entries, then thread 1's, ...
    std::vector<int64_t> at((size_t)T * T), begin(T + 1, 0);
    int64_t run = 0;
    for (int b = 0; b < T; ++b) {
        begin[b] = run;
        for (int k = 0; k < T; ++k) {
            at[(size_t)k * T + b] = run;
            run += count[(size_t)k * T + b];
        }
    }
    begin[T] = run;
    std::vector<Key> out((size_t)n);
    on_threads(T, [&](int k) {
        int64_t *mine = &at[(size_t)k * T];
        for (int64_t i = share(n, T, k); i < share(n, T, k + 1); ++i) out[mine[where[i]]++] = keys[i];
    });
    on_threads(T, [&](int b) { std::sort(out.begin() + begin[b], out.begin() + begin[b + 1], before); });
    keys.swap(out);
}

struct TileKey {  // sort key of a dof in the mesh-tile order
    int64_t t[3];
    double c[3];
    int64_t i;
};

}  // namespace

extern "C" int stk_tri_refine(int64_t nv, int64_t nt, const double *points, const int64_t *tris,
                              const int64_t *tri_colors, int64_t capacity, double *new_points,
                              int64_t *new_parents, int64_t *new_colors, int64_t *child_tris,
                              int64_t *child_colors, int64_t *n_new)
{
    STK_REQUIRE(nv > 0 && nt > 0 && points && tris && tri_colors && new_points && new_parents && new_colors &&
                    child_tris && child_colors && n_new,
                "stk_tri_refine: bad arguments");
    STK_REQUIRE(nv < ((int64_t)1 << 31) && nt < ((int64_t)1 << 29), "stk_tri_refine: mesh too large for 32-bit tables");
    for (int64_t q = 0; q < 3 * nt; ++q)
        STK_REQUIRE(tris[q] >= 0 && tris[q] < nv, "stk_tri_refine: triangle %lld names vertex %lld", (long long)(q / 3),
                    (long long)tris[q]);
    const int64_t ns = 3 * nt;  // slot 3 t + k: the edge opposite local vertex k of triangle t
    const int T = host_threads(ns);

    // --- buckets by the smaller end ---
    std::vector<int32_t> lo((size_t)ns), hi((size_t)ns);
    std::vector<std::atomic<int32_t>> fill((size_t)nv + 1);
    on_threads(T, [&](int k) {
        for (int64_t v = share(nv + 1, T, k); v < share(nv + 1, T, k + 1); ++v) fill[v].store(0, std::memory_order_relaxed);
    });
    std::atomic<int> degenerate{0};
    on_threads(T, [&](int k) {
        for (int64_t t = share(nt, T, k); t < share(nt, T, k + 1); ++t)
            for (int e = 0; e < 3; ++e) {
                const int64_t a = tris[3 * t + (e + 1) % 3], b = tris[3 * t + (e + 2) % 3];
                if (a == b) degenerate = 1;
                lo[3 * t + e] = (int32_t)std::min(a, b);
                hi[3 * t + e] = (int32_t)std::max(a, b);
                fill[lo[3 * t + e]].fetch_add(1, std::memory_order_relaxed);
            }
    });
    STK_REQUIRE(!degenerate.load(), "stk_tri_refine: a triangle names a vertex twice");
    std::vector<int64_t> start((size_t)nv + 1);
    {
        int64_t run = 0;
        for (int64_t v = 0; v < nv; ++v) {
            start[v] = run;
            run += fill[v].load(std::memory_order_relaxed);
            fill[v].store(0, std::memory_order_relaxed);
        }
        start[nv] = run;
    }
    // the slots of a bucket in any order (what is kept of them is sorted below)
    std::vector<int32_t> slot((size_t)ns);
    on_threads(T, [&](int k) {
        for (int64_t q = share(ns, T, k); q < share(ns, T, k + 1); ++q)
            slot[start[lo[q]] + fill[lo[q]].fetch_add(1, std::memory_order_relaxed)] = (int32_t)q;
    });
    // vertex ranges with equal shares of the slots
    std::vector<int64_t> cut(T + 1, nv);
    cut[0] = 0;
    for (int k = 1; k < T; ++k) cut[k] = std::lower_bound(start.begin(), start.end(), ns * k / T) - start.begin();
    for (int k = 1; k <= T; ++k) cut[k] = std::max(cut[k], cut[k - 1]);
    // --- every bucket by the larger end; unique edges counted ---
    std::vector<int64_t> edges_of((size_t)nv + 1, 0);  // unique edges with smaller end v
    on_threads(T, [&](int k) {
        for (int64_t v = cut[k]; v < cut[k + 1]; ++v) {
            int32_t *s = slot.data() + start[v];
            const int64_t m = start[v + 1] - start[v];
            for (int64_t i = 1; i < m; ++i) {  // insertion sort by (hi, slot)
                const int32_t q = s[i];
                int64_t j = i;
                while (j > 0 && (hi[s[j - 1]] > hi[q] || (hi[s[j - 1]] == hi[q] && s[j - 1] > q))) {
                    s[j] = s[j - 1];
                    --j;
                }
                s[j] = q;
            }
            int64_t u = 0;
            for (int64_t i = 0; i < m; ++i) u += (i == 0 || hi[s[i]] != hi[s[i - 1]]);
            edges_of[v] = u;
        }
    });
    int64_t ne = 0;
    for (int64_t v = 0; v < nv; ++v) {
        const int64_t u = edges_of[v];
        edges_of[v] = ne;
        ne += u;
    }
    edges_of[nv] = ne;
    *n_new = ne;
    STK_REQUIRE(ne <= capacity, "stk_tri_refine: %lld edges, room for %lld", (long long)ne, (long long)capacity);
    // --- edge of every slot; colour and sort key of every edge ---
    std::vector<int32_t> edge_of_slot((size_t)ns);
    std::vector<EdgeKey> keys((size_t)ne);
    std::vector<int32_t> e_lo((size_t)ne), e_hi((size_t)ne);
    std::atomic<int> clash{0};
    on_threads(T, [&](int k) {
        for (int64_t v = cut[k]; v < cut[k + 1]; ++v) {
            const int32_t *s = slot.data() + start[v];
            const int64_t m = start[v + 1] - start[v];
            int64_t e = edges_of[v] - 1;
            for (int64_t i = 0; i < m; ++i) {
                const int32_t q = s[i];
                if (i == 0 || hi[q] != hi[s[i - 1]]) {
                    ++e;
                    e_lo[e] = (int32_t)v, e_hi[e] = hi[q];
                    const double *a = points + 2 * v, *b = points + 2 * (int64_t)hi[q];
                    keys[e] = {tri_colors[q], 0.5 * (a[1] + b[1]), 0.5 * (a[0] + b[0]), e};
                }
                // NumPy's scatter keeps the colour of the LAST slot of an edge; the two
                // triangles of an edge agree by construction -- checked, not assumed
                if (keys[e].col != tri_colors[q]) clash = 1;
                edge_of_slot[q] = (int32_t)e;
            }
        }
    });
    STK_REQUIRE(!clash.load(), "stk_tri_refine: the two triangles of an edge give it different colours");
    // --- numbering of the new vertices ---
    sample_sort(keys, T, before);
    std::vector<int64_t> new_id((size_t)ne);
    on_threads(T, [&](int k) {
        for (int64_t r = share(ne, T, k); r < share(ne, T, k + 1); ++r) {
            const EdgeKey &q = keys[r];
            new_id[q.e] = nv + r;
            new_points[2 * r] = q.x, new_points[2 * r + 1] = q.y;
            new_parents[2 * r] = e_lo[q.e], new_parents[2 * r + 1] = e_hi[q.e];
            new_colors[r] = q.col;
        }
    });
    // --- children ---
    on_threads(T, [&](int k) {
        for (int64_t t = share(nt, T, k); t < share(nt, T, k + 1); ++t) {
            const int64_t *v = tris + 3 * t, *c = tri_colors + 3 * t;
            const int64_t m[3] = {new_id[edge_of_slot[3 * t]], new_id[edge_of_slot[3 * t + 1]],
                                  new_id[edge_of_slot[3 * t + 2]]};
            const int64_t kids[4][3] = {{v[0], m[2], m[1]}, {v[1], m[0], m[2]}, {v[2], m[1], m[0]}, {m[0], m[1], m[2]}};
            const int64_t cols[4][3] = {{c[0], c[1], c[2]}, {c[1], c[2], c[0]}, {c[2], c[0], c[1]}, {c[0], c[1], c[2]}};
            for (int b = 0; b < 4; ++b)
                for (int a = 0; a < 3; ++a) {
                    child_tris[3 * (b * nt + t) + a] = kids[b][a];
                    child_colors[3 * (b * nt + t) + a] = cols[b][a];
                }
        }
    });
    return 0;
}

// ---- load vector of a triangulation: int f phi_i with an nq-point rule ----------
// (reference heateq_mpi.py:102-103: LinearForm(u0 * v * dx).assemble() through NGSolve;
// source/assembly.py:space_load is the NumPy form).  Two calls around the caller's
// evaluation of f at the quadrature points: their coordinates, then the sums.

extern "C" int stk_p1_load_points_2d(int64_t nv, int64_t nt, const double *points, const int64_t *tris, int32_t nq,
                                     const double *rule_points, double *qx, double *qy)
{
    STK_REQUIRE(nv > 0 && nt > 0 && points && tris && nq > 0 && rule_points && qx && qy,
                "stk_p1_load_points_2d: bad arguments");
    for (int64_t q = 0; q < 3 * nt; ++q)
        STK_REQUIRE(tris[q] >= 0 && tris[q] < nv, "stk_p1_load_points_2d: triangle %lld names vertex %lld",
                    (long long)(q / 3), (long long)tris[q]);
    const int T = host_threads(nt * nq);
    on_threads(T, [&](int k) {
        for (int64_t t = share(nt, T, k); t < share(nt, T, k + 1); ++t) {
            const double *p0 = points + 2 * tris[3 * t], *p1 = points + 2 * tris[3 * t + 1],
                         *p2 = points + 2 * tris[3 * t + 2];
            for (int q = 0; q < nq; ++q) {
                const double *l = rule_points + 3 * q;
                qx[t * nq + q] = l[0] * p0[0] + l[1] * p1[0] + l[2] * p2[0];
                qy[t * nq + q] = l[0] * p0[1] + l[1] * p1[1] + l[2] * p2[1];
            }
        }
    });
    return 0;
}

extern "C" int stk_p1_load_sum_2d(int64_t nv, int64_t nt, const double *points, const int64_t *tris, int32_t nq,
                                  const double *rule_weights, const double *rule_points, const double *f, double *vec)
{
    STK_REQUIRE(nv > 0 && nt > 0 && points && tris && nq > 0 && rule_weights && rule_points && f && vec,
                "stk_p1_load_sum_2d: bad arguments");
    STK_REQUIRE(nv < ((int64_t)1 << 31) && nt < ((int64_t)1 << 29), "stk_p1_load_sum_2d: mesh too large for 32-bit tables");
    for (int64_t q = 0; q < 3 * nt; ++q)
        STK_REQUIRE(tris[q] >= 0 && tris[q] < nv, "stk_p1_load_sum_2d: triangle %lld names vertex %lld",
                    (long long)(q / 3), (long long)tris[q]);
    const int64_t ns = 3 * nt;
    const int T = host_threads(ns);
    // share of triangle t in the entry of its local vertex a: (sum_q f_q w_q l_qa) * |T|
    std::vector<double> loc((size_t)ns);
    std::vector<std::atomic<int32_t>> fill((size_t)nv + 1);
    on_threads(T, [&](int k) {
        for (int64_t v = share(nv + 1, T, k); v < share(nv + 1, T, k + 1); ++v) fill[v].store(0, std::memory_order_relaxed);
    });
    on_threads(T, [&](int k) {
        for (int64_t t = share(nt, T, k); t < share(nt, T, k + 1); ++t) {
            const double *p0 = points + 2 * tris[3 * t], *p1 = points + 2 * tris[3 * t + 1],
                         *p2 = points + 2 * tris[3 * t + 2];
            const double e0x = p1[0] - p0[0], e0y = p1[1] - p0[1];
            const double e1x = p2[0] - p0[0], e1y = p2[1] - p0[1];
            const double vol = std::fabs(e0x * e1y - e0y * e1x) / 2.0;
            for (int a = 0; a < 3; ++a) {
                double s = 0.0;
                for (int q = 0; q < nq; ++q) s += (f[t * nq + q] * rule_weights[q]) * rule_points[3 * q + a];
                loc[3 * t + a] = s * vol;
                fill[tris[3 * t + a]].fetch_add(1, std::memory_order_relaxed);
            }
        }
    });
    std::vector<int64_t> start((size_t)nv + 1);
    {
        int64_t run = 0;
        for (int64_t v = 0; v < nv; ++v) {
            start[v] = run;
            run += fill[v].load(std::memory_order_relaxed);
            fill[v].store(0, std::memory_order_relaxed);
        }
        start[nv] = run;
    }
    std::vector<int32_t> slot((size_t)ns);
    on_threads(T, [&](int k) {
        for (int64_t q = share(ns, T, k); q < share(ns, T, k + 1); ++q)
            slot[start[tris[q]] + fill[tris[q]].fetch_add(1, std::memory_order_relaxed)] = (int32_t)q;
    });
    // every entry: its shares in ascending (triangle, local vertex) -- the order of
    // np.bincount over the flattened cells, whatever order the threads filled the lists in
    on_threads(T, [&](int k) {
        for (int64_t v = share(nv, T, k); v < share(nv, T, k + 1); ++v) {
            int32_t *s = slot.data() + start[v];
            const int64_t m = start[v + 1] - start[v];
            std::sort(s, s + m);
            double sum = 0.0;
            for (int64_t i = 0; i < m; ++i) sum += loc[s[i]];
            vec[v] = sum;
        }
    });
    return 0;
}

// ---- processing order of the dofs: mesh tiles ------------------------------------
// (source/assembly.py:tile_order_from_coords, its NumPy form: np.lexsort over the
// coordinates and the tile indices floor((p - lo) / side), last axis slowest.)  The
// caller computes lo and side, so the only arithmetic here is that subtraction,
// division and floor: the order is NumPy's, index for index.

extern "C" int stk_tile_order(int64_t n, int32_t d, const double *coords, const double *lo, double side,
                              int32_t *order)
{
    STK_REQUIRE(n > 0 && (d == 2 || d == 3) && coords && lo && side > 0 && order, "stk_tile_order: bad arguments");
    STK_REQUIRE(n < ((int64_t)1 << 31), "stk_tile_order: %lld points do not fit 32-bit indices", (long long)n);
    const int T = host_threads(n);
    std::vector<TileKey> keys((size_t)n);
    on_threads(T, [&](int k) {
        for (int64_t i = share(n, T, k); i < share(n, T, k + 1); ++i) {
            TileKey &q = keys[i];
            for (int a = 0; a < 3; ++a) q.t[a] = 0, q.c[a] = 0.0;
            for (int a = 0; a < d; ++a) {
                q.c[a] = coords[i * d + a];
                q.t[a] = (int64_t)std::floor((q.c[a] - lo[a]) / side);
            }
            q.i = i;
        }
    });
    sample_sort(keys, T, [](const TileKey &a, const TileKey &b) {
        for (int k = 2; k >= 0; --k)
            if (a.t[k] != b.t[k]) return a.t[k] < b.t[k];
        for (int k = 2; k >= 0; --k)
            if (a.c[k] != b.c[k]) return a.c[k] < b.c[k];
        return a.i < b.i;  // lexsort is stable
    });
    on_threads(T, [&](int k) {
        for (int64_t r = share(n, T, k); r < share(n, T, k + 1); ++r) order[r] = (int32_t)keys[r].i;
    });
    return 0;
}

// ---- one CSR pattern for several matrices -----------------------------------------
// (source/linop.py:union_pattern: the fused kernels walk one pattern for all terms of
// a Kronecker sum, mpi_kron.py:186-200 SumMPI.)  Rows with ascending columns and no
// duplicates in, the merged rows out; every matrix's values land on the union in their
// own order, zeros where it has no entry.  Two calls: the row sizes, then the entries.

extern "C" int stk_csr_union_count(int64_t n, int32_t n_mats, const int32_t *const *indptr,
                                   const int32_t *const *indices, int32_t *out_indptr)
{
    STK_REQUIRE(n > 0 && n_mats >= 1 && n_mats <= 30 && indptr && indices && out_indptr, "stk_csr_union_count: bad arguments");
    const int T = host_threads(n * 8);
    std::atomic<int> bad{0};
    std::vector<int32_t> count((size_t)n);
    on_threads(T, [&](int k) {
        std::vector<int32_t> at((size_t)n_mats);
        for (int64_t r = share(n, T, k); r < share(n, T, k + 1); ++r) {
            for (int m = 0; m < n_mats; ++m) {
                at[m] = indptr[m][r];
                for (int32_t p = indptr[m][r] + 1; p < indptr[m][r + 1]; ++p)
                    if (indices[m][p] <= indices[m][p - 1]) bad = 1;
            }
            int32_t c = 0;
            for (;;) {
                int64_t col = INT64_MAX;
                for (int m = 0; m < n_mats; ++m)
                    if (at[m] < indptr[m][r + 1]) col = std::min<int64_t>(col, indices[m][at[m]]);
                if (col == INT64_MAX) break;
                for (int m = 0; m < n_mats; ++m)
                    if (at[m] < indptr[m][r + 1] && indices[m][at[m]] == col) ++at[m];
                ++c;
            }
            count[r] = c;
        }
    });
    STK_REQUIRE(!bad.load(), "stk_csr_union_count: a row's columns are not strictly ascending");
    int64_t run = 0;
    for (int64_t r = 0; r < n; ++r) {
        out_indptr[r] = (int32_t)run;
        run += count[r];
    }
    STK_REQUIRE(run < ((int64_t)1 << 31), "stk_csr_union_count: %lld entries do not fit 32-bit offsets", (long long)run);
    out_indptr[n] = (int32_t)run;
    return 0;
}

extern "C" int stk_csr_union_fill(int64_t n, int32_t n_mats, const int32_t *const *indptr,
                                  const int32_t *const *indices, const double *const *data,
                                  const int32_t *out_indptr, int32_t *out_indices, double *const *out_data)
{
    STK_REQUIRE(n > 0 && n_mats >= 1 && n_mats <= 30 && indptr && indices && data && out_indptr && out_indices && out_data,
                "stk_csr_union_fill: bad arguments");
    const int T = host_threads(n * 8);
    std::atomic<int> bad{0};
    on_threads(T, [&](int k) {
        std::vector<int32_t> at((size_t)n_mats);
        for (int64_t r = share(n, T, k); r < share(n, T, k + 1); ++r) {
            for (int m = 0; m < n_mats; ++m) at[m] = indptr[m][r];
            int32_t o = out_indptr[r];
            for (;;) {
                int64_t col = INT64_MAX;
                for (int m = 0; m < n_mats; ++m)
                    if (at[m] < indptr[m][r + 1]) col = std::min<int64_t>(col, indices[m][at[m]]);
                if (col == INT64_MAX) break;
                if (o >= out_indptr[r + 1]) {
                    bad = 1;
                    break;
                }
                out_indices[o] = (int32_t)col;
                for (int m = 0; m < n_mats; ++m) {
                    if (at[m] < indptr[m][r + 1] && indices[m][at[m]] == col)
                        out_data[m][o] = data[m][at[m]++];
                    else
                        out_data[m][o] = 0.0;
                }
                ++o;
            }
            if (o != out_indptr[r + 1]) bad = 1;
        }
    });
    STK_REQUIRE(!bad.load(), "stk_csr_union_fill: the row sizes are not those of stk_csr_union_count");
    return 0;
}
