// P1 mass and stiffness matrices of a triangulation on its free dofs, assembled
// row by row on the host threads of libstk (problem set-up, SURVEY section 8 row f1;
// replaces the NGSolve BilForm(...).assemble() calls of reference
// heateq_mpi.py:91-96 and the restriction to the free dofs of
// source/ngsolve_helper.py:38-45).  Host code only.
//
// Why not COO -> CSR: the NumPy/SciPy form of this assembly (source/assembly.py)
// builds 9 entries per triangle, lets SciPy bucket them and sum the duplicates in
// the order its sort leaves them, twice (mass, stiffness), and then filters the
// boundary out: 0.87 s of a 2.1 s set-up at J_space = 9.  Here every row collects
// its entries from the triangles around its vertex -- in ascending triangle
// number, the local matrices computed on the fly -- sorts them by column with a
// STABLE insertion sort and sums the duplicates from left to right, so the order
// of every addition is defined by the mesh alone:
//     a_ij = sum over triangles T containing i and j, ascending T, of
//            ((g_i . g_j) * vol)(T)          g = gradients of the barycentric
//     m_ij = the same sum of vol(T) * (1 + delta_ij) / 12   coordinates
// with g and vol formed exactly as source/assembly.py:_simplex_geometry does and
// without contraction into fused multiply-adds (the pragma below).  On the
// uniformly refined meshes of the BASELINE configurations every product and
// every partial sum is exact or a sum of equal terms, so the matrices are bit for
// bit those of the SciPy path whatever the order (tests/test_host_cpu.py
// test_p1_assembler_matches_scipy_path); on a general mesh they differ from it in
// the last bit of some entries, as two summation orders do.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "stk_common.h"

#pragma clang fp contract(off)

namespace {

struct Entry {
    int32_t col;
    double a, m;
};

struct RowBlock {  // what one thread produced for its range of rows
    std::vector<Entry> e;
    std::vector<int32_t> count;  // entries per row of the range
    double max_abs_a = 0.0;
};

struct Local {  // local matrices of one triangle
    double k[3][3], vol;
};

inline Local local_matrices(const double *p, const int64_t *c)
{
    const double *p0 = p + 2 * c[0], *p1 = p + 2 * c[1], *p2 = p + 2 * c[2];
    const double e0x = p1[0] - p0[0], e0y = p1[1] - p0[1];
    const double e1x = p2[0] - p0[0], e1y = p2[1] - p0[1];
    const double det = e0x * e1y - e0y * e1x;
    Local L;
    L.vol = std::fabs(det) / 2.0;
    double g[3][2];
    g[1][0] = e1y / det, g[1][1] = (-e1x) / det;
    g[2][0] = (-e0y) / det, g[2][1] = e0x / det;
    g[0][0] = -(g[1][0] + g[2][0]), g[0][1] = -(g[1][1] + g[2][1]);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) L.k[a][b] = (g[a][0] * g[b][0] + g[a][1] * g[b][1]) * L.vol;
    return L;
}

}  // namespace

struct stk_p1_result {
    int32_t n_free = 0;
    std::vector<int32_t> indptr_a, indices_a, indptr_m, indices_m;
    std::vector<double> data_a, data_m;
};

extern "C" int stk_p1_assemble_2d(int64_t nv, int64_t nc, const double *points, const int64_t *cells,
                                  const uint8_t *boundary, double zero_rel, stk_p1_result **out)
{
    STK_REQUIRE(nv > 0 && nc > 0 && points && cells && boundary && out, "stk_p1_assemble_2d: bad arguments");
    STK_REQUIRE(nv < ((int64_t)1 << 31), "stk_p1_assemble_2d: %lld vertices do not fit 32-bit indices", (long long)nv);
    for (int64_t t = 0; t < nc; ++t)
        for (int a = 0; a < 3; ++a)
            STK_REQUIRE(cells[3 * t + a] >= 0 && cells[3 * t + a] < nv, "stk_p1_assemble_2d: cell %lld names vertex %lld",
                        (long long)t, (long long)cells[3 * t + a]);
    // triangles around every vertex, ascending triangle number: inc[start[v] ..] = 3 * t + local index
    std::vector<int64_t> start(nv + 1, 0);
    for (int64_t q = 0; q < 3 * nc; ++q) ++start[cells[q] + 1];
    for (int64_t v = 0; v < nv; ++v) start[v + 1] += start[v];
    std::vector<int64_t> inc(3 * nc);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t q = 0; q < 3 * nc; ++q) inc[fill[cells[q]]++] = q;
    }
    std::vector<int32_t> new_id(nv);
    int32_t n_free = 0;
    for (int64_t v = 0; v < nv; ++v) new_id[v] = boundary[v] ? -1 : n_free++;

    int T = (int)std::thread::hardware_concurrency();
    if (const char *env = getenv("STK_HOST_THREADS")) T = atoi(env);
    T = std::max(1, std::min(T, 32));
    if (nv < 4096) T = 1;
    // row ranges with equal shares of the incidences
    std::vector<int64_t> cut(T + 1, nv);
    cut[0] = 0;
    for (int k = 1; k < T; ++k)
        cut[k] = std::lower_bound(start.begin(), start.end(), 3 * nc * k / T) - start.begin();
    for (int k = 1; k <= T; ++k) cut[k] = std::max(cut[k], cut[k - 1]);
    std::vector<RowBlock> blocks(T);

    auto rows = [&](int k) {
        RowBlock &B = blocks[k];
        const int64_t v0 = cut[k], v1 = cut[k + 1];
        B.count.assign(v1 - v0, 0);
        B.e.reserve((size_t)(v1 - v0) * 8);
        std::vector<Entry> row;
        for (int64_t v = v0; v < v1; ++v) {
            // every row is summed (the drop threshold below looks at boundary rows too,
            // as source/assembly.py does), only free rows are kept
            row.clear();
            for (int64_t q = start[v]; q < start[v + 1]; ++q) {
                const int64_t t = inc[q] / 3;
                const int a = (int)(inc[q] - 3 * t);
                const Local L = local_matrices(points, cells + 3 * t);
                for (int b = 0; b < 3; ++b) {
                    const double m = L.vol * (a == b ? 2.0 / 12.0 : 1.0 / 12.0);
                    row.push_back({(int32_t)cells[3 * t + b], L.k[a][b], m});
                }
            }
            // stable insertion sort by column (rows have a few dozen entries at most)
            for (size_t i = 1; i < row.size(); ++i) {
                const Entry x = row[i];
                size_t j = i;
                for (; j > 0 && row[j - 1].col > x.col; --j) row[j] = row[j - 1];
                row[j] = x;
            }
            size_t n = 0;
            for (size_t i = 0; i < row.size(); ++i) {
                if (n > 0 && row[n - 1].col == row[i].col) {
                    row[n - 1].a += row[i].a;
                    row[n - 1].m += row[i].m;
                } else {
                    row[n++] = row[i];
                }
            }
            for (size_t i = 0; i < n; ++i) B.max_abs_a = std::max(B.max_abs_a, std::fabs(row[i].a));
            if (boundary[v]) continue;
            int32_t kept = 0;
            for (size_t i = 0; i < n; ++i)
                if (!boundary[row[i].col]) {
                    B.e.push_back({new_id[row[i].col], row[i].a, row[i].m});
                    ++kept;
                }
            B.count[v - v0] = kept;
        }
    };
    {
        std::vector<std::thread> pool;
        for (int k = 1; k < T; ++k) pool.emplace_back(rows, k);
        rows(0);
        for (auto &th : pool) th.join();
    }
    double max_abs_a = 0.0;
    for (const RowBlock &B : blocks) max_abs_a = std::max(max_abs_a, B.max_abs_a);
    const double drop = zero_rel * max_abs_a;  // stiffness entries that are zero up to rounding (assembly.py:217-220)

    stk_p1_result *R = new stk_p1_result();
    R->n_free = n_free;
    R->indptr_a.assign((size_t)n_free + 1, 0);
    R->indptr_m.assign((size_t)n_free + 1, 0);
    // entries per free row, then offsets, then the copy: per block, in parallel
    std::vector<int64_t> first_a(T + 1, 0), first_m(T + 1, 0);
    auto count = [&](int k) {
        const RowBlock &B = blocks[k];
        size_t pos = 0;
        int64_t na = 0, nm = 0;
        for (int64_t v = cut[k]; v < cut[k + 1]; ++v) {
            if (boundary[v]) continue;
            int32_t ca = 0, cm = 0;
            for (int32_t i = 0; i < B.count[v - cut[k]]; ++i, ++pos) {
                const Entry &x = B.e[pos];
                ca += (x.a != 0.0 && !(std::fabs(x.a) < drop)) ? 1 : 0;
                cm += (x.m != 0.0) ? 1 : 0;
            }
            R->indptr_a[new_id[v] + 1] = ca;
            R->indptr_m[new_id[v] + 1] = cm;
            na += ca, nm += cm;
        }
        first_a[k + 1] = na, first_m[k + 1] = nm;
    };
    {
        std::vector<std::thread> pool;
        for (int k = 1; k < T; ++k) pool.emplace_back(count, k);
        count(0);
        for (auto &th : pool) th.join();
    }
    for (int k = 0; k < T; ++k) first_a[k + 1] += first_a[k], first_m[k + 1] += first_m[k];
    if (first_a[T] >= ((int64_t)1 << 31) || first_m[T] >= ((int64_t)1 << 31)) {
        delete R;
        stk_set_error("stk_p1_assemble_2d: more than 2^31 entries");
        return 2;
    }
    for (int32_t i = 0; i < n_free; ++i) {
        R->indptr_a[i + 1] += R->indptr_a[i];
        R->indptr_m[i + 1] += R->indptr_m[i];
    }
    R->indices_a.resize(first_a[T]);
    R->data_a.resize(first_a[T]);
    R->indices_m.resize(first_m[T]);
    R->data_m.resize(first_m[T]);
    auto copy = [&](int k) {
        const RowBlock &B = blocks[k];
        size_t pos = 0;
        int64_t pa = first_a[k], pm = first_m[k];
        for (int64_t v = cut[k]; v < cut[k + 1]; ++v) {
            if (boundary[v]) continue;
            for (int32_t i = 0; i < B.count[v - cut[k]]; ++i, ++pos) {
                const Entry &x = B.e[pos];
                if (x.a != 0.0 && !(std::fabs(x.a) < drop)) {
                    R->indices_a[pa] = x.col;
                    R->data_a[pa++] = x.a;
                }
                if (x.m != 0.0) {
                    R->indices_m[pm] = x.col;
                    R->data_m[pm++] = x.m;
                }
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (int k = 1; k < T; ++k) pool.emplace_back(copy, k);
        copy(0);
        for (auto &th : pool) th.join();
    }
    *out = R;
    return 0;
}

extern "C" int stk_p1_result_sizes(const stk_p1_result *r, int32_t *n_free, int64_t *nnz_a, int64_t *nnz_m)
{
    STK_REQUIRE(r && n_free && nnz_a && nnz_m, "stk_p1_result_sizes: null pointer");
    *n_free = r->n_free;
    *nnz_a = (int64_t)r->data_a.size();
    *nnz_m = (int64_t)r->data_m.size();
    return 0;
}

extern "C" int stk_p1_result_copy(const stk_p1_result *r, int32_t which, int32_t *indptr, int32_t *indices, double *data)
{
    STK_REQUIRE(r && indptr && indices && data && (which == 0 || which == 1), "stk_p1_result_copy: bad arguments");
    const std::vector<int32_t> &ip = which == 0 ? r->indptr_a : r->indptr_m;
    const std::vector<int32_t> &ix = which == 0 ? r->indices_a : r->indices_m;
    const std::vector<double> &dv = which == 0 ? r->data_a : r->data_m;
    std::memcpy(indptr, ip.data(), ip.size() * sizeof(int32_t));
    if (!ix.empty()) std::memcpy(indices, ix.data(), ix.size() * sizeof(int32_t));
    if (!dv.empty()) std::memcpy(data, dv.data(), dv.size() * sizeof(double));
    return 0;
}

extern "C" int stk_p1_result_free(stk_p1_result *r)
{
    delete r;
    return 0;
}
