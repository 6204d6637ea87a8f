// Plan construction behind the C ABI: from the CSR matrices a caller of the
// reference holds (mpi_shared_mem.py:46-48: int32 pattern, float64 values) to
// the device-resident forms the Kronecker kernels stream -- union pattern,
// sliced-ELL copy in a processing order, dictionary of value tuples, packed
// slot words -- so that a host program in any language reaches
// the fast path of  y = beta*y + sum_k (T_k kron X_k) x  (reference
// TridiagKronMatMPI / SumMPI, mpi_kron.py:77-90, 204-222) with three calls:
// stk_kron_plan_create, stk_kron_plan_apply, stk_kron_plan_destroy.
// The Python classes build the same forms with NumPy (source/linop.py); tests
// compare the results of the two bit for bit (the numbering of the dictionary
// codes may differ, the products and their order do not).
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "stk_common.h"

struct stk_kron_plan {
    int32_t M = 0, K = 0, n_mats = 0;
    bool packed = false;
    // plain ELL form
    stk_ell_pattern ell{};
    std::vector<double *> ell_vals, ovf_vals;
    // packed form: one row per slot row, and row pairs (slabs of >= PAIR_MIN_STEPS steps)
    stk_pack_pattern pack{}, pack_pairs{};
    bool paired = false;
    // row pairs with explicit values (matrices without a dictionary): slot words
    // and rows on the device, the values of all matrices on the host; the device
    // array of a call lists the matrices of its terms in term order and is built
    // on first use (stk_pack_pattern.vals)
    bool explicit_pairs = false;
    stk_pack_pattern pairs_x{};
    std::vector<double> pairs_x_vals;  // [units][K][2][n_mats]
    std::map<std::vector<int32_t>, double *> pairs_x_for;
    std::vector<void *> owned;  // every device allocation
    int64_t nnz_union = 0;
};

int g_plan_pack_rows = 2;  // tuning key "pack_rows": matrix rows per slot row of the packed form (1 or 2)

// Slots of a unit of up to `rp` rows with K slots each (the instantiations of
// csrc/kron_pack.hip); 0: no such form.
extern "C" int32_t stk_pack_unit_slots(int32_t K, int32_t rp)
{
    if (rp == 1) return K;
    if (rp == 2) return K == 5 ? 8 : K == 7 ? 10 : K == 9 ? 12 : 0;
    return 0;  // three / four rows per unit (K = 7 -> 13 / 16): measured slower than pairs, not instantiated
}

// Row pairs need rows that follow each other in the processing order to share
// columns.  A mesh-tile order of a structured mesh has that by itself (x-neighbours
// are consecutive); the order of an unstructured mesh does not.  This walks the
// given order and moves, behind every row that has not been placed yet, ONE of its
// not yet placed neighbours (a column of the row) whose union of columns with it
// fits K_out slots -- the nearest one in the given order, at most `window`
// positions away, so that the locality of the order survives.  perm[q] = position
// (in the given order) of the row that goes to position q of the new order.
extern "C" int stk_pack_match_order(int32_t M, int32_t K, const int32_t *counts, const int32_t *cols,
                                    const int32_t *own, int32_t K_out, int32_t window, int32_t *perm)
{
    STK_REQUIRE(M > 0 && K >= 1 && counts && cols && own && perm && K_out >= K && window >= 1,
                "stk_pack_match_order: bad arguments");
    std::vector<int32_t> pos_of((size_t)M, -1);
    for (int32_t p = 0; p < M; ++p) {
        STK_REQUIRE(own[p] >= 0 && own[p] < M && pos_of[own[p]] < 0, "stk_pack_match_order: own is not a permutation");
        pos_of[own[p]] = p;
    }
    std::vector<char> placed((size_t)M, 0);
    int32_t q = 0;
    for (int32_t p = 0; p < M; ++p) {
        if (placed[p]) continue;
        placed[p] = 1;
        perm[q++] = p;
        const int32_t *ca = cols + (size_t)p * K;
        const int na = counts[p];
        int32_t best = -1, best_dist = window + 1;
        for (int e = 0; e < na; ++e) {
            const int32_t c = ca[e];
            if (c < 0 || c >= M) continue;
            const int32_t pc = pos_of[c];
            if (pc == p || placed[pc]) continue;
            const int32_t dist = pc > p ? pc - p : p - pc;
            if (dist >= best_dist) continue;
            // size of the union of the two sorted column lists
            const int32_t *cb = cols + (size_t)pc * K;
            const int nb = counts[pc];
            int ia = 0, ib = 0, m = 0;
            while (ia < na || ib < nb) {
                const int32_t va = ia < na ? ca[ia] : INT32_MAX, vb = ib < nb ? cb[ib] : INT32_MAX;
                ia += va <= vb, ib += vb <= va, ++m;
            }
            if (m <= K_out) best = pc, best_dist = dist;
        }
        if (best >= 0) {
            placed[best] = 1;
            perm[q++] = best;
        }
    }
    return 0;
}

// Rows that follow each other in the processing order and share columns are
// served by one slot row ("unit") of the packed form: greedily, left to right, a
// unit takes the next row as long as it has fewer than `rp` rows and the union of
// the columns still fits K_out slots.  cols/codes: [M][K] with the real entries
// of position p in the first counts[p] slots, columns ascending.  Outputs (sized
// for M units): ucols [units][K_out] (ascending union, unused slots = the first
// row's own column), ucodes [units][K_out][rp] (zero_code where a row has no
// entry in the column), urows [units][rp] (-1: no row).
extern "C" int stk_pack_group_rows(int32_t M, int32_t K, const int32_t *counts, const int32_t *cols,
                                   const int32_t *codes, const int32_t *own, int32_t zero_code, int32_t rp,
                                   int32_t K_out, int32_t *n_units, int32_t *ucols, int32_t *ucodes, int32_t *urows)
{
    STK_REQUIRE(M > 0 && K >= 1 && counts && cols && codes && own && n_units && ucols && ucodes && urows,
                "stk_pack_group_rows: bad arguments");
    STK_REQUIRE(rp >= 1 && rp <= 4 && K_out >= K && K_out <= 64, "stk_pack_group_rows: rp=%d K_out=%d (K=%d)", rp,
                K_out, K);
    std::vector<int32_t> ucol(K_out), next(K_out);
    std::vector<int32_t> ucode((size_t)K_out * rp), ncode((size_t)K_out * rp);
    int32_t units = 0;
    for (int32_t pos = 0; pos < M;) {
        STK_REQUIRE(counts[pos] >= 0 && counts[pos] <= K, "stk_pack_group_rows: counts[%d]=%d", pos, counts[pos]);
        int n = counts[pos], rows = 1;
        for (int e = 0; e < n; ++e) {
            ucol[e] = cols[(size_t)pos * K + e];
            for (int j = 0; j < rp; ++j) ucode[(size_t)e * rp + j] = j == 0 ? codes[(size_t)pos * K + e] : zero_code;
        }
        while (rows < rp && pos + rows < M) {
            const int32_t q = pos + rows;
            const int nb = counts[q];
            if (nb < 0 || nb > K) break;
            // merge the union so far with row q
            int ia = 0, ib = 0, m = 0;
            bool fits = true;
            while (ia < n || ib < nb) {
                if (m == K_out) {
                    fits = false;
                    break;
                }
                const int32_t ca = ia < n ? ucol[ia] : INT32_MAX;
                const int32_t cb = ib < nb ? cols[(size_t)q * K + ib] : INT32_MAX;
                next[m] = std::min(ca, cb);
                for (int j = 0; j < rp; ++j)
                    ncode[(size_t)m * rp + j] = ca <= cb ? ucode[(size_t)ia * rp + j] : zero_code;
                if (cb <= ca) ncode[(size_t)m * rp + rows] = codes[(size_t)q * K + ib];
                ia += ca <= cb, ib += cb <= ca, ++m;
            }
            if (!fits) break;
            n = m;
            std::copy(next.begin(), next.begin() + n, ucol.begin());
            std::copy(ncode.begin(), ncode.begin() + (size_t)n * rp, ucode.begin());
            ++rows;
        }
        int32_t *oc = ucols + (size_t)units * K_out, *od = ucodes + (size_t)units * K_out * rp;
        for (int e = 0; e < K_out; ++e) {
            oc[e] = e < n ? ucol[e] : own[pos];
            for (int j = 0; j < rp; ++j) od[(size_t)e * rp + j] = e < n ? ucode[(size_t)e * rp + j] : zero_code;
        }
        for (int j = 0; j < rp; ++j) urows[(size_t)units * rp + j] = j < rows ? own[pos + j] : -1;
        ++units;
        pos += rows;
    }
    *n_units = units;
    return 0;
}

namespace {

template <typename T>
int upload(stk_kron_plan *p, const std::vector<T> &host, T **dev)
{
    *dev = nullptr;
    if (host.empty()) return 0;
    STK_HIP(hipMalloc((void **)dev, host.size() * sizeof(T)));
    p->owned.push_back(*dev);
    STK_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// Row pairs for matrices whose values do not repeat: partners first brought next to
// each other (stk_pack_match_order), then grouped like the dictionary form with
// every entry as its own code (its position in the ELL arrays, 0 = no entry).
int build_explicit_pairs(stk_kron_plan *p, int32_t M, int K, int n_mats, int col_bits,
                         const std::vector<int32_t> &counts, const std::vector<int32_t> &ell_idx,
                         const std::vector<int32_t> &row_ids, const std::vector<std::vector<double>> &ell_val)
{
    const int rp = 2, K2 = stk_pack_unit_slots(K, rp);
    if (K2 == 0 || g_plan_pack_rows < 2) return 0;
    std::vector<int32_t> perm(M);
    if (stk_pack_match_order(M, K, counts.data(), ell_idx.data(), row_ids.data(), K2, 8192, perm.data())) return 1;
    std::vector<int32_t> cnt2(M), idx2((size_t)M * K), own2(M), ent2((size_t)M * K);
    for (int q = 0; q < M; ++q) {
        const int pos = perm[q];
        cnt2[q] = counts[pos];
        own2[q] = row_ids[pos];
        for (int e = 0; e < K; ++e) {
            idx2[(size_t)q * K + e] = ell_idx[(size_t)pos * K + e];
            ent2[(size_t)q * K + e] = (int32_t)((size_t)pos * K + e) + 1;
        }
    }
    std::vector<int32_t> ucol((size_t)M * K2), ucode((size_t)M * K2 * rp), urows((size_t)M * rp);
    int32_t n_units = 0;
    if (stk_pack_group_rows(M, K, cnt2.data(), idx2.data(), ent2.data(), own2.data(), 0, rp, K2, &n_units, ucol.data(),
                            ucode.data(), urows.data()))
        return 1;
    const size_t U = (size_t)n_units;
    if (U > (size_t)(0.95 * M)) return 0;
    std::vector<uint32_t> slots(U * K2);
    for (size_t s2 = 0; s2 < U * K2; ++s2) slots[s2] = (uint32_t)ucol[s2];
    p->pairs_x_vals.assign(U * K2 * rp * n_mats, 0.0);
    for (size_t s2 = 0; s2 < U * K2 * rp; ++s2) {
        const int32_t ent = ucode[s2];
        if (ent > 0)
            for (int m = 0; m < n_mats; ++m) p->pairs_x_vals[s2 * n_mats + m] = ell_val[m][(size_t)ent - 1];
    }
    urows.resize(U * rp);
    uint32_t *d_slots;
    int32_t *d_urows;
    if (upload(p, slots, &d_slots) || upload(p, urows, &d_urows)) return 1;
    p->pairs_x = stk_pack_pattern{M, K2, col_bits, 1, n_mats, rp, (int32_t)U, d_slots, d_urows, nullptr, nullptr};
    p->explicit_pairs = true;
    return 0;
}

// The explicit-value pattern for the matrices of a call's terms, in term order.
const stk_pack_pattern *explicit_pattern_for(stk_kron_plan *p, int32_t n_terms, const stk_kron_pack_term *t,
                                             stk_pack_pattern *out)
{
    std::vector<int32_t> mats(n_terms);
    for (int k = 0; k < n_terms; ++k) mats[k] = t[k].mat;
    auto it = p->pairs_x_for.find(mats);
    if (it == p->pairs_x_for.end()) {
        const size_t slots = (size_t)p->pairs_x.n_units * p->pairs_x.K * 2;
        std::vector<double> sub(slots * n_terms);
        for (size_t s2 = 0; s2 < slots; ++s2)
            for (int k = 0; k < n_terms; ++k) sub[s2 * n_terms + k] = p->pairs_x_vals[s2 * p->n_mats + mats[k]];
        double *dev = nullptr;
        if (upload(p, sub, &dev)) return nullptr;
        it = p->pairs_x_for.emplace(mats, dev).first;
    }
    *out = p->pairs_x;
    out->n_mats = n_terms;
    out->vals = it->second;
    return out;
}

int build(stk_kron_plan *p, int32_t M, int32_t n_mats, const int32_t *const *indptr, const int32_t *const *indices,
          const double *const *data, const int32_t *order)
{
    p->M = M;
    p->n_mats = n_mats;
    // ---- union pattern, row by row, in processing order ------------------------
    std::vector<int32_t> u_ptr(M + 1, 0), u_idx;
    std::vector<std::vector<double>> u_val(n_mats);
    std::vector<int32_t> cols;
    int kmax = 0;
    for (int pos = 0; pos < M; ++pos) {
        const int i = order ? order[pos] : pos;
        cols.clear();
        for (int m = 0; m < n_mats; ++m)
            cols.insert(cols.end(), indices[m] + indptr[m][i], indices[m] + indptr[m][i + 1]);
        std::sort(cols.begin(), cols.end());
        cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
        const size_t base = u_idx.size();
        u_idx.insert(u_idx.end(), cols.begin(), cols.end());
        for (int m = 0; m < n_mats; ++m) {
            u_val[m].resize(u_idx.size(), 0.0);
            for (int e = indptr[m][i]; e < indptr[m][i + 1]; ++e) {
                const size_t at = std::lower_bound(cols.begin(), cols.end(), indices[m][e]) - cols.begin();
                u_val[m][base + at] += data[m][e];  // duplicate entries of a row add up, as in CSR arithmetic
            }
        }
        u_ptr[pos + 1] = (int32_t)u_idx.size();
        kmax = std::max(kmax, (int)cols.size());
    }
    p->nnz_union = (int64_t)u_idx.size();
    static const int slots[] = {5, 7, 9, 12, 16};
    int K = 16;
    for (int s : slots)
        if (s >= kmax) {
            K = s;
            break;
        }
    p->K = K;
    // ---- sliced ELL: K slots per row, the rest into an overflow CSR -------------
    std::vector<int32_t> ell_idx((size_t)M * K), row_ids(M), ovf_ptr(M + 1, 0), ovf_idx;
    std::vector<std::vector<double>> ell_val(n_mats, std::vector<double>((size_t)M * K, 0.0)), ovf_val(n_mats);
    for (int pos = 0; pos < M; ++pos) {
        const int i = order ? order[pos] : pos;
        row_ids[pos] = i;
        const int n = u_ptr[pos + 1] - u_ptr[pos];
        for (int e = 0; e < K; ++e) {
            ell_idx[(size_t)pos * K + e] = e < n ? u_idx[u_ptr[pos] + e] : i;  // unused: own column, value 0
            if (e < n)
                for (int m = 0; m < n_mats; ++m) ell_val[m][(size_t)pos * K + e] = u_val[m][u_ptr[pos] + e];
        }
        for (int e = K; e < n; ++e) {
            ovf_idx.push_back(u_idx[u_ptr[pos] + e]);
            for (int m = 0; m < n_mats; ++m) ovf_val[m].push_back(u_val[m][u_ptr[pos] + e]);
        }
        ovf_ptr[pos + 1] = (int32_t)ovf_idx.size();
    }
    const bool overflow = !ovf_idx.empty();
    int32_t *d_idx, *d_rows, *d_optr = nullptr, *d_oidx = nullptr;
    if (upload(p, ell_idx, &d_idx) || upload(p, row_ids, &d_rows)) return 1;
    if (overflow && (upload(p, ovf_ptr, &d_optr) || upload(p, ovf_idx, &d_oidx))) return 1;
    p->ell = stk_ell_pattern{M, K, d_idx, order ? d_rows : nullptr, d_optr, d_oidx};
    p->ell_vals.resize(n_mats);
    p->ovf_vals.assign(n_mats, nullptr);
    for (int m = 0; m < n_mats; ++m) {
        if (upload(p, ell_val[m], &p->ell_vals[m])) return 1;
        if (overflow && upload(p, ovf_val[m], &p->ovf_vals[m])) return 1;
    }
    // ---- dictionary of value tuples, packed slots, row records ------------------
    if (overflow) return 0;
    int col_bits = 1;
    while (((int64_t)1 << col_bits) < M) ++col_bits;
    const int64_t max_codes = std::min<int64_t>((int64_t)1 << (32 - col_bits), 512);  // the dictionary lives in LDS
    std::map<std::vector<uint64_t>, uint32_t> dict;  // bit patterns: +0.0 and -0.0 stay apart
    std::vector<uint32_t> code((size_t)M * K);
    std::vector<uint64_t> key(n_mats);
    for (size_t s = 0; s < (size_t)M * K; ++s) {
        for (int m = 0; m < n_mats; ++m) std::memcpy(&key[m], &ell_val[m][s], 8);
        auto it = dict.find(key);
        if (it == dict.end()) {
            if ((int64_t)dict.size() >= max_codes) {
                // too many distinct tuples for a dictionary: rows may still share
                // COLUMNS -- pairs with explicit values (kron_pack.hip, DICT = false)
                std::vector<int32_t> counts(M);
                for (int pos = 0; pos < M; ++pos) counts[pos] = u_ptr[pos + 1] - u_ptr[pos];
                return build_explicit_pairs(p, M, K, n_mats, col_bits, counts, ell_idx, row_ids, ell_val);
            }
            it = dict.emplace(key, (uint32_t)dict.size()).first;
        }
        code[s] = it->second;
    }
    // codes numbered in the order of the tuples' bit patterns (deterministic)
    std::vector<uint32_t> rank(dict.size());
    {
        uint32_t r = 0;
        for (auto &kv : dict) rank[kv.second] = r++;
    }
    const int n_codes = (int)dict.size();
    std::vector<double> table((size_t)n_mats * n_codes);
    for (auto &kv : dict)
        for (int m = 0; m < n_mats; ++m) std::memcpy(&table[(size_t)m * n_codes + rank[kv.second]], &kv.first[m], 8);
    std::vector<uint32_t> slots_w((size_t)M * K);
    for (size_t s = 0; s < (size_t)M * K; ++s) slots_w[s] = (rank[code[s]] << col_bits) | (uint32_t)ell_idx[s];
    uint32_t *d_slots;
    double *d_dict;
    if (upload(p, slots_w, &d_slots) || upload(p, table, &d_dict)) return 1;
    p->pack = stk_pack_pattern{M, K, col_bits, n_codes, n_mats, 1, M, d_slots, order ? d_rows : nullptr, d_dict};
    p->packed = true;
    // ---- several rows per slot row (stk_pack_group_rows above; source/linop.py
    // calls the same function) -----------------------------------------------------
    const int rp = g_plan_pack_rows;
    const int K2 = stk_pack_unit_slots(K, rp);
    if (K2 == 0 || rp < 2) return 0;
    uint32_t zero_code = (uint32_t)n_codes;  // the tuple "no entry": +0.0 in every matrix
    {
        std::fill(key.begin(), key.end(), (uint64_t)0);
        auto it = dict.find(key);
        if (it != dict.end()) zero_code = rank[it->second];
    }
    const int64_t n1 = zero_code == (uint32_t)n_codes ? n_codes + 1 : n_codes;
    std::vector<int32_t> counts(M), rcode((size_t)M * K);
    for (int pos = 0; pos < M; ++pos) counts[pos] = u_ptr[pos + 1] - u_ptr[pos];
    for (size_t s2 = 0; s2 < (size_t)M * K; ++s2) rcode[s2] = (int32_t)rank[code[s2]];
    std::vector<int32_t> ucol((size_t)M * K2), ucode((size_t)M * K2 * rp), urows((size_t)M * rp);
    int32_t n_units = 0;
    if (stk_pack_group_rows(M, K, counts.data(), ell_idx.data(), rcode.data(), row_ids.data(), (int32_t)zero_code, rp,
                            K2, &n_units, ucol.data(), ucode.data(), urows.data()))
        return 1;
    const size_t U = (size_t)n_units;
    if (U > (size_t)(0.95 * M)) return 0;  // hardly any rows share a unit: the one-row form stays
    std::map<int64_t, uint32_t> unit_codes;  // sum_j code_j * n1^(rp-1-j) -> number (assigned below, in key order)
    std::vector<int64_t> combined(U * K2);
    for (size_t s2 = 0; s2 < U * K2; ++s2) {
        int64_t c = 0;
        for (int j = 0; j < rp; ++j) c = c * n1 + ucode[s2 * rp + j];
        combined[s2] = c;
        unit_codes.emplace(c, 0u);
    }
    if ((int64_t)unit_codes.size() * rp > 1024 || (int64_t)unit_codes.size() > ((int64_t)1 << (32 - col_bits)))
        return 0;
    {
        uint32_t r = 0;
        for (auto &kv : unit_codes) kv.second = r++;
    }
    const int n_pair = (int)unit_codes.size();
    auto value_of = [&](int64_t c, int m) { return c < n_codes ? table[(size_t)m * n_codes + c] : 0.0; };
    std::vector<double> pair_table((size_t)n_mats * n_pair * rp);
    for (auto &kv : unit_codes) {
        int64_t c = kv.first;
        for (int j = rp - 1; j >= 0; --j, c /= n1)
            for (int m = 0; m < n_mats; ++m)
                pair_table[((size_t)m * n_pair + kv.second) * rp + j] = value_of(c % n1, m);
    }
    std::vector<uint32_t> pair_slots(U * K2);
    for (size_t s2 = 0; s2 < U * K2; ++s2)
        pair_slots[s2] = (unit_codes[combined[s2]] << col_bits) | (uint32_t)ucol[s2];
    urows.resize(U * rp);
    uint32_t *d_pslots;
    int32_t *d_urows;
    double *d_pdict;
    if (upload(p, pair_slots, &d_pslots) || upload(p, urows, &d_urows) || upload(p, pair_table, &d_pdict)) return 1;
    p->pack_pairs = stk_pack_pattern{M, K2, col_bits, n_pair, n_mats, rp, (int32_t)U, d_pslots, d_urows, d_pdict};
    p->paired = true;
    return 0;
}

}  // namespace

extern "C" int stk_kron_plan_create(int32_t M, int32_t n_mats, const int32_t *const *indptr_host,
                                    const int32_t *const *indices_host, const double *const *data_host,
                                    const int32_t *row_order_host, stk_kron_plan **out)
{
    STK_REQUIRE(M > 0 && n_mats >= 1 && n_mats <= 8 && indptr_host && indices_host && data_host && out,
                "stk_kron_plan_create: bad arguments");
    for (int m = 0; m < n_mats; ++m)
        STK_REQUIRE(indptr_host[m] && indices_host[m] && data_host[m] && indptr_host[m][0] == 0,
                    "stk_kron_plan_create: matrix %d incomplete", m);
    if (row_order_host) {
        std::vector<char> seen(M, 0);
        for (int pos = 0; pos < M; ++pos) {
            const int i = row_order_host[pos];
            STK_REQUIRE(i >= 0 && i < M && !seen[i], "stk_kron_plan_create: row_order is not a permutation");
            seen[i] = 1;
        }
    }
    for (int m = 0; m < n_mats; ++m)
        for (int32_t e = 0; e < indptr_host[m][M]; ++e)
            STK_REQUIRE(indices_host[m][e] >= 0 && indices_host[m][e] < M,
                        "stk_kron_plan_create: matrix %d has column %d outside 0..%d", m, indices_host[m][e], M - 1);
    stk_kron_plan *p = new stk_kron_plan();
    const int rc = build(p, M, n_mats, indptr_host, indices_host, data_host, row_order_host);
    if (rc) {
        stk_kron_plan_destroy(p);
        return rc;
    }
    *out = p;
    return 0;
}

extern "C" int stk_kron_plan_destroy(stk_kron_plan *p)
{
    if (!p) return 0;
    for (void *d : p->owned) (void)hipFree(d);
    delete p;
    return 0;
}

extern "C" int stk_kron_plan_info(const stk_kron_plan *p, int32_t *K, int32_t *n_codes, int32_t *packed,
                                  int64_t *nnz_union, int32_t *rows_per_unit)
{
    STK_REQUIRE(p, "stk_kron_plan_info: null plan");
    if (rows_per_unit) *rows_per_unit = p->paired ? p->pack_pairs.rows_per_unit : (p->explicit_pairs ? 2 : 1);
    if (K) *K = p->K;
    if (n_codes) *n_codes = p->packed ? p->pack.n_codes : 0;
    if (packed) *packed = p->packed ? 1 : 0;
    if (nnz_union) *nnz_union = p->nnz_union;
    return 0;
}

extern "C" int stk_kron_plan_apply(stk_kron_plan *p, void *stream, int32_t n_loc, int32_t ld, int32_t n_terms,
                                   const stk_kron_pack_term *t, const double *x, const double *x_lo,
                                   const double *x_hi, double *ghost_work, double beta, double *y)
{
    STK_REQUIRE(p && t && x && y, "stk_kron_plan_apply: null pointer");
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_plan_apply: n_terms=%d not in 1..3", n_terms);
    for (int k = 0; k < n_terms; ++k)
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < p->n_mats, "stk_kron_plan_apply: term %d names matrix %d of %d", k,
                    t[k].mat, p->n_mats);
    const bool ghosts = x_lo || x_hi;
    if (p->packed) {
        if (ghosts) {
            STK_REQUIRE(ghost_work, "stk_kron_plan_apply: ghost rows need ghost_work (2*M doubles)");
            int rc = stk_interleave_ghosts(stream, p->M, x_lo, x_hi, ghost_work);
            if (rc) return rc;
        }
        // short slabs keep one row per slot row (measured: source/linop.py PAIR_MIN_STEPS)
        const stk_pack_pattern *form = p->paired && n_loc >= 8 ? &p->pack_pairs : &p->pack;
        return stk_kron_pack_apply(stream, form, n_loc, ld, n_terms, t, x, ghosts ? ghost_work : nullptr, beta, y);
    }
    if (p->explicit_pairs && n_loc >= 24) {
        // no dictionary, but pairs: the values of the terms' matrices travel with the slots
        stk_pack_pattern form;
        if (!explicit_pattern_for(p, n_terms, t, &form)) return 1;
        stk_kron_pack_term in_order[3];
        for (int k = 0; k < n_terms; ++k) in_order[k] = stk_kron_pack_term{t[k].tri, k};
        if (ghosts) {
            STK_REQUIRE(ghost_work, "stk_kron_plan_apply: ghost rows need ghost_work (2*M doubles)");
            int rc = stk_interleave_ghosts(stream, p->M, x_lo, x_hi, ghost_work);
            if (rc) return rc;
        }
        return stk_kron_pack_apply(stream, &form, n_loc, ld, n_terms, in_order, x, ghosts ? ghost_work : nullptr, beta,
                                   y);
    }
    stk_kron_ell_term terms[3];
    for (int k = 0; k < n_terms; ++k)
        terms[k] = stk_kron_ell_term{t[k].tri, p->ell_vals[t[k].mat], p->ovf_vals[t[k].mat], x, x_lo, x_hi};
    return stk_kron_ell_apply(stream, &p->ell, n_loc, ld, n_terms, terms, beta, y);
}

extern "C" int stk_kron_plan_ghost_apply(stk_kron_plan *p, void *stream, int32_t n_loc, int32_t ld, int32_t n_terms,
                                         const stk_kron_pack_term *t, const double *x, const double *x_lo,
                                         const double *x_hi, double *y)
{
    STK_REQUIRE(p && t && y, "stk_kron_plan_ghost_apply: null pointer");
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_plan_ghost_apply: n_terms=%d not in 1..3", n_terms);
    for (int k = 0; k < n_terms; ++k)
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < p->n_mats, "stk_kron_plan_ghost_apply: term %d names matrix %d of %d",
                    k, t[k].mat, p->n_mats);
    if (!x_lo && !x_hi) return 0;
    if (p->packed) {
        const stk_pack_pattern *form = p->paired && n_loc >= 8 ? &p->pack_pairs : &p->pack;
        return stk_kron_pack_ghost_apply(stream, form, n_loc, ld, n_terms, t, x, x_lo, x_hi, y);
    }
    if (p->explicit_pairs && n_loc >= 24) {
        stk_pack_pattern form;
        if (!explicit_pattern_for(p, n_terms, t, &form)) return 1;
        stk_kron_pack_term in_order[3];
        for (int k = 0; k < n_terms; ++k) in_order[k] = stk_kron_pack_term{t[k].tri, k};
        return stk_kron_pack_ghost_apply(stream, &form, n_loc, ld, n_terms, in_order, x, x_lo, x_hi, y);
    }
    STK_REQUIRE(x, "stk_kron_plan_ghost_apply: the plain form needs x");
    stk_kron_ell_term terms[3];
    for (int k = 0; k < n_terms; ++k)
        terms[k] = stk_kron_ell_term{t[k].tri, p->ell_vals[t[k].mat], p->ovf_vals[t[k].mat], x, x_lo, x_hi};
    return stk_kron_ell_ghost_apply(stream, &p->ell, n_loc, ld, n_terms, terms, y);
}

// The boundary steps from the compact operands (stk_halo_pack_records' records, the received
// rows x_lo / x_hi, interleaved here into ghost_work): the fast form of
// stk_kron_plan_ghost_apply on plans with a packed stream; plans without one take the
// slab form (they need `x`).
extern "C" int stk_kron_plan_boundary_apply(stk_kron_plan *p, void *stream, int32_t n_loc, int32_t ld, int32_t n_terms,
                                            const stk_kron_pack_term *t, const double *x, const double *records,
                                            const double *x_lo, const double *x_hi, double *ghost_work, double *y)
{
    STK_REQUIRE(p && t && y, "stk_kron_plan_boundary_apply: null pointer");
    STK_REQUIRE(n_terms >= 1 && n_terms <= 3, "stk_kron_plan_boundary_apply: n_terms=%d not in 1..3", n_terms);
    for (int k = 0; k < n_terms; ++k)
        STK_REQUIRE(t[k].mat >= 0 && t[k].mat < p->n_mats,
                    "stk_kron_plan_boundary_apply: term %d names matrix %d of %d", k, t[k].mat, p->n_mats);
    if (!x_lo && !x_hi) return 0;
    const bool pairs_explicit = !p->packed && p->explicit_pairs && n_loc >= 24;
    if (!records || !(p->packed || pairs_explicit))
        return stk_kron_plan_ghost_apply(p, stream, n_loc, ld, n_terms, t, x, x_lo, x_hi, y);
    STK_REQUIRE(ghost_work, "stk_kron_plan_boundary_apply: the received rows need ghost_work (2*M doubles)");
    int rc = stk_interleave_ghosts(stream, p->M, x_lo, x_hi, ghost_work);
    if (rc) return rc;
    if (p->packed) {
        const stk_pack_pattern *form = p->paired && n_loc >= 8 ? &p->pack_pairs : &p->pack;
        return stk_kron_pack_boundary_apply(stream, form, n_loc, ld, n_terms, t, records, ghost_work, x_lo != nullptr,
                                            x_hi != nullptr, y);
    }
    stk_pack_pattern form;
    if (!explicit_pattern_for(p, n_terms, t, &form)) return 1;
    stk_kron_pack_term in_order[3];
    for (int k = 0; k < n_terms; ++k) in_order[k] = stk_kron_pack_term{t[k].tri, k};
    return stk_kron_pack_boundary_apply(stream, &form, n_loc, ld, n_terms, in_order, records, ghost_work,
                                        x_lo != nullptr, x_hi != nullptr, y);
}
