// Plan construction on the device (the set-up that
// the reference does with SciPy on the host, multigrid.py:130-166).
//
//  * stk_ell_from_csr: a sliced-ELL copy of a CSR matrix that already lives on the
//    device -- rows listed in a processing order, K slots per row, optionally
//    without the diagonal (Gauss-Seidel copies, stk_ell_rows.diag_free), optionally
//    only the entries towards rows of an earlier dependency group (the zero-start
//    copies of csrc/mg.hip).  One hierarchy needs some thirty such copies; built
//    with NumPy fancy indexing on million-row arrays they were a third of the
//    set-up time.
//  * stk_csr_galerkin: the Galerkin product R A P (multigrid.py:142-145) row by row
//    on the device, every sum accumulated in the order SciPy's csr_matmat
//    accumulates it -- (R A) first, its rows traversed in the order that product
//    emits them, then times P -- with separately rounded multiply and add, so that
//    the coarse matrices are BIT FOR BIT those of the reference's `R @ A @ P`
//    (the solve's history is sensitive to their last bit).
#include "stk_common.h"

namespace {

__global__ __launch_bounds__(256) void ell_from_csr_kernel(int32_t n_pos, int32_t K, const int32_t *__restrict__ order,
                                                           const int32_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices,
                                                           const double *__restrict__ va, const double *__restrict__ vm,
                                                           int32_t strip_diag, const int32_t *__restrict__ grp,
                                                           int32_t pad_col, int32_t *__restrict__ idx_out,
                                                           double *__restrict__ va_out, double *__restrict__ vm_out,
                                                           double *__restrict__ dia_a, double *__restrict__ dia_m,
                                                           int32_t *__restrict__ overflow)
{
    const int stride = gridDim.x * 256;
    for (int pos = blockIdx.x * 256 + threadIdx.x; pos < n_pos; pos += stride) {
        const int row = order ? order[pos] : pos;
        const int e0 = indptr[row], e1 = indptr[row + 1];
        const int g_row = grp ? grp[row] : 0;
        int s = 0;
        double da = 0.0, dm = 0.0;
        for (int e = e0; e < e1; ++e) {
            const int col = indices[e];
            if (col == row) {
                da = va[e];
                if (vm) dm = vm[e];
                if (strip_diag) continue;
            }
            if (grp && !(grp[col] < g_row)) continue;
            if (s < K) {
                idx_out[(size_t)pos * K + s] = col;
                va_out[(size_t)pos * K + s] = va[e];
                if (vm_out) vm_out[(size_t)pos * K + s] = vm[e];
            }
            ++s;
        }
        if (s > K) atomicMax(overflow, s);
        // pad_col < 0: a column this row is known to be allowed to read -- its first
        // kept entry (written before the row by any order of the sweep that
        // respects the dependencies), or the row itself if it keeps none
        const int32_t pad = pad_col >= 0 ? pad_col : (s > 0 ? idx_out[(size_t)pos * K] : row);
        for (; s < K; ++s) {
            idx_out[(size_t)pos * K + s] = pad;
            va_out[(size_t)pos * K + s] = 0.0;
            if (vm_out) vm_out[(size_t)pos * K + s] = 0.0;
        }
        if (dia_a) dia_a[pos] = da;
        if (dia_m) dia_m[pos] = dm;
    }
}

// One relaxation of the depth of every row in the dependency DAG of a Gauss-Seidel
// sweep in dof order (row i waits for its neighbours j < i forward, j > i
// backward): depth[i] = max over those neighbours of depth[j] + 1.  The host
// repeats it until nothing changes (the DAG of a P1 matrix is 3-8 levels deep).
__global__ __launch_bounds__(256) void gs_depth_kernel(int32_t n, const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices, int32_t backward,
                                                       const int32_t *__restrict__ depth_in,
                                                       int32_t *__restrict__ depth_out, int32_t *__restrict__ changed)
{
    const int stride = gridDim.x * 256;
    bool any = false;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        int d = 0;
        for (int e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int j = indices[e];
            if (backward ? j > i : j < i) d = max(d, depth_in[j] + 1);
        }
        depth_out[i] = d;
        any |= d != depth_in[i];
    }
    if (any) *changed = 1;
}

constexpr int CAP = 64;  // entries of one row of R A / R A P held by a thread

// One thread per coarse row: t = R[i, :] A (first-touch list like csr_matmat's
// linked list), then c = t P.  SciPy emits a row of a product in REVERSE order of
// first touch and traverses it in that order in the next product.
__global__ __launch_bounds__(64) void galerkin_kernel(int32_t nc, const int32_t *__restrict__ r_ptr,
                                                      const int32_t *__restrict__ r_idx, const double *__restrict__ r_val,
                                                      const int32_t *__restrict__ a_ptr, const int32_t *__restrict__ a_idx,
                                                      const double *__restrict__ a_val, const int32_t *__restrict__ p_ptr,
                                                      const int32_t *__restrict__ p_idx, const double *__restrict__ p_val,
                                                      int32_t cap_out, int32_t *__restrict__ out_count,
                                                      int32_t *__restrict__ out_idx, double *__restrict__ out_val,
                                                      int32_t *__restrict__ overflow)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nc) return;
    int32_t t_col[CAP];
    double t_val[CAP];
    int nt = 0;
    for (int jj = r_ptr[i]; jj < r_ptr[i + 1]; ++jj) {
        const int j = r_idx[jj];
        const double v = r_val[jj];
        for (int kk = a_ptr[j]; kk < a_ptr[j + 1]; ++kk) {
            const int k = a_idx[kk];
            const double prod = __dmul_rn(v, a_val[kk]);
            int q = 0;
            while (q < nt && t_col[q] != k) ++q;
            if (q == nt) {
                if (nt == CAP) {
                    atomicMax(overflow, CAP + 1);
                    return;
                }
                t_col[nt] = k, t_val[nt] = 0.0, ++nt;
            }
            t_val[q] = __dadd_rn(t_val[q], prod);
        }
    }
    int32_t c_col[CAP];
    double c_val[CAP];
    int ncol = 0;
    if (p_ptr == nullptr) {  // C = R A alone (the restricted-residual product of mg.hip)
        for (int q = 0; q < nt; ++q) c_col[q] = t_col[q], c_val[q] = t_val[q];
        ncol = nt;
        nt = 0;
    }
    for (int q = nt - 1; q >= 0; --q) {  // the emitted order of the row of R A
        const int l = t_col[q];
        const double v = t_val[q];
        for (int kk = p_ptr[l]; kk < p_ptr[l + 1]; ++kk) {
            const int k = p_idx[kk];
            const double prod = __dmul_rn(v, p_val[kk]);
            int z = 0;
            while (z < ncol && c_col[z] != k) ++z;
            if (z == ncol) {
                if (ncol == CAP) {
                    atomicMax(overflow, CAP + 1);
                    return;
                }
                c_col[ncol] = k, c_val[ncol] = 0.0, ++ncol;
            }
            c_val[z] = __dadd_rn(c_val[z], prod);
        }
    }
    if (ncol > cap_out) {
        atomicMax(overflow, ncol);
        return;
    }
    // ascending columns (what sort_indices gives): insertion sort of a short row
    for (int a = 1; a < ncol; ++a) {
        const int32_t c = c_col[a];
        const double v = c_val[a];
        int b = a - 1;
        for (; b >= 0 && c_col[b] > c; --b) c_col[b + 1] = c_col[b], c_val[b + 1] = c_val[b];
        c_col[b + 1] = c, c_val[b + 1] = v;
    }
    // csr_matmat emits an entry only if its sum is not exactly zero
    int n_out = 0;
    for (int a = 0; a < ncol; ++a) {
        if (c_val[a] == 0.0) continue;
        out_idx[(size_t)i * cap_out + n_out] = c_col[a];
        out_val[(size_t)i * cap_out + n_out] = c_val[a];
        ++n_out;
    }
    out_count[i] = n_out;
}

}  // namespace

extern "C" int stk_ell_from_csr(void *stream, int32_t n_pos, int32_t K, const int32_t *order, const int32_t *indptr,
                                const int32_t *indices, const double *va, const double *vm, int32_t strip_diag,
                                const int32_t *grp, int32_t pad_col, int32_t *idx_out, double *va_out, double *vm_out,
                                double *dia_a, double *dia_m, int32_t *overflow)
{
    STK_REQUIRE(n_pos >= 0 && K >= 1 && indptr && indices && va && idx_out && va_out && overflow,
                "stk_ell_from_csr: bad arguments");
    STK_REQUIRE((vm == nullptr) == (vm_out == nullptr), "stk_ell_from_csr: vm and vm_out go together");
    if (n_pos == 0) return 0;
    hipLaunchKernelGGL(ell_from_csr_kernel, dim3(stk_flat_grid(n_pos, 256)), dim3(256), 0, stk_stream(stream), n_pos, K,
                       order, indptr, indices, va, vm, strip_diag, grp, pad_col, idx_out, va_out, vm_out, dia_a, dia_m,
                       overflow);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_csr_galerkin(void *stream, int32_t nc, const int32_t *r_indptr, const int32_t *r_indices,
                                const double *r_data, const int32_t *a_indptr, const int32_t *a_indices,
                                const double *a_data, const int32_t *p_indptr, const int32_t *p_indices,
                                const double *p_data, int32_t cap, int32_t *row_counts, int32_t *out_indices,
                                double *out_data, int32_t *overflow)
{
    STK_REQUIRE(nc > 0 && r_indptr && r_indices && r_data && a_indptr && a_indices && a_data && row_counts &&
                    out_indices && out_data && overflow,
                "stk_csr_galerkin: null pointer");
    STK_REQUIRE((p_indptr == nullptr) == (p_indices == nullptr) && (p_indptr == nullptr) == (p_data == nullptr),
                "stk_csr_galerkin: P is given whole or not at all");
    STK_REQUIRE(cap >= 1 && cap <= CAP, "stk_csr_galerkin: cap=%d not in 1..%d", cap, CAP);
    hipLaunchKernelGGL(galerkin_kernel, dim3((nc + 63) / 64), dim3(64), 0, stk_stream(stream), nc, r_indptr, r_indices,
                       r_data, a_indptr, a_indices, a_data, p_indptr, p_indices, p_data, cap, row_counts, out_indices,
                       out_data, overflow);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_gs_depth_step(void *stream, int32_t n, const int32_t *indptr, const int32_t *indices,
                                 int32_t backward, const int32_t *depth_in, int32_t *depth_out, int32_t *changed)
{
    STK_REQUIRE(n > 0 && indptr && indices && depth_in && depth_out && changed && depth_in != depth_out,
                "stk_gs_depth_step: bad arguments");
    hipLaunchKernelGGL(gs_depth_kernel, dim3(stk_flat_grid(n, 256)), dim3(256), 0, stk_stream(stream), n, indptr,
                       indices, backward, depth_in, depth_out, changed);
    STK_LAUNCH_CHECK();
    return 0;
}
