// Slab storage and the byte-moving steps around it: what a host WITHOUT PyTorch
// needs to own a KronVectorMPI slab (reference source/mpi_vector.py:62-71,
// 124-138: X_loc allocation, scatter / gather), and the kernels that used to be
// PyTorch strided copies on the product path:
//
//  * stk_transpose: time-major (N_loc, M) <-> space-major slab (M, ld) and the
//    block transposes of KronVectorMPI.permute (mpi_vector.py:212-240), one tiled
//    kernel through LDS, both sides coalesced;
//  * stk_halo_pack: the first and the last time step of a slab -- the rows the
//    neighbour ranks receive as X_loc_bdr[-1] / X_loc_bdr[0]
//    (mpi_vector.py:148-167) -- read ONCE from the slab's lines and written with a
//    caller-chosen element stride: stride 1 = contiguous send buffers for RCCL,
//    stride 2 = straight into an interleaved ghost buffer gh[j] = (lo, hi)
//    (kron_pack.hip), the local one or a neighbour's mapped by stk_ipc_open;
//  * stk_outer: the right-hand side u0_t kron u0_x (heateq_mpi.py:189-191).
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "stk_common.h"

namespace {

constexpr int TILE = 32;

// dst[c * ld_dst + r] = src[r * ld_src + c], r < rows, c < cols; the columns
// rows .. zero_to-1 of dst (slab padding) are written as zero
__global__ __launch_bounds__(TILE * 8) void transpose_kernel(int32_t rows, int32_t cols, const double *__restrict__ src,
                                                             int64_t ld_src, double *__restrict__ dst, int64_t ld_dst,
                                                             int32_t zero_to)
{
    __shared__ double tile[TILE][TILE + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tiles_c = (cols + TILE - 1) / TILE;
    const int tiles_r = (max(rows, zero_to) + TILE - 1) / TILE;
    const int64_t n_tiles = (int64_t)tiles_c * tiles_r;
    for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int r0 = (int)(tl % tiles_r) * TILE, c0 = (int)(tl / tiles_r) * TILE;
#pragma unroll
        for (int k = 0; k < TILE; k += 8) {
            const int r = r0 + ty + k, c = c0 + tx;
            tile[ty + k][tx] = (r < rows && c < cols) ? src[(int64_t)r * ld_src + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TILE; k += 8) {
            const int c = c0 + ty + k, r = r0 + tx;
            if (c < cols && (r < rows || r < zero_to)) dst[(int64_t)c * ld_dst + r] = tile[tx][ty + k];
        }
        __syncthreads();
    }
}

// first[j * s_first] = x[j][0], last[j * s_last] = x[j][n_loc - 1]; one thread per
// spatial dof, both steps of a row from the same pass over its lines
__global__ __launch_bounds__(256) void halo_pack_kernel(int32_t M, int32_t n_loc, int32_t ld,
                                                        const double *__restrict__ x, double *first, int32_t s_first,
                                                        double *last, int32_t s_last)
{
    const int stride = gridDim.x * 256;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += stride) {
        const double *row = x + (size_t)j * ld;
        if (first) first[(size_t)j * s_first] = row[0];
        if (last) last[(size_t)j * s_last] = row[n_loc - 1];
    }
}

// ... and the four entries of every row that the boundary steps of a Kronecker apply read once the halo is
// there (stk_kron_pack_boundary_apply): rec[j] = (x[j][0], x[j][1], x[j][n_loc-2], x[j][n_loc-1]), from the same
// two lines of the row (slabs of one step repeat it)
__global__ __launch_bounds__(256) void halo_pack_records_kernel(int32_t M, int32_t n_loc, int32_t ld,
                                                                const double *__restrict__ x, double *first,
                                                                int32_t s_first, double *last, int32_t s_last,
                                                                double4 *__restrict__ rec)
{
    const int stride = gridDim.x * 256;
    const int t1 = n_loc > 1 ? 1 : 0, t2 = n_loc > 1 ? n_loc - 2 : 0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += stride) {
        const double *row = x + (size_t)j * ld;
        const double a = row[0], b = row[t1], c = row[t2], d = row[n_loc - 1];
        if (first) first[(size_t)j * s_first] = a;
        if (last) last[(size_t)j * s_last] = d;
        rec[j] = make_double4(a, b, c, d);
    }
}

// out[k * ld_out + j] = x[j][t_idx[k]]: the time rows `communicate_dofs` sends
// (mpi_vector.py:189-203), all of them from one pass over the slab
__global__ __launch_bounds__(256) void extract_rows_kernel(int32_t M, int32_t n_rows, const int32_t *__restrict__ t_idx,
                                                           const double *__restrict__ x, int32_t ld,
                                                           double *__restrict__ out, int64_t ld_out)
{
    const int stride = gridDim.x * 256;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < M; j += stride) {
        const double *row = x + (size_t)j * ld;
        for (int k = 0; k < n_rows; ++k) out[(size_t)k * ld_out + j] = row[t_idx[k]];
    }
}

// dst[r * ld_dst + c] = src[r * ld_src + c]: re-striding of a block of rows (the
// pieces of the all-to-all transposes around the wavelet kernel)
__global__ __launch_bounds__(256) void copy_block_kernel(int64_t total, int32_t cols, const double *__restrict__ src,
                                                         int64_t ld_src, double *__restrict__ dst, int64_t ld_dst)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        const int64_t r = idx / cols;
        const int c = (int)(idx - r * cols);
        dst[r * ld_dst + c] = src[r * ld_src + c];
    }
}

__global__ __launch_bounds__(256) void outer_kernel(int64_t total, int32_t n_loc, int32_t ld,
                                                    const double *__restrict__ u_t, const double *__restrict__ u_x,
                                                    double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        const int64_t i = idx / ld;
        const int t = (int)(idx - i * ld);
        y[idx] = t < n_loc ? u_t[t] * u_x[i] : 0.0;
    }
}

// one staging buffer per process for host <-> slab transfers, grown on demand
std::mutex g_stage_mutex;
double *g_stage = nullptr;
size_t g_stage_doubles = 0;

int stage(size_t doubles, double **out)
{
    if (doubles > g_stage_doubles) {
        if (g_stage) (void)hipFree(g_stage);
        g_stage = nullptr, g_stage_doubles = 0;
        STK_HIP(hipMalloc((void **)&g_stage, doubles * sizeof(double)));
        g_stage_doubles = doubles;
    }
    *out = g_stage;
    return 0;
}

int launch_transpose(hipStream_t st, int32_t rows, int32_t cols, const double *src, int64_t ld_src, double *dst,
                     int64_t ld_dst, int32_t zero_to)
{
    const int64_t tiles = (int64_t)((cols + TILE - 1) / TILE) * ((max(rows, zero_to) + TILE - 1) / TILE);
    const unsigned grid = (unsigned)(tiles < 256 * 32 ? (tiles < 1 ? 1 : tiles) : 256 * 32);
    hipLaunchKernelGGL(transpose_kernel, dim3(grid), dim3(TILE, 8), 0, st, rows, cols, src, ld_src, dst, ld_dst,
                       zero_to);
    STK_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int stk_slab_ld(int32_t n_loc) { return n_loc + (n_loc & 1); }

extern "C" int stk_slab_alloc(int32_t M, int32_t n_loc, int32_t *ld, double **slab)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld && slab, "stk_slab_alloc: bad arguments");
    *ld = stk_slab_ld(n_loc);
    const size_t bytes = sizeof(double) * (size_t)M * (size_t)*ld;
    STK_HIP(hipMalloc((void **)slab, bytes));
    STK_HIP(hipMemset(*slab, 0, bytes));
    return 0;
}

extern "C" int stk_slab_free(double *slab)
{
    if (slab) STK_HIP(hipFree(slab));
    return 0;
}

extern "C" int stk_transpose(void *stream, int32_t rows, int32_t cols, const double *src, int64_t ld_src, double *dst,
                             int64_t ld_dst, int32_t zero_to)
{
    STK_REQUIRE(rows > 0 && cols > 0 && src && dst && src != dst, "stk_transpose: bad arguments");
    STK_REQUIRE(ld_src >= cols && ld_dst >= rows && ld_dst >= zero_to, "stk_transpose: leading dimensions too small");
    return launch_transpose(stk_stream(stream), rows, cols, src, ld_src, dst, ld_dst, zero_to);
}

extern "C" int stk_slab_upload(void *stream, int32_t M, int32_t n_loc, int32_t ld, const double *x_host, double *slab)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc && x_host && slab, "stk_slab_upload: bad arguments");
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    hipStream_t st = stk_stream(stream);
    double *tmp = nullptr;
    if (int rc = stage((size_t)M * n_loc, &tmp)) return rc;
    // the staging buffer is shared: what the stream still does with it must be over
    STK_HIP(hipStreamSynchronize(st));
    STK_HIP(hipMemcpyAsync(tmp, x_host, sizeof(double) * (size_t)M * n_loc, hipMemcpyHostToDevice, st));
    if (int rc = launch_transpose(st, n_loc, M, tmp, M, slab, ld, ld)) return rc;
    STK_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" int stk_slab_download(void *stream, int32_t M, int32_t n_loc, int32_t ld, const double *slab, double *x_host)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc && x_host && slab, "stk_slab_download: bad arguments");
    std::lock_guard<std::mutex> lock(g_stage_mutex);
    hipStream_t st = stk_stream(stream);
    double *tmp = nullptr;
    if (int rc = stage((size_t)M * n_loc, &tmp)) return rc;
    STK_HIP(hipStreamSynchronize(st));
    if (int rc = launch_transpose(st, M, n_loc, slab, ld, tmp, M, 0)) return rc;
    STK_HIP(hipMemcpyAsync(x_host, tmp, sizeof(double) * (size_t)M * n_loc, hipMemcpyDeviceToHost, st));
    STK_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" int stk_halo_pack(void *stream, int32_t M, int32_t n_loc, int32_t ld, const double *x, double *first,
                             int32_t stride_first, double *last, int32_t stride_last)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc && x, "stk_halo_pack: bad arguments");
    STK_REQUIRE((first || last) && (!first || stride_first >= 1) && (!last || stride_last >= 1),
                "stk_halo_pack: no destination, or a stride below 1");
    hipLaunchKernelGGL(halo_pack_kernel, dim3(stk_flat_grid(M, 256)), dim3(256), 0, stk_stream(stream), M, n_loc, ld,
                       x, first, stride_first, last, stride_last);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_halo_pack_records(void *stream, int32_t M, int32_t n_loc, int32_t ld, const double *x, double *first,
                                     int32_t stride_first, double *last, int32_t stride_last, double *records)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc && x && records, "stk_halo_pack_records: bad arguments");
    STK_REQUIRE((!first || stride_first >= 1) && (!last || stride_last >= 1), "stk_halo_pack_records: a stride below 1");
    STK_REQUIRE(((uintptr_t)records & 31) == 0, "stk_halo_pack_records: records must be 32-byte aligned");
    hipLaunchKernelGGL(halo_pack_records_kernel, dim3(stk_flat_grid(M, 256)), dim3(256), 0, stk_stream(stream), M,
                       n_loc, ld, x, first, stride_first, last, stride_last, reinterpret_cast<double4 *>(records));
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_outer(void *stream, int32_t M, int32_t n_loc, int32_t ld, const double *u_t, const double *u_x,
                         double *y)
{
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc && u_t && u_x && y, "stk_outer: bad arguments");
    const int64_t total = (int64_t)M * ld;
    hipLaunchKernelGGL(outer_kernel, dim3(stk_flat_grid(total, 256)), dim3(256), 0, stk_stream(stream), total, n_loc,
                       ld, u_t, u_x, y);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_slab_extract_time_rows(void *stream, int32_t M, int32_t n_rows, const int32_t *t_idx,
                                          const double *x, int32_t ld, double *out, int64_t ld_out)
{
    STK_REQUIRE(M > 0 && n_rows > 0 && t_idx && x && out && ld_out >= M, "stk_slab_extract_time_rows: bad arguments");
    hipLaunchKernelGGL(extract_rows_kernel, dim3(stk_flat_grid(M, 256)), dim3(256), 0, stk_stream(stream), M, n_rows,
                       t_idx, x, ld, out, ld_out);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_copy_block(void *stream, int64_t rows, int32_t cols, const double *src, int64_t ld_src, double *dst,
                              int64_t ld_dst)
{
    STK_REQUIRE(rows >= 0 && cols >= 0 && ld_src >= cols && ld_dst >= cols, "stk_copy_block: bad sizes");
    if (rows == 0 || cols == 0) return 0;
    STK_REQUIRE(src && dst, "stk_copy_block: null pointer");
    const int64_t total = rows * cols;
    hipLaunchKernelGGL(copy_block_kernel, dim3(stk_flat_grid(total, 256)), dim3(256), 0, stk_stream(stream), total,
                       cols, src, ld_src, dst, ld_dst);
    STK_LAUNCH_CHECK();
    return 0;
}
