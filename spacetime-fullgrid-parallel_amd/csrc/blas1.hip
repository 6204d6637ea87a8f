// BLAS-1 kernels on flat fp64 arrays: the arithmetic of KronVectorMPI
// (reference source/mpi_vector.py:84-122) and the local part of dot (:205-210).
// All are HBM-bound streaming kernels: 16 B per lane per access, grid-stride.
#include "stk_common.h"

namespace {

constexpr int BS = 256;
constexpr int DOT_BLOCKS = 2048;

__global__ __launch_bounds__(BS) void axpbyz_kernel(int64_t n, double a, const double *__restrict__ x,
                                                    double b, const double *y, double *z)
{
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * BS;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    const double2 *y2 = reinterpret_cast<const double2 *>(y);
    double2 *z2 = reinterpret_cast<double2 *>(z);
    if (b == 0.0) {
        for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += stride) {
            double2 xv = x2[i];
            z2[i] = make_double2(a * xv.x, a * xv.y);
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) z[n - 1] = a * x[n - 1];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += stride) {
            double2 xv = x2[i], yv = y2[i];
            // b * y rounded, then ONE fused multiply-add: with a = 1 this is exactly the
            // two steps `y *= b; y += x` (reference linalg.py:39-40), which lets PCG update
            // its search direction in one pass without changing a bit (written out so that
            // it does not depend on how the compiler contracts a * x + b * y)
            z2[i] = make_double2(fma(a, xv.x, b * yv.x), fma(a, xv.y, b * yv.y));
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) z[n - 1] = fma(a, x[n - 1], b * y[n - 1]);
    }
}

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sums one value per thread over the block; result valid in thread 0.
__device__ inline double block_sum(double v, double *sm)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) sm[wid] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) r += sm[w];
    }
    return r;
}

__global__ __launch_bounds__(BS) void dot_partial_kernel(int64_t n, const double *__restrict__ x,
                                                         const double *__restrict__ y, double *partial)
{
    __shared__ double sm[BS / 64];
    const int64_t n2 = n >> 1;
    const int64_t stride = (int64_t)gridDim.x * BS;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    const double2 *y2 = reinterpret_cast<const double2 *>(y);
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < n2; i += stride) {
        double2 xv = x2[i], yv = y2[i];
        acc0 += xv.x * yv.x;
        acc1 += xv.y * yv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc0 += x[n - 1] * y[n - 1];
    double r = block_sum(acc0 + acc1, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// One block: fixed-order sum of the partials, so the result does not depend
// on scheduling (bitwise reproducible run to run).
__global__ __launch_bounds__(1024) void dot_final_kernel(int n_partial, const double *partial, double *out)
{
    __shared__ double sm[1024 / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 1024) acc += partial[i];
    double r = block_sum(acc, sm);
    if (threadIdx.x == 0) out[0] = r;
}

// Time slices of a slab: y[i][k] = x[i][cols[k]] (gather; columns k >= n_cols of y,
// its padding, are written as zero) or y[i][cols[k]] = x[i][k] (scatter).
__global__ __launch_bounds__(BS) void slab_columns_kernel(int64_t total, int32_t n_cols, int32_t width,
                                                           const int32_t *__restrict__ cols,
                                                           const double *__restrict__ x, int32_t ld_x, double *y,
                                                           int32_t ld_y, int scatter)
{
    const int64_t stride = (int64_t)gridDim.x * BS;
    for (int64_t idx = (int64_t)blockIdx.x * BS + threadIdx.x; idx < total; idx += stride) {
        const int64_t i = idx / width;
        const int k = (int)(idx - i * width);
        if (scatter)
            y[i * ld_y + cols[k]] = x[i * ld_x + k];
        else
            y[i * ld_y + k] = k < n_cols ? x[i * ld_x + cols[k]] : 0.0;
    }
}

}  // namespace

extern "C" int stk_slab_gather_columns(void *stream, int32_t M, int32_t n_cols, const int32_t *cols,
                                       const double *x, int32_t ld_x, double *y, int32_t ld_y)
{
    const stk_timed timed_(STK_OP_BLAS1, stream);
    STK_REQUIRE(M > 0 && n_cols > 0 && cols && x && y && x != y, "stk_slab_gather_columns: bad arguments");
    STK_REQUIRE(ld_y >= n_cols && ld_x >= 1, "stk_slab_gather_columns: ld_y=%d is smaller than n_cols=%d", ld_y,
                n_cols);
    const int64_t total = (int64_t)M * ld_y;
    hipLaunchKernelGGL(slab_columns_kernel, dim3(stk_flat_grid(total, BS)), dim3(BS), 0, stk_stream(stream), total,
                       n_cols, ld_y, cols, x, ld_x, y, ld_y, 0);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_slab_scatter_columns(void *stream, int32_t M, int32_t n_cols, const int32_t *cols,
                                        const double *x, int32_t ld_x, double *y, int32_t ld_y)
{
    const stk_timed timed_(STK_OP_BLAS1, stream);
    STK_REQUIRE(M > 0 && n_cols > 0 && cols && x && y && x != y, "stk_slab_scatter_columns: bad arguments");
    STK_REQUIRE(ld_x >= n_cols && ld_y >= 1, "stk_slab_scatter_columns: ld_x=%d is smaller than n_cols=%d", ld_x,
                n_cols);
    const int64_t total = (int64_t)M * n_cols;
    hipLaunchKernelGGL(slab_columns_kernel, dim3(stk_flat_grid(total, BS)), dim3(BS), 0, stk_stream(stream), total,
                       n_cols, n_cols, cols, x, ld_x, y, ld_y, 1);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_axpbyz(void *stream, int64_t n, double a, const double *x, double b, const double *y,
                          double *z)
{
    const stk_timed timed_(STK_OP_BLAS1, stream);
    if (n <= 0) return 0;
    STK_REQUIRE(x && z && (b == 0.0 || y), "stk_axpbyz: null pointer");
    STK_REQUIRE((((uintptr_t)x | (uintptr_t)z | (uintptr_t)(b == 0.0 ? z : y)) & 15) == 0,
                "stk_axpbyz: arrays must be 16-byte aligned");
    unsigned grid = stk_flat_grid(n / 2 + 1, BS);
    hipLaunchKernelGGL(axpbyz_kernel, dim3(grid), dim3(BS), 0, stk_stream(stream), n, a, x, b, y, z);
    STK_LAUNCH_CHECK();
    return 0;
}

extern "C" int stk_axpby(void *stream, int64_t n, double a, const double *x, double b, double *y)
{
    return stk_axpbyz(stream, n, a, x, b, y, y);
}

extern "C" int64_t stk_dot_work_size(void) { return DOT_BLOCKS; }

extern "C" int stk_dot(void *stream, int64_t n, const double *x, const double *y, double *work, double *out)
{
    const stk_timed timed_(STK_OP_BLAS1, stream);
    STK_REQUIRE(work && out, "stk_dot: null work/out");
    STK_REQUIRE(n == 0 || (x && y), "stk_dot: null input");
    STK_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "stk_dot: arrays must be 16-byte aligned");
    unsigned grid = stk_flat_grid(n / 2 + 1, BS);
    if (grid > DOT_BLOCKS) grid = DOT_BLOCKS;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(grid), dim3(BS), 0, stk_stream(stream), n, x, y, work);
    STK_LAUNCH_CHECK();
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(1024), 0, stk_stream(stream), (int)grid, work, out);
    STK_LAUNCH_CHECK();
    return 0;
}
