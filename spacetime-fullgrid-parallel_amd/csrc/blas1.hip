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

// ---- inner product of two slabs, one partial per TIME STEP -----------------------
// KronVectorMPI.dot (reference mpi_vector.py:205-210) sums the local products and
// all-reduces one number, so the order of the additions -- and the last digits of
// every r.Pr of a solve -- depend on how many ranks share the time axis (the
// reference against itself on 8 ranks: 4.6e-11 in the history).  Here the sum over
// the SPATIAL index of every time step has one fixed shape that depends on M alone:
//   rows in blocks of SD_ROWS; inside a block, row i goes to class i mod SD_CLASSES,
//   a class adds its products in increasing i with fused multiply-adds from zero;
//   the SD_CLASSES class sums of a block are added pairwise (0+1, 2+3, ...; then
//   pairs of pairs; ...); the block sums of a time step are added by 256 partial
//   sums (block b to partial b mod 256, increasing b) that meet in the fixed
//   wavefront / workgroup tree of block_sum.
// The n_loc results land at the global positions of the slab's time steps in an
// N-vector whose other entries are written as zero; the caller all-reduces that
// (adding zeros is exact in any order) and adds the N entries in increasing t
// (stk_sum_steps): the value is the same on 1, 2, 4 or 8 ranks, bit for bit.
constexpr int SD_ROWS = 256;
constexpr int SD_CLASSES = 8;
constexpr int SD_ITEMS = 4;  // (class, pair of steps) items per thread at most

// One virtual group of G = SD_CLASSES * ld2 items works on one block of rows: item f
// = class * ld2 + pair reads 16 bytes at (row, pair) -- the items of a group read
// SD_CLASSES consecutive rows, i.e. one contiguous run of SD_CLASSES * ld * 8 bytes
// per step.  A workgroup holds as many groups as fit (short slabs) or walks the
// items of one group in SD_ITEMS rounds (long slabs); p_lo / p_hi cut slabs of more
// pairs than that into column ranges.
template <int BSZ>
__global__ __launch_bounds__(BSZ) void slab_dot_blocks_kernel(int32_t M, int32_t ld2, int32_t p_lo, int32_t p_hi,
                                                               int32_t groups, int32_t n_blocks,
                                                               const double2 *__restrict__ x,
                                                               const double2 *__restrict__ y,
                                                               double2 *__restrict__ part)
{
    __shared__ double2 sm[BSZ * SD_ITEMS];
    const int np = p_hi - p_lo;             // pairs handled by this launch
    const int G = SD_CLASSES * np;          // items of a virtual group
    const int tid = threadIdx.x;
    const int grp = groups > 1 ? tid / G : 0;
    const int f0 = groups > 1 ? tid - grp * G : tid;
    const int blk = (int)blockIdx.x * groups + grp;
    const bool live = grp < groups && blk < n_blocks;
    const int row0 = blk * SD_ROWS;
    const int row_end = min(row0 + SD_ROWS, M);
    double2 acc[SD_ITEMS];
#pragma unroll
    for (int q = 0; q < SD_ITEMS; ++q) {
        acc[q] = make_double2(0.0, 0.0);
        const int f = f0 + q * BSZ;
        if (!live || f >= G) continue;
        const int cls = f / np, pr = p_lo + (f - cls * np);
        double2 a = make_double2(0.0, 0.0);
        size_t o = (size_t)(row0 + cls) * ld2 + pr;
        const size_t step = (size_t)SD_CLASSES * ld2;
#pragma unroll 4
        for (int i = row0 + cls; i < row_end; i += SD_CLASSES, o += step) {
            const double2 xv = x[o], yv = y[o];
            a.x = fma(xv.x, yv.x, a.x);
            a.y = fma(xv.y, yv.y, a.y);
        }
        acc[q] = a;
    }
    // class sums -> LDS [group][item], then the pairwise tree over the classes
#pragma unroll
    for (int q = 0; q < SD_ITEMS; ++q) {
        const int f = f0 + q * BSZ;
        if (live && f < G) sm[grp * G + f] = acc[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SD_ITEMS; ++q) {
        const int f = f0 + q * BSZ;
        if (!live || f >= np) continue;  // items of class 0 finish their pair of steps
        const double2 *c = sm + grp * G + f;
        double2 v[SD_CLASSES];
#pragma unroll
        for (int g = 0; g < SD_CLASSES; ++g) v[g] = c[g * np];
#pragma unroll
        for (int w = 1; w < SD_CLASSES; w <<= 1)
#pragma unroll
            for (int g = 0; g < SD_CLASSES; g += 2 * w) {
                v[g].x += v[g + w].x;
                v[g].y += v[g + w].y;
            }
        part[(size_t)(p_lo + f) * n_blocks + blk] = v[0];  // [pair][block]
    }
}

// One workgroup per pair of time steps: the block sums in a fixed order, written to
// out[t_begin + t]; workgroup 0 zeroes the entries of the other ranks' steps.
__global__ __launch_bounds__(256) void slab_dot_steps_kernel(int32_t n_loc, int32_t n_blocks,
                                                             const double2 *__restrict__ part, int32_t N,
                                                             int32_t t_begin, double *out_all)
{
    __shared__ double sm[2][256 / 64];
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < N; t += 256)
            if (t < t_begin || t >= t_begin + n_loc) out_all[t] = 0.0;
    double *out = out_all + t_begin;
    const double2 *c = part + (size_t)blockIdx.x * n_blocks;
    double2 a = make_double2(0.0, 0.0);
    for (int b = threadIdx.x; b < n_blocks; b += 256) {
        a.x += c[b].x;
        a.y += c[b].y;
    }
    const double r0 = block_sum(a.x, sm[0]), r1 = block_sum(a.y, sm[1]);
    if (threadIdx.x == 0) {
        const int t = 2 * (int)blockIdx.x;
        out[t] = r0;
        if (t + 1 < n_loc) out[t + 1] = r1;
    }
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

extern "C" int64_t stk_slab_dot_work_size(int32_t M, int32_t n_loc)
{
    if (M <= 0 || n_loc <= 0) return 0;
    const int64_t n_blocks = ((int64_t)M + SD_ROWS - 1) / SD_ROWS;
    return 2 * (int64_t)((n_loc + 1) / 2) * n_blocks;
}

extern "C" double stk_sum_steps(const double *steps_host, int32_t N)
{
    double s = 0.0;
    for (int t = 0; t < N; ++t) s += steps_host[t];
    return s;
}

extern "C" int stk_slab_dot(void *stream, int32_t M, int32_t n_loc, int32_t ld, const double *x, const double *y,
                            double *work, int32_t N, int32_t t_begin, double *out_steps)
{
    const stk_timed timed_(STK_OP_BLAS1, stream);
    STK_REQUIRE(M > 0 && n_loc > 0 && ld >= n_loc && (ld & 1) == 0,
                "stk_slab_dot: bad sizes M=%d n_loc=%d ld=%d (ld must be even)", M, n_loc, ld);
    STK_REQUIRE(x && y && work && out_steps, "stk_slab_dot: null pointer");
    STK_REQUIRE(t_begin >= 0 && t_begin + n_loc <= N, "stk_slab_dot: steps [%d, %d) of %d", t_begin, t_begin + n_loc, N);
    STK_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)work) & 15) == 0,
                "stk_slab_dot: x, y and work must be 16-byte aligned");
    hipStream_t st = stk_stream(stream);
    const int ld2 = ld / 2, np_all = (n_loc + 1) / 2;
    const int n_blocks = (M + SD_ROWS - 1) / SD_ROWS;
    const double2 *x2 = reinterpret_cast<const double2 *>(x), *y2 = reinterpret_cast<const double2 *>(y);
    double2 *part = reinterpret_cast<double2 *>(work);
    // pairs per launch: what SD_ITEMS rounds of a 256-thread workgroup hold
    constexpr int BSZ = 256;
    const int np_max = BSZ * SD_ITEMS / SD_CLASSES;
    for (int p_lo = 0; p_lo < np_all; p_lo += np_max) {
        const int p_hi = p_lo + np_max < np_all ? p_lo + np_max : np_all;
        const int G = SD_CLASSES * (p_hi - p_lo);
        const int groups = G <= BSZ ? BSZ / G : 1;
        const unsigned grid = (unsigned)((n_blocks + groups - 1) / groups);
        hipLaunchKernelGGL((slab_dot_blocks_kernel<BSZ>), dim3(grid), dim3(BSZ), 0, st, M, ld2, p_lo, p_hi, groups,
                           n_blocks, x2, y2, part);
        STK_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(slab_dot_steps_kernel, dim3(np_all), dim3(256), 0, st, n_loc, n_blocks, part, N, t_begin,
                       out_steps);
    STK_LAUNCH_CHECK();
    return 0;
}
