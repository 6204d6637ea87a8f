// Wavelet-in-time transform, whole time axis resident on this GPU.
//
// y = (W_t kron I) x or (W_t^T kron I) x, interleaved numbering, all J levels
// fused: a workgroup stages R time columns (R x N doubles, N = 2^J + 1) in LDS
// with coalesced loads, runs the J levels there and stores the tile once.
// HBM traffic is 16 B per unknown for the whole transform; the reference's
// composite form makes J passes (source/wavelets.py:172-198).
//
// One level (wavelets.py:106-118, 136-169; S = 2^(J-j), s = 2^(j/2), reading
// pre-level values only):
//   W   odd  k: y = 1/2 (x[(k-1)S] + x[(k+1)S]) + s x[kS]
//       even k: y = x[kS] - 1/2 s (x[(k-1)S] + x[(k+1)S]); end nodes: - s x[nb]
//   W^T even k: y = x[kS] + 1/2 (x[(k-1)S] + x[(k+1)S])  (missing nb dropped)
//       odd  k: y = s (x[kS] - cl x[(k-1)S] - cr x[(k+1)S]), c = 1 next to an
//               end node, else 1/2                       (wavelets.py:120-134)
#include <cmath>
#include <cstring>

#include "stk_common.h"

namespace {

constexpr int WBS = 256;
constexpr int TILE_ELEMS = 4160;                           // 64 columns of 65
constexpr int MAXI = (TILE_ELEMS + WBS - 1) / WBS;         // outputs per thread per level

template <bool TRANSPOSED>
__global__ __launch_bounds__(WBS) void wavelet_kernel(int32_t M, int32_t J, int32_t ld, int32_t R,
                                                      const double *x, double *y)
{
    extern __shared__ double tile[];  // [R][N], N odd => column-per-lane access is conflict free
    const int N = (1 << J) + 1;
    const int row0 = blockIdx.x * R;
    const int nrows = min(R, M - row0);
    const int tid = threadIdx.x;

    for (int k = tid; k < nrows * N; k += WBS) {
        const int r = k / N, t = k - r * N;
        tile[k] = x[(size_t)(row0 + r) * ld + t];
    }
    __syncthreads();

    for (int step = 0; step < J; ++step) {
        const int j = TRANSPOSED ? (J - step) : (step + 1);
        const int S = 1 << (J - j);
        const int n = 1 << j;  // nodes k = 0..n
        const double s = exp2(0.5 * j);
        const int items = nrows * (n + 1);
        double out[MAXI];
#pragma unroll
        for (int q = 0; q < MAXI; ++q) {
            const int it = tid + q * WBS;
            if (it < items) {
                const int k = it / nrows, r = it - k * nrows;  // lanes walk the columns
                const double *c = tile + r * N;
                const double xc = c[k * S];
                const double xl = (k > 0) ? c[(k - 1) * S] : 0.0;
                const double xr = (k < n) ? c[(k + 1) * S] : 0.0;
                double v;
                if (!TRANSPOSED) {
                    if (k & 1)
                        v = 0.5 * (xl + xr) + s * xc;
                    else if (k == 0)
                        v = xc - s * xr;
                    else if (k == n)
                        v = xc - s * xl;
                    else
                        v = xc - 0.5 * s * (xl + xr);
                } else {
                    if (k & 1) {
                        const double cl = (k - 1 == 0) ? 1.0 : 0.5;
                        const double cr = (k + 1 == n) ? 1.0 : 0.5;
                        v = s * (xc - cl * xl - cr * xr);
                    } else {
                        v = xc + 0.5 * xl + 0.5 * xr;
                    }
                }
                out[q] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXI; ++q) {
            const int it = tid + q * WBS;
            if (it < items) {
                const int k = it / nrows, r = it - k * nrows;
                tile[r * N + k * S] = out[q];
            }
        }
        __syncthreads();
    }

    for (int k = tid; k < nrows * N; k += WBS) {
        const int r = k / N, t = k - r * N;
        y[(size_t)(row0 + r) * ld + t] = tile[k];
    }
    const int npad = ld - N;  // padding columns stay zero
    for (int k = tid; k < nrows * npad; k += WBS) {
        const int r = k / npad, t = N + (k - r * npad);
        y[(size_t)(row0 + r) * ld + t] = 0.0;
    }
}

// ---------------------------------------------------------------------------
// Register variant for J <= 6 and ld == 2^J + 2: one wavefront per workgroup
// owns 64 consecutive time columns, which are one contiguous run of memory.
// The run is copied to LDS with 16-byte loads, every lane then pulls its own
// column into registers, runs all J levels there (compile-time unrolled, the
// previous pre-level value carried in a scalar so updates are in place) and
// puts it back; the tile leaves with 16-byte stores.  No barriers between
// levels; 33 KiB in flight per wavefront keep HBM busy at 4 wavefronts per CU.
template <int J>
struct WConst {
    static constexpr int N = (1 << J) + 1;
    static constexpr int LD = N + 1;
};

template <int J, bool TRANSPOSED>
__device__ inline void wavelet_levels(double (&v)[(1 << J) + 1])
{
#pragma unroll
    for (int step = 0; step < J; ++step) {
        const int j = TRANSPOSED ? (J - step) : (step + 1);
        const int S = 1 << (J - j);
        const int n = 1 << j;
        // 2^(j/2): a power of two, times sqrt(2) (correctly rounded) for odd j
        const double s = (double)(1 << (j / 2)) * ((j & 1) ? 1.4142135623730951 : 1.0);
        double prev = 0.0;
#pragma unroll
        for (int k = 0; k <= n; ++k) {
            const double xc = v[k * S];
            const double xl = prev;
            const double xr = (k < n) ? v[(k + 1) * S] : 0.0;
            double o;
            if (!TRANSPOSED) {
                if (k & 1)
                    o = 0.5 * (xl + xr) + s * xc;
                else if (k == 0)
                    o = xc - s * xr;
                else if (k == n)
                    o = xc - s * xl;
                else
                    o = xc - 0.5 * s * (xl + xr);
            } else {
                if (k & 1) {
                    const double cl = (k - 1 == 0) ? 1.0 : 0.5;
                    const double cr = (k + 1 == n) ? 1.0 : 0.5;
                    o = s * (xc - cl * xl - cr * xr);
                } else {
                    o = xc + 0.5 * xl + 0.5 * xr;
                }
            }
            v[k * S] = o;
            prev = xc;
        }
    }
}

template <int J, bool TRANSPOSED>
__global__ __launch_bounds__(64) void wavelet_reg_kernel(int32_t M, const double *x, double *y)
{
    constexpr int N = WConst<J>::N, LD = WConst<J>::LD, Q = LD / 2;
    extern __shared__ double tile[];  // [64][LD], a verbatim copy of the global run
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    const int nrows = min(64, M - row0);
    const int n2 = nrows * Q;  // double2 pieces of the run
    const double2 *x2 = reinterpret_cast<const double2 *>(x + (size_t)row0 * LD);
    double2 *y2 = reinterpret_cast<double2 *>(y + (size_t)row0 * LD);
    double2 *t2 = reinterpret_cast<double2 *>(tile);

    if (nrows == 64) {  // full tile: no per-piece predicates
        double2 buf[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) buf[q] = x2[lane + 64 * q];
#pragma unroll
        for (int q = 0; q < Q; ++q) t2[lane + 64 * q] = buf[q];
    } else {
        for (int i = lane; i < n2; i += 64) t2[i] = x2[i];
    }
    __syncthreads();

    if (lane < nrows) {
        double v[N];
        const double *c = tile + lane * LD;
#pragma unroll
        for (int t = 0; t < N; ++t) v[t] = c[t];
        wavelet_levels<J, TRANSPOSED>(v);
        double *cw = tile + lane * LD;
#pragma unroll
        for (int t = 0; t < N; ++t) cw[t] = v[t];
    }
    __syncthreads();

    if (nrows == 64) {
        double2 buf[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) buf[q] = t2[lane + 64 * q];
#pragma unroll
        for (int q = 0; q < Q; ++q) y2[lane + 64 * q] = buf[q];
    } else {
        for (int i = lane; i < n2; i += 64) y2[i] = t2[i];
    }
}

template <int J>
int launch_reg(hipStream_t st, int32_t M, int transposed, const double *x, double *y)
{
    const unsigned grid = (unsigned)((M + 63) / 64);
    const size_t lds = sizeof(double) * 64 * WConst<J>::LD;
    if (lds > 64 * 1024) {
        // J = 7: 66 560 bytes of dynamic LDS, above the 64 KiB a kernel gets without
        // asking (the device has 160 KiB per CU); the attribute belongs to the
        // function on the current device, so it is set on every launch (cheap)
        STK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wavelet_reg_kernel<J, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        STK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wavelet_reg_kernel<J, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (transposed)
        hipLaunchKernelGGL((wavelet_reg_kernel<J, true>), dim3(grid), dim3(64), lds, st, M, x, y);
    else
        hipLaunchKernelGGL((wavelet_reg_kernel<J, false>), dim3(grid), dim3(64), lds, st, M, x, y);
    STK_LAUNCH_CHECK();
    return 0;
}

int g_wavelet_variant = 0;  // 0 = automatic, 1 = LDS kernel only

}  // namespace

int stk_wavelet_set_tuning(const char *key, int32_t value)
{
    if (std::strcmp(key, "wavelet_variant") == 0) {
        g_wavelet_variant = value;
        return 0;
    }
    return 1;
}

extern "C" int stk_wavelet_apply(void *stream, int32_t M, int32_t J, int32_t ld, int32_t transposed,
                                 const double *x, double *y)
{
    const stk_timed timed_(STK_OP_WAVELET, stream);
    STK_REQUIRE(M > 0 && J >= 0 && J <= 11, "stk_wavelet_apply: bad M=%d J=%d", M, J);
    const int N = (1 << J) + 1;
    STK_REQUIRE(ld >= N, "stk_wavelet_apply: ld=%d < 2^J+1=%d", ld, N);
    STK_REQUIRE(x && y, "stk_wavelet_apply: null pointer");
    STK_REQUIRE(N <= TILE_ELEMS, "stk_wavelet_apply: J too large");
    if (g_wavelet_variant == 0 && J >= 1 && J <= 7 && ld == N + 1 &&
        (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        hipStream_t st = stk_stream(stream);
        switch (J) {
            case 1: return launch_reg<1>(st, M, transposed, x, y);
            case 2: return launch_reg<2>(st, M, transposed, x, y);
            case 3: return launch_reg<3>(st, M, transposed, x, y);
            case 4: return launch_reg<4>(st, M, transposed, x, y);
            case 5: return launch_reg<5>(st, M, transposed, x, y);
            case 6: return launch_reg<6>(st, M, transposed, x, y);
            default: return launch_reg<7>(st, M, transposed, x, y);
        }
    }
    int R = TILE_ELEMS / N;
    if (R > 128) R = 128;
    const unsigned grid = (unsigned)((M + R - 1) / R);
    const size_t lds = sizeof(double) * (size_t)R * N;
    if (transposed)
        hipLaunchKernelGGL(wavelet_kernel<true>, dim3(grid), dim3(WBS), lds, stk_stream(stream), M, J, ld, R, x,
                           y);
    else
        hipLaunchKernelGGL(wavelet_kernel<false>, dim3(grid), dim3(WBS), lds, stk_stream(stream), M, J, ld, R,
                           x, y);
    STK_LAUNCH_CHECK();
    return 0;
}
