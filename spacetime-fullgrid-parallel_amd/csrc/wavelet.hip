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

}  // namespace

extern "C" int stk_wavelet_apply(void *stream, int32_t M, int32_t J, int32_t ld, int32_t transposed,
                                 const double *x, double *y)
{
    STK_REQUIRE(M > 0 && J >= 0 && J <= 11, "stk_wavelet_apply: bad M=%d J=%d", M, J);
    const int N = (1 << J) + 1;
    STK_REQUIRE(ld >= N, "stk_wavelet_apply: ld=%d < 2^J+1=%d", ld, N);
    STK_REQUIRE(x && y, "stk_wavelet_apply: null pointer");
    STK_REQUIRE(N <= TILE_ELEMS, "stk_wavelet_apply: J too large");
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
