// Internal helpers shared by the libstk translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "stk.h"

void stk_set_error(const char *fmt, ...);

#define STK_HIP(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) {                                              \
            stk_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, \
                          hipGetErrorString(e_));                            \
            return 1;                                                        \
        }                                                                    \
    } while (0)

#define STK_LAUNCH_CHECK() STK_HIP(hipGetLastError())

#define STK_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            stk_set_error(__VA_ARGS__); \
            return 2;                   \
        }                               \
    } while (0)

static inline hipStream_t stk_stream(void *s) { return (hipStream_t)s; }

// Per-operation device-time counters of the C ABI (stk_timing_*, common.hip; the
// counterpart of LinearOperatorMPI.time_applies, reference mpi_kron.py:23-36): a
// public entry point brackets what it enqueues with two HIP events while timing
// is enabled.  Costs one branch otherwise.
enum { STK_OP_KRON = 0, STK_OP_SPACE, STK_OP_TIME, STK_OP_WAVELET, STK_OP_MULTIGRID, STK_OP_BLAS1, STK_OP_CLASSES };
extern bool g_stk_timing;
void *stk_timing_begin(int op, hipStream_t st);
void stk_timing_end(void *slot, hipStream_t st);
struct stk_timed {
    hipStream_t st;
    void *slot;
    stk_timed(int op, void *stream)
        : st(stk_stream(stream)), slot(g_stk_timing ? stk_timing_begin(op, stk_stream(stream)) : nullptr)
    {
    }
    ~stk_timed()
    {
        if (slot) stk_timing_end(slot, st);
    }
};

// Compute units of the current device, queried once (a property query per
// launch costs more host time than the launch itself).
int stk_cu_count();

// Grid size for a flat, memory-bound kernel: enough workgroups to fill
// 256 CUs x 8 resident blocks, grid-stride for the rest.
static inline unsigned stk_flat_grid(int64_t work_items, int block, int per_thread = 1)
{
    int64_t blocks = (work_items + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
    if (blocks < 1) blocks = 1;
    const int64_t cap = 256 * 16;
    return (unsigned)(blocks < cap ? blocks : cap);
}

// ---- fused coarse sub-V-cycle (mg_coarse.hip), driven by mg.hip --------------------
// Description of one multigrid level handed over by mg.hip.
struct stk_coarse_plan;
struct stk_coarse_level {
    int n;
    bool ok;  // all sliced-ELL pieces present
    stk_ell_rows a, fwd, bwd, p, r;
    const int32_t *fwd_pos, *bwd_pos;  // group g of a sweep = ELL positions pos[g]..pos[g+1]
    int n_fwd, n_bwd;
    double *u, *f, *res;  // workspaces of this level
};
stk_coarse_plan *stk_coarse_plan_build(const stk_coarse_level *lv, int Lc, int smoothsteps);
void stk_coarse_plan_free(stk_coarse_plan *p);
bool stk_coarse_plan_in_lds(const stk_coarse_plan *p);
int stk_coarse_plan_levels(const stk_coarse_plan *p);
int stk_coarse_plan_set_members(stk_coarse_plan *p, int level, int n_kinds, int32_t n, const int32_t *const *indptr,
                                const int32_t *const *indices, const double *const *data);
int stk_coarse_plan_run(const stk_coarse_plan *p, hipStream_t st, int n_loc, int ld, double ca, const double *cm,
                        const int32_t *kind, const double *coarse_inv);

// The plan takes over device arrays (freed by stk_mg_destroy): mg_build.hip.
void stk_mg_adopt(stk_mg *mg, void *const *dev_ptrs, int n);

// Row-gather engine launcher of rows_ell.hip (mode 0 = SPMM, 1 = Gauss-Seidel group).
int stk_rows_ell_launch(hipStream_t st, int mode, const stk_ell_rows *e, int32_t pos_begin, int32_t pos_end,
                        int32_t n_loc, int32_t ld, int64_t x_rows, int64_t y_rows, double ca, const double *cm,
                        const double *x, double alpha, double beta, const double *z, double *y, int zero_own = 0);

// y = A(t) x - B x2 in one pass (the restricted residual (R A) u - R f); -1: no
// instantiation for this pair of matrices, take the two passes.
int stk_rows_ell2_launch(hipStream_t st, const stk_ell_rows *e, const stk_ell_rows *e2, int32_t n_loc, int32_t ld,
                         int64_t x_rows, int64_t y_rows, double ca, const double *cm, const double *x,
                         const double *x2, double *y);

// ---- slab access from device code (kron_ell.hip, rows_ell.hip, mg_coarse.hip) -----
#if defined(__HIPCC__)
typedef int stk_v4i __attribute__((ext_vector_type(4)));

// 128-bit buffer descriptor over a whole array: gathers and stores then take a
// 32-bit byte offset per lane (no 64-bit address arithmetic, fewer VGPRs).
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

__device__ inline double2 buf_load2(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off)
{
    const stk_v4i v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
    double2 out;
    __builtin_memcpy(&out, &v, 16);
    return out;
}

__device__ inline void buf_store2(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, double2 val)
{
    stk_v4i v;
    __builtin_memcpy(&v, &val, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, byte_off, 0, 0);
}

// A slab x[row*ld + t] as a kernel sees it; a lane moves one pair of time steps.
// WIDE = false: buffer descriptor, row offsets are 32-bit BYTE offsets (slabs
// below 4 GiB).  WIDE = true: row offsets count 16-byte units and are widened
// to 64-bit addresses (slabs up to 64 GiB; costs address VGPRs).
template <bool WIDE>
struct stk_slab {
    __amdgpu_buffer_rsrc_t rs;
    const char *base;
    __device__ stk_slab() : base(nullptr) {}
    __device__ stk_slab(const void *ptr, uint32_t bytes)
        : rs(make_rsrc(ptr, bytes)), base(static_cast<const char *>(ptr))
    {
    }
    // units of a row offset per row
    __device__ static inline uint32_t row_stride(int ld) { return WIDE ? ((uint32_t)ld * 8u) >> 4 : (uint32_t)ld * 8u; }
    __device__ inline double2 load(uint32_t row_off, uint32_t t_bytes) const
    {
        if constexpr (WIDE)
            return *reinterpret_cast<const double2 *>(base + (((size_t)row_off) << 4) + t_bytes);
        else
            return buf_load2(rs, row_off + t_bytes);
    }
    __device__ inline void store(uint32_t row_off, uint32_t t_bytes, double2 v) const
    {
        if constexpr (WIDE)
            *reinterpret_cast<double2 *>(const_cast<char *>(base) + (((size_t)row_off) << 4) + t_bytes) = v;
        else
            buf_store2(rs, row_off + t_bytes, v);
    }
};
#endif
