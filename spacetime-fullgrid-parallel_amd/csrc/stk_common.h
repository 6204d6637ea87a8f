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

// Grid size for a flat, memory-bound kernel: enough workgroups to fill
// 256 CUs x 8 resident blocks, grid-stride for the rest.
static inline unsigned stk_flat_grid(int64_t work_items, int block, int per_thread = 1)
{
    int64_t blocks = (work_items + (int64_t)block * per_thread - 1) / ((int64_t)block * per_thread);
    if (blocks < 1) blocks = 1;
    const int64_t cap = 256 * 16;
    return (unsigned)(blocks < cap ? blocks : cap);
}
