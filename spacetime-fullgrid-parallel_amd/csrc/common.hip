// Error reporting and device queries of libstk.
#include <cstdarg>
#include <cstdio>

#include "stk_common.h"

static thread_local char g_err[1024] = "";

void stk_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *stk_last_error(void) { return g_err; }

extern "C" int stk_version(void) { return 100; }

int stk_cu_count()
{
    static int cached = 0;
    if (cached == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cached = n;
        else
            cached = 256;
    }
    return cached;
}

extern "C" int stk_device_info(int32_t *n_cu, int32_t *wave_size, int64_t *hbm_bytes)
{
    int dev = 0;
    STK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    STK_HIP(hipGetDeviceProperties(&p, dev));
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (wave_size) *wave_size = p.warpSize;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return 0;
}
