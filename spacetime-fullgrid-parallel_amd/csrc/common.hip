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

extern "C" int stk_version(void) { return 200; }

int stk_cu_count()
{
    // per device (one process may drive several), queried once per device and thread
    static thread_local int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cached[dev] = n;
        else
            cached[dev] = 256;
    }
    return cached[dev];
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
