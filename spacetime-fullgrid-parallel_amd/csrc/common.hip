// Error reporting and device queries of libstk.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

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

// ---- per-operation device-time counters ----------------------------------------
bool g_stk_timing = false;

namespace {

struct Bracket {
    hipEvent_t begin, end;
    int op;
};
std::mutex g_timing_mutex;
std::vector<Bracket *> g_timing_open, g_timing_spare;  // recorded, not yet read / reusable
double g_timing_seconds[STK_OP_CLASSES] = {0};
int64_t g_timing_calls[STK_OP_CLASSES] = {0};
const char *const g_timing_names[STK_OP_CLASSES] = {"kron", "space", "time", "wavelet", "multigrid", "blas1"};

// reads every recorded bracket (waits for its end event); caller holds the mutex
void timing_collect()
{
    for (Bracket *b : g_timing_open) {
        float ms = 0.0f;
        if (hipEventSynchronize(b->end) == hipSuccess && hipEventElapsedTime(&ms, b->begin, b->end) == hipSuccess) {
            g_timing_seconds[b->op] += 1.0e-3 * ms;
            g_timing_calls[b->op] += 1;
        }
        g_timing_spare.push_back(b);
    }
    g_timing_open.clear();
}

}  // namespace

void *stk_timing_begin(int op, hipStream_t st)
{
    Bracket *b = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_timing_mutex);
        if (!g_timing_spare.empty()) {
            b = g_timing_spare.back();
            g_timing_spare.pop_back();
        }
    }
    if (!b) {
        b = new Bracket();
        if (hipEventCreate(&b->begin) != hipSuccess || hipEventCreate(&b->end) != hipSuccess) {
            delete b;
            return nullptr;
        }
    }
    b->op = op;
    (void)hipEventRecord(b->begin, st);
    return b;
}

void stk_timing_end(void *slot, hipStream_t st)
{
    Bracket *b = static_cast<Bracket *>(slot);
    (void)hipEventRecord(b->end, st);
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    g_timing_open.push_back(b);
    if (g_timing_open.size() >= 4096) timing_collect();  // bounded number of live events
}

extern "C" int stk_timing_enable(int32_t on)
{
    g_stk_timing = on != 0;
    return 0;
}

extern "C" int stk_timing_reset(void)
{
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    timing_collect();
    for (int c = 0; c < STK_OP_CLASSES; ++c) g_timing_seconds[c] = 0.0, g_timing_calls[c] = 0;
    return 0;
}

extern "C" int stk_timing_get(const char *op_class, int64_t *calls, double *seconds)
{
    STK_REQUIRE(op_class != nullptr, "stk_timing_get: null class name");
    std::lock_guard<std::mutex> lock(g_timing_mutex);
    timing_collect();
    for (int c = 0; c < STK_OP_CLASSES; ++c)
        if (std::strcmp(op_class, g_timing_names[c]) == 0) {
            if (calls) *calls = g_timing_calls[c];
            if (seconds) *seconds = g_timing_seconds[c];
            return 0;
        }
    stk_set_error("stk_timing_get: unknown class '%s' (kron, space, time, wavelet, multigrid, blas1)", op_class);
    return 2;
}
