// Sizing prototype for an XCD-local, flag-synchronised Gauss-Seidel wavefront
// (DESIGN.md section 8.2).  NOT part of libstk.
//
// A persistent grid; every workgroup finds its XCD (HW_REG_XCC_ID) and the
// workgroups of one XCD walk `steps` stages together:
//   mode 0  per-XCD barrier only (one monotonic counter per XCD, agent-scope add,
//           sc1 poll), empty stages: the bare synchronisation price;
//   mode 1  + wait for the two neighbour XCDs to have finished the stage before
//           (what a band wavefront crossing XCD borders needs), with the lane-0
//           agent release before the arrival and the acquire after the wait;
//   mode 2  mode 0 + a stage body the size of an L2-resident Gauss-Seidel stage:
//           every lane gathers 7 x 16 B (sc1: L1 bypassed, same-XCD producers) from
//           rows of its XCD's own `region_kb` window, and stores 16 B;
//   mode 3  mode 2 without any synchronisation (the body alone);
//   mode 4  (round 4) NO barrier: every workgroup owns a contiguous band of the XCD's
//           window and a flag of its own; before stage q it waits until its two
//           neighbour workgroups OF THE SAME XCD have published stage q - 1, runs
//           the body, publishes stage q.  The stages pipeline along the skew (a
//           workgroup may be one stage ahead of its neighbour), nobody waits for
//           the slowest workgroup of the XCD;
//   mode 5  mode 4 with empty stages: the bare price of the neighbour flags.
// Every spin is bounded (the kernel gives up, sets a flag and every later wait
// falls through), so the grid always drains.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gs_sync_proto tools/gs_sync_proto.hip
//   tools/bin/gs_sync_proto [wg_per_cu=1..3] [steps=2000] [region_kb=3072] [rows_per_wg=60]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int BS = 512;
constexpr int LINE = 32;  // uint32 per 128-byte line: one counter per line

struct Sync {
    uint32_t *census;   // [8][LINE] workgroups seen per XCD
    uint32_t *bar;      // [8][LINE] monotonic arrival counters
    uint32_t *all;      // [LINE]    one-time whole-grid arrival counter
    uint32_t *gave_up;  // [LINE]
    uint32_t *wgflag;   // [8][MAXWG][LINE] stage published by workgroup `me` of an XCD
};
constexpr int MAXWG = 128;

__device__ __forceinline__ uint32_t load_sc1(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool wait_ge(const uint32_t *p, uint32_t want, uint32_t *gave_up)
{
    for (int spin = 0; spin < (1 << 20); ++spin) {
        if ((int32_t)(load_sc1(p) - want) >= 0) return true;
        if (load_sc1(gave_up)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

__global__ __launch_bounds__(BS) void proto_kernel(Sync s, int mode, int steps, double2 *data, int region_vec,
                                                   int rows_per_wg, int row_vec, long long *cycles)
{
    __shared__ uint32_t sh[4];
    const int tid = threadIdx.x;
    uint32_t xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (tid == 0) {
        sh[0] = atomicAdd(&s.census[xcc * LINE], 1u);  // my index inside the XCD
        atomicAdd(s.all, 1u);
        wait_ge(s.all, gridDim.x, s.gave_up);  // census complete
        sh[1] = load_sc1(&s.census[xcc * LINE]);
        sh[2] = load_sc1(&s.census[((xcc + 7) & 7) * LINE]);
        sh[3] = load_sc1(&s.census[((xcc + 1) & 7) * LINE]);
    }
    __syncthreads();
    const uint32_t me = sh[0], n_here = sh[1], n_lo = sh[2], n_hi = sh[3];
    double2 *mine = data + (size_t)xcc * region_vec;
    // a lane = one 16-byte pair of a row; a row = row_vec pairs; this workgroup's
    // rows of a stage are spread over the XCD's window like the rows of one
    // dependency group (every 4th row)
    const int lanes_per_row = row_vec;
    const int r_in_wg = tid / lanes_per_row, p = tid % lanes_per_row;
    const int rows_in_window = region_vec / row_vec;
    double2 acc = make_double2(0.0, 0.0);
    const long long t0 = wall_clock64();
    uint32_t *my_flag = s.wgflag + ((size_t)xcc * MAXWG + me) * LINE;
    for (int q = 0; q < steps; ++q) {
        if (mode >= 4) {
            // neighbour flags instead of a barrier: stage q may start once both
            // neighbours have finished stage q - 1 (they then cannot start q + 1
            // before this workgroup has published q)
            if (tid == 0 && q > 0) {
                if (me > 0) wait_ge(my_flag - LINE, (uint32_t)q, s.gave_up);
                if (me + 1 < n_here) wait_ge(my_flag + LINE, (uint32_t)q, s.gave_up);
            }
            __syncthreads();
        }
        if (mode == 2 || mode == 3 || mode == 4) {
            for (int rr = r_in_wg; rr < rows_per_wg; rr += BS / lanes_per_row) {
                // row of this stage: strided by 4 (groups), shifted by the stage
                const int row = (int)(((me * rows_per_wg + rr) * 4u + (q & 3)) % (uint32_t)rows_in_window);
                double2 g[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    // neighbours: rows a mesh row apart and next to the row
                    const int off[7] = {-257, -256, -1, 0, 1, 256, 257};
                    int nb = row + off[k];
                    nb = nb < 0 ? nb + rows_in_window : (nb >= rows_in_window ? nb - rows_in_window : nb);
                    const double2 *src = mine + (size_t)nb * row_vec + p;
                    double lo, hi;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(g[k]) : "v"(src) : "memory");
                    (void)lo;
                    (void)hi;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    acc.x = fma(g[k].x, 0.125, acc.x);
                    acc.y = fma(g[k].y, 0.125, acc.y);
                }
                mine[(size_t)row * row_vec + p] = acc;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (mode == 3) continue;
        if (mode >= 4) {
            __syncthreads();  // every wave's stores are out (s_waitcnt above)
            if (tid == 0) __hip_atomic_store(my_flag, (uint32_t)(q + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        __syncthreads();
        if (tid == 0) {
            if (mode == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            atomicAdd(&s.bar[xcc * LINE], 1u);
            wait_ge(&s.bar[xcc * LINE], n_here * (uint32_t)(q + 1), s.gave_up);
            if (mode == 1 && q > 0) {
                wait_ge(&s.bar[((xcc + 7) & 7) * LINE], n_lo * (uint32_t)q, s.gave_up);
                wait_ge(&s.bar[((xcc + 1) & 7) * LINE], n_hi * (uint32_t)q, s.gave_up);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc.x == 123.456) data[0] = acc;
}

int main(int argc, char **argv)
{
    const int per_cu = argc > 1 ? atoi(argv[1]) : 1;
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;
    const int region_kb = argc > 3 ? atoi(argv[3]) : 3072;
    const int rows_per_wg = argc > 4 ? atoi(argv[4]) : 60;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int grid = n_cu * per_cu;
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    printf("device %s, %d CUs, grid %d x %d threads, wall clock %d kHz, window %d KB per XCD\n", prop.name, n_cu,
           grid, BS, clk_khz, region_kb);
    int resident = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, proto_kernel, BS, 0));
    if (resident < per_cu) {
        fprintf(stderr, "only %d workgroups per CU are resident: refusing a grid that cannot be co-resident\n",
                resident);
        return 2;
    }
    Sync s;
    uint32_t *words;
    const size_t n_words = (8 + 8 + 1 + 1 + 8 * MAXWG) * LINE;
    CK(hipMalloc(&words, n_words * 4));
    s.census = words;
    s.bar = words + 8 * LINE;
    s.all = words + 16 * LINE;
    s.gave_up = words + 17 * LINE;
    s.wgflag = words + 18 * LINE;
    if (grid / 8 > MAXWG) {
        fprintf(stderr, "more than %d workgroups per XCD\n", MAXWG);
        return 2;
    }
    // a row of a slab that one XCD owns: 8 time steps = 4 pairs (64 bytes)
    const int row_vec = 4;
    const int region_vec = region_kb * 1024 / 16;
    double2 *data;
    CK(hipMalloc(&data, (size_t)8 * region_vec * 16));
    CK(hipMemset(data, 0, (size_t)8 * region_vec * 16));
    long long *cycles;
    CK(hipMalloc(&cycles, sizeof(long long) * grid));
    std::vector<long long> h(grid);
    std::vector<uint32_t> hw(n_words);
    const char *names[6] = {"XCD barrier, empty stages", "XCD barrier + neighbour XCD flags + release/acquire",
                            "XCD barrier + L2-window stage body", "stage body alone (no synchronisation)",
                            "neighbour-workgroup flags + stage body, no barrier", "neighbour-workgroup flags, empty stages"};
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(words, 0, n_words * 4));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(proto_kernel, dim3(grid), dim3(BS), 0, 0, s, mode, steps, data, region_vec, rows_per_wg,
                               row_vec, cycles);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), cycles, sizeof(long long) * grid, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hw.data(), words, n_words * 4, hipMemcpyDeviceToHost));
            long long mx = 0;
            for (long long c : h) mx = c > mx ? c : mx;
            if (rep == 1) {
                printf("mode %d  %-52s %8.3f us per stage (event %8.3f)  gave_up=%u  per-XCD workgroups:", mode,
                       names[mode], (double)mx / clk_khz * 1e3 / steps, ms * 1e3 / steps, hw[17 * LINE]);
                for (int x = 0; x < 8; ++x) printf(" %u", hw[x * LINE]);
                if (mode >= 2 && mode <= 4)
                    printf("  lanes/stage/XCD %d, bytes gathered+stored per stage per XCD %.2f MB", rows_per_wg * row_vec * grid / 8,
                           rows_per_wg * row_vec * (grid / 8) * 128.0 / 1e6);
                printf("\n");
            }
        }
    }
    return 0;
}
