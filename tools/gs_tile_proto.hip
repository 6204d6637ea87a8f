// Sizing prototype for an LDS-tiled, temporally blocked Gauss-Seidel smoothing call
// (all sweeps x dependency groups of a level visit on one mesh tile while it sits in
// LDS).  NOT part of libstk.  It moves exactly the bytes such a kernel would move --
// and nothing else -- to answer the one question that decides the design on slabs
// whose rows do not fit LDS whole:
//
//   A tile with the halo its 8 stages need (2 sweeps x 4 groups: 8 layers) is
//   (c + 16)^2 rows.  160 KB of LDS hold that only for a CHUNK of 8 time steps
//   (64 bytes of every 528-byte row at J_time = 6).  The sibling chunks of a row
//   share its 128-byte lines: unless the L2 of the XCD serves them to the sibling
//   workgroups, every line crosses the fabric 2-3 times.
//
// A persistent grid, one 512-thread workgroup per CU; an item = (tile, chunk):
//   phase 1  the chunk's piece of every row of the tile + halo  -> LDS
//   phase 2  the piece of f of every row the stages update      -> registers
//   phase 3  the chunk's piece of the tile's own rows, LDS      -> out
// order 0: items tile-major, chunk-minor, every XCD a contiguous range: the 32
//          workgroups of an XCD work on the sibling chunks of 3-4 tiles at once;
// order 1: chunk-major (no sibling is ever near in time): the worst case;
// order 2: whole rows, tiles of (c2 + 16)^2 <= 300 rows (what fits without chunks).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gs_tile_proto tools/gs_tile_proto.hip
//   tools/bin/gs_tile_proto [n=1023] [n_loc=65] [ld=66] [core=28] [halo=8] [reps=5]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

constexpr int BS = 512;

struct Args {
    const double *u, *f;
    double *out;
    const int32_t *perm;  // row number of mesh vertex (x, y), or NULL: lexicographic
    double *sink;
    int32_t n, n_loc, ld;
    int32_t core, halo, tiles_x, n_tiles;
    int32_t n_chunks, chunk_steps;  // chunks of chunk_steps time steps, the last one takes the remainder
    int32_t lanes;                  // lanes (16 bytes each) per row and chunk in LDS
    int32_t order;
    int32_t items, per_xcd;
};

__device__ inline int row_of(const Args &a, int x, int y) { return a.perm ? a.perm[y * a.n + x] : y * a.n + x; }

__global__ __launch_bounds__(BS) void tile_kernel(const Args a)
{
    extern __shared__ double2 lds[];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int ext = a.core + 2 * a.halo;
    double acc = 0.0;
    for (int it = xcd * a.per_xcd + w; it < min((xcd + 1) * a.per_xcd, a.items); it += wpx) {
        int tile, chunk;
        if (a.order == 1) {
            chunk = it / a.n_tiles;
            tile = it - chunk * a.n_tiles;
        } else {
            tile = it / a.n_chunks;
            chunk = it - tile * a.n_chunks;
        }
        const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        const int x0 = tx * a.core - a.halo, y0 = ty * a.core - a.halo;
        const int t_begin = chunk * a.chunk_steps;
        const int t_end = (chunk == a.n_chunks - 1) ? a.n_loc : t_begin + a.chunk_steps;
        const int pieces = (t_end - t_begin + 1) / 2;  // 16-byte pieces of this chunk per row
        // phase 1: tile + halo -> LDS
        for (int p = tid; p < ext * ext * a.lanes; p += BS) {
            const int r = p / a.lanes, l = p - r * a.lanes;
            const int lx = r % ext, ly = r / ext;
            const int x = x0 + lx, y = y0 + ly;
            if (l < pieces && x >= 0 && y >= 0 && x < a.n && y < a.n) {
                const double *src = a.u + (size_t)row_of(a, x, y) * a.ld + t_begin + 2 * l;
                lds[p] = *reinterpret_cast<const double2 *>(src);
            }
        }
        // phase 2: f of the rows that are updated (one layer less)
        const int fext = ext - 2;
        for (int p = tid; p < fext * fext * a.lanes; p += BS) {
            const int r = p / a.lanes, l = p - r * a.lanes;
            const int lx = r % fext, ly = r / fext;
            const int x = x0 + 1 + lx, y = y0 + 1 + ly;
            if (l < pieces && x >= 0 && y >= 0 && x < a.n && y < a.n) {
                const double *src = a.f + (size_t)row_of(a, x, y) * a.ld + t_begin + 2 * l;
                const double2 v = *reinterpret_cast<const double2 *>(src);
                acc += v.x + v.y;
            }
        }
        __syncthreads();
        // phase 3: the tile's own rows, LDS -> out
        for (int p = tid; p < a.core * a.core * a.lanes; p += BS) {
            const int r = p / a.lanes, l = p - r * a.lanes;
            const int lx = r % a.core, ly = r / a.core;
            const int x = x0 + a.halo + lx, y = y0 + a.halo + ly;
            if (l < pieces && x < a.n && y < a.n) {
                double *dst = a.out + (size_t)row_of(a, x, y) * a.ld + t_begin + 2 * l;
                *reinterpret_cast<double2 *>(dst) = lds[((ly + a.halo) * ext + lx + a.halo) * a.lanes + l];
            }
        }
        __syncthreads();
    }
    if (acc == 123.456) a.sink[0] = acc;
}

// reference rate: out = u + f, flat
__global__ __launch_bounds__(256) void stream_kernel(size_t n2, const double2 *u, const double2 *f, double2 *out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        const double2 a = u[i], b = f[i];
        out[i] = make_double2(a.x + b.x, a.y + b.y);
    }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1023;
    const int n_loc = argc > 2 ? atoi(argv[2]) : 65;
    const int ld = argc > 3 ? atoi(argv[3]) : 66;
    const int core = argc > 4 ? atoi(argv[4]) : 28;
    const int halo = argc > 5 ? atoi(argv[5]) : 8;
    const int reps = argc > 6 ? atoi(argv[6]) : 5;
    const size_t M = (size_t)n * n, words = M * ld;
    double *u, *f, *out, *sink;
    CK(hipMalloc(&u, words * 8));
    CK(hipMalloc(&f, words * 8));
    CK(hipMalloc(&out, words * 8));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(u, 0, words * 8));
    CK(hipMemset(f, 0, words * 8));
    CK(hipMemset(out, 0, words * 8));
    // a numbering that scatters mesh neighbours like the hierarchical one: the
    // vertices of the four parity classes (x & 1, y & 1) follow each other
    std::vector<int32_t> perm(M);
    {
        int32_t next = 0;
        for (int cls = 0; cls < 4; ++cls)
            for (int y = 0; y < n; ++y)
                for (int x = 0; x < n; ++x)
                    if (((x & 1) | ((y & 1) << 1)) == cls) perm[(size_t)y * n + x] = next++;
    }
    int32_t *d_perm;
    CK(hipMalloc(&d_perm, M * 4));
    CK(hipMemcpy(d_perm, perm.data(), M * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double pass_mb = (double)M * n_loc * 8 / 1e6;
    printf("grid %d x %d, %d steps, ld %d: one pass over a vector = %.0f MB\n", n, n, n_loc, ld, pass_mb);
    {
        float best = 1e9f;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, 0, words / 2, (const double2 *)u,
                               (const double2 *)f, (double2 *)out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2 && ms < best) best = ms;
        }
        printf("stream out = u + f (3 passes incl. padding): %.3f ms = %.2f TB/s\n", best, 3.0 * words * 8 / best / 1e9);
    }
    CK(hipFuncSetAttribute((const void *)tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int numbering = 0; numbering < 2; ++numbering)
        for (int order = 0; order < 3; ++order) {
            Args a;
            a.u = u, a.f = f, a.out = out, a.sink = sink;
            a.perm = numbering ? d_perm : nullptr;
            a.n = n, a.n_loc = n_loc, a.ld = ld;
            a.halo = halo;
            a.order = order;
            if (order == 2) {  // whole rows: as many as LDS holds
                a.chunk_steps = n_loc;
                a.n_chunks = 1;
                a.lanes = (n_loc + 1) / 2;
                int ext = 2 * halo + 1;
                while ((size_t)(ext + 1) * (ext + 1) * a.lanes * 16 <= 160 * 1024 - 512) ++ext;
                a.core = ext - 2 * halo;
            } else {
                a.chunk_steps = 8;
                a.n_chunks = n_loc / 8 > 0 ? n_loc / 8 : 1;
                const int last = n_loc - (a.n_chunks - 1) * 8;
                a.lanes = (last + 1) / 2;
                a.core = core;
            }
            const int ext = a.core + 2 * halo;
            const size_t lds = (size_t)ext * ext * a.lanes * 16;
            if (lds > 160 * 1024 - 256) {
                printf("order %d: %zu bytes of LDS do not fit\n", order, lds);
                continue;
            }
            a.tiles_x = (n + a.core - 1) / a.core;
            a.n_tiles = a.tiles_x * a.tiles_x;
            a.items = a.n_tiles * a.n_chunks;
            a.per_xcd = (a.items + 7) / 8;
            hipDeviceProp_t prop;
            CK(hipGetDeviceProperties(&prop, 0));
            const int grid = prop.multiProcessorCount / 8 * 8;
            float best = 1e9f;
            for (int r = 0; r < reps + 2; ++r) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(tile_kernel, dim3(grid), dim3(BS), lds, 0, a);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2 && ms < best) best = ms;
            }
            const double red_u = (double)ext * ext / (a.core * a.core), red_f = (double)(ext - 2) * (ext - 2) / (a.core * a.core);
            printf("%-14s order %d (%s): core %d, tile+halo %d^2 rows, %d chunk(s), LDS %zu B, %d items on %d workgroups: "
                   "%.3f ms = %.2f x the 3-pass stream bound at 5 TB/s; requested %.2f passes (u %.2f, f %.2f, out 1)\n",
                   numbering ? "class-major" : "lexicographic", order,
                   order == 0 ? "sibling chunks together" : order == 1 ? "chunk-major" : "whole rows", a.core, ext,
                   a.n_chunks, lds, a.items, grid, best, best / (3.0 * pass_mb / 5e3 * 1e-3 * 1e3), red_u + red_f + 1.0, red_u,
                   red_f);
        }
    return 0;
}
