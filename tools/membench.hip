// membench.hip -- what the MI355X memory system gives the ACCESS PATTERNS of the epoch kernels:
// random embedding-row gathers (256 B / 512 B rows, a burst of rows per wavefront, then a
// dependent consume) and random-row float atomics (global_atomic_add_f32, one row = 64 lanes x 4 B
// per instruction), from tables of the sizes of BASELINE's configurations, cached and uncached
// allocations.  The numbers are the practical ceilings the roofline fractions of bench.py are read
// against (DESIGN.md "What the memory system gives this access pattern").
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/membench tools/membench.hip
//   tools/_bin/membench            (prints one line per experiment)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t s) { return s * 1103515245u + 12345u; }
__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// Each wavefront: `iters` passes; per pass BURST instructions, each fetching 64/LPR random rows of
// LPR lanes x 16 B (LPR = 16: 256-B rows, 4 rows per instruction; LPR = 32: 512-B rows, 2 rows),
// all issued back to back, then summed (the dependent consume).
template <int LPR, int BURST>
__global__ __launch_bounds__(256) void gather_kernel(const float *tab, uint32_t rows, int iters, float *out)
{
    const int lane = threadIdx.x & 63, grp = lane / LPR, p = lane % LPR;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t s = mix(wave * 977u + 13u);
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        float4 v[BURST];
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            s = lcg(s);
            const uint32_t r = mix(s + grp * 0x9e3779b9u) % rows;
            v[b] = *reinterpret_cast<const float4 *>(tab + (size_t)r * (LPR * 4) + p * 4);
        }
#pragma unroll
        for (int b = 0; b < BURST; ++b) acc += v[b].x + v[b].y + v[b].z + v[b].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

// Each wavefront: `iters` passes of BURST atomic instructions; an instruction adds to NC*64
// consecutive floats of one random row (lane c -> coordinate c), fire and forget.
template <int BURST>
__global__ __launch_bounds__(256) void atomic_kernel(float *tab, uint32_t rows, int row_floats, int iters, int mode)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t s = mix(wave * 977u + 13u);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            s = lcg(s);
            const uint32_t r = mix(s) % rows;
            float *p = tab + (size_t)r * row_floats;
            for (int c = lane; c < row_floats; c += 64) {
                if (mode == 0) atomicAdd(p + c, 1e-9f);
                else p[c] = 1e-9f;  // plain stores, for comparison
            }
        }
    }
}

// The update of one row as the kernels do it: read the row (W), then atomically add to it; rows
// per pass = BURST, all reads first, then all atomics.
template <int BURST>
__global__ __launch_bounds__(256) void rmw_kernel(float *tab, uint32_t rows, int iters)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t s = mix(wave * 977u + 13u);
    for (int it = 0; it < iters; ++it) {
        float v[BURST];
        uint32_t r[BURST];
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            s = lcg(s);
            r[b] = mix(s) % rows;
            v[b] = tab[(size_t)r[b] * 64 + lane];
        }
#pragma unroll
        for (int b = 0; b < BURST; ++b) atomicAdd(tab + (size_t)r[b] * 64 + lane, v[b] * 1e-9f);
    }
}

// Other read-modify-write flavours on random 256-B rows (64 lanes, one element per lane), to learn what the
// atomic rate depends on -- operations or dwords:
//   0 atomicAdd(float) whose OLD value is used (returning atomic)      1 atomicAdd(unsigned)
//   2 atomicAdd(double): 64 lanes x 8 B = a 512-B row                   3 atomicCAS(64 bit), one attempt
//   4 atomicMax(unsigned)
template <int BURST>
__global__ __launch_bounds__(256) void atomic_flavour_kernel(float *tab, uint32_t rows, int iters, int flavour, float *out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t s = mix(wave * 977u + 13u);
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < BURST; ++b) {
            s = lcg(s);
            const uint32_t r = mix(s) % rows;
            if (flavour == 0) {
                acc += atomicAdd(tab + (size_t)r * 64 + lane, 1e-9f);
            } else if (flavour == 1) {
                atomicAdd(reinterpret_cast<unsigned *>(tab) + (size_t)r * 64 + lane, 1u);
            } else if (flavour == 2) {
                atomicAdd(reinterpret_cast<double *>(tab) + (size_t)(r / 2) * 64 + lane, 1e-9);
            } else if (flavour == 3) {
                unsigned long long *q = reinterpret_cast<unsigned long long *>(tab) + (size_t)(r / 2) * 64 + lane;
                atomicCAS(q, 0ull, (unsigned long long)it);
            } else {
                atomicMax(reinterpret_cast<unsigned *>(tab) + (size_t)r * 64 + lane, (unsigned)it);
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// LDS read-modify-write rate (round 6: the ceiling of accumulating C3's 1 128 shared tag rows in LDS slices instead
// of publishing every cell of every interaction through the L2 float-atomic unit, csrc/tag_slices.hip).  A workgroup
// owns a table of ROWS x 8 cells x {W, G} in LDS (1 128 rows: 72 KB); a wavefront pass touches 8 random rows x 8
// cells (lane = row slot * 8 + cell), reads W and G, optionally evaluates the reference's float64 adagrad cell
// (PYX:416-449) and writes back -- mode 0: ds_add_f32 of new - old (several wavefronts may hit a cell: nothing is
// lost), mode 1: plain ds_write (a single writer), mode 2: ds_add_f32 without the arithmetic (the LDS rate alone).
template <int MODE>
__global__ __launch_bounds__(256) void lds_rmw_kernel(int rows, int iters, float *out)
{
    extern __shared__ float lds[];
    float *W = lds, *G = lds + (size_t)rows * 8;
    for (int i = threadIdx.x; i < rows * 8; i += blockDim.x) {
        W[i] = 0.01f * (float)(i & 15);
        G[i] = 1.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = mix(blockIdx.x * 8u + wave + 1u);
    float keep = 0.0f;
    for (int it = 0; it < iters; ++it) {
        s = lcg(s);
        const uint32_t r = mix(s + (lane >> 3)) % (uint32_t)rows;  // 8 random rows per pass
        const int cell = (int)r * 8 + (lane & 7);
        const float oW = W[cell], oG = G[cell];
        float nW = oW + 1e-3f, nG = oG + 1e-6f;
        if (MODE != 2) {
            const double g = 1e-3 * (double)(lane + 1), lr = 0.05 / sqrt((double)oG);
            nW = (float)((double)oW - lr * g);
            nG = (float)((double)oG + g * g);
        }
        if (MODE == 1) {
            W[cell] = nW;
            G[cell] = nG;
        } else {
            atomicAdd(W + cell, nW - oW);
            atomicAdd(G + cell, nG - oG);
        }
        keep += oW;
    }
    if (keep == 123.456f) out[0] = keep;
}

static float *alloc(size_t bytes, bool uncached)
{
    void *p = nullptr;
    if (uncached) CHECK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached));
    else CHECK(hipMalloc(&p, bytes));
    CHECK(hipMemset(p, 0, bytes));
    return (float *)p;
}

template <typename F>
static double time_ms(F launch, int reps = 3)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs\n", prop.gcnArchName, cus);
    float *out = alloc(256, false);
    {   // LDS read-modify-write of 8-cell row slices (`membench lds` runs only this section)
        const int rows = 1128, iters = 20000;
        const size_t smem = (size_t)rows * 8 * 2 * sizeof(float);
        CHECK(hipFuncSetAttribute((const void *)lds_rmw_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        CHECK(hipFuncSetAttribute((const void *)lds_rmw_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        CHECK(hipFuncSetAttribute((const void *)lds_rmw_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const char *names[3] = {"f64 adagrad cell + ds_add_f32 x2", "f64 adagrad cell + ds_write x2", "ds_read x2 + ds_add_f32 x2 only"};
        for (int wgs_per_cu : {1, 2}) {
            for (int threads : {256, 512, 1024}) {
                if (threads > 256) continue;  // (__launch_bounds__(256): the shape the slice kernel uses)
                for (int mode = 0; mode < 3; ++mode) {
                    const int grid = cus * wgs_per_cu;
                    double ms = 0;
                    if (mode == 0) ms = time_ms([&] { lds_rmw_kernel<0><<<grid, threads, smem>>>(rows, iters, out); });
                    if (mode == 1) ms = time_ms([&] { lds_rmw_kernel<1><<<grid, threads, smem>>>(rows, iters, out); });
                    if (mode == 2) ms = time_ms([&] { lds_rmw_kernel<2><<<grid, threads, smem>>>(rows, iters, out); });
                    const double cells = (double)grid * (threads / 64) * iters * 64;
                    printf("lds    %d rows x 8 cells x {W, G} (%zu KB) %d workgroup(s)/CU x %d threads  %-34s: %8.1f G cells/s "
                           "(%.1f G dword read-modify-writes/s; the L2 float-atomic unit: 320 G dwords/s)\n", rows, smem >> 10,
                           wgs_per_cu, threads, names[mode], cells / ms / 1e6, 2 * cells / ms / 1e6);
                }
            }
        }
        if (argc > 1 && !strcmp(argv[1], "lds")) return 0;
    }
    struct Tab { const char *name; uint32_t rows; };
    // 256-B rows: ML-20M items (6.8 MB), ML-20M users (35 MB), C4 items (1.28 GB)
    const Tab tabs256[] = {{"26744 rows x 256 B (6.8 MB)", 26744u}, {"138493 rows x 256 B (35 MB)", 138493u},
                           {"5000000 rows x 256 B (1.28 GB)", 5000000u}};
    for (int unc = 0; unc < 2; ++unc) {
        for (const Tab &t : tabs256) {
            float *tab = alloc((size_t)t.rows * 256, unc);
            for (int wpc : {8, 16, 32}) {
                const int grid = cus * wpc / 4, iters = 400;
                auto run = [&](auto kernel, int burst) {
                    double ms = time_ms([&] { kernel<<<grid, 256>>>(tab, t.rows, iters, out); });
                    double rows_n = (double)grid * 4 * iters * burst * 4;
                    printf("gather %-34s %s %2d waves/CU burst %2d: %7.1f GB/s  (%.2f G rows/s)\n", t.name,
                           unc ? "uncached" : "cached  ", wpc, burst, rows_n * 256 / ms / 1e6, rows_n / ms / 1e6);
                };
                run(gather_kernel<16, 4>, 4);
                run(gather_kernel<16, 12>, 12);
                if (wpc == 8) run(gather_kernel<16, 24>, 24);
            }
            // atomics on the same table
            for (int wpc : {8, 16}) {
                const int grid = cus * wpc / 4, iters = 200;
                for (int mode = 0; mode < 2; ++mode) {
                    double ms = time_ms([&] { atomic_kernel<6><<<grid, 256>>>(tab, t.rows, 64, iters, mode); });
                    double rows_n = (double)grid * 4 * iters * 6;
                    printf("%s %-34s %s %2d waves/CU burst  6: %7.1f GB/s payload (%.2f G row-ops/s = %.2f G 128-B "
                           "line-ops/s)\n", mode ? "store " : "atomic", t.name, unc ? "uncached" : "cached  ", wpc,
                           rows_n * 256 / ms / 1e6, rows_n / ms / 1e6, 2 * rows_n / ms / 1e6);
                }
                double ms = time_ms([&] { rmw_kernel<6><<<grid, 256>>>(tab, t.rows, iters); });
                double rows_n = (double)grid * 4 * iters * 6;
                printf("rmw    %-34s %s %2d waves/CU burst  6: %7.1f GB/s read + the same atomically added (%.2f G "
                       "rows/s)\n", t.name, unc ? "uncached" : "cached  ", wpc, rows_n * 256 / ms / 1e6, rows_n / ms / 1e6);
                if (wpc == 8) {
                    const char *fl[5] = {"f32 add, returning", "u32 add", "f64 add (512-B rows)", "u64 CAS (512-B rows)", "u32 max"};
                    for (int f = 0; f < 5; ++f) {
                        ms = time_ms([&] { atomic_flavour_kernel<6><<<grid, 256>>>(tab, t.rows, iters, f, out); });
                        const double bytes = (f == 2 || f == 3) ? 512.0 : 256.0;
                        printf("atomic %-34s %s %2d waves/CU %-22s: %7.1f GB/s payload (%.2f G row-ops/s = %.1f G lane-ops/s)\n",
                               t.name, unc ? "uncached" : "cached  ", wpc, fl[f], rows_n * bytes / ms / 1e6, rows_n / ms / 1e6,
                               64 * rows_n / ms / 1e6);
                    }
                }
            }
            CHECK(hipFree(tab));
        }
    }
    // 512-B rows (d = 128): C3's 1128 hot tag rows, its 27872-row item table, C5's 1 M rows
    const Tab tabs512[] = {{"1128 rows x 512 B (0.6 MB, hot)", 1128u}, {"27872 rows x 512 B (14 MB)", 27872u},
                           {"1000000 rows x 512 B (512 MB)", 1000000u}};
    for (int unc = 0; unc < 2; ++unc) {
        for (const Tab &t : tabs512) {
            float *tab = alloc((size_t)t.rows * 512, unc);
            for (int wpc : {8, 16}) {
                const int grid = cus * wpc / 4, iters = 200;
                double ms = time_ms([&] { gather_kernel<32, 10><<<grid, 256>>>(tab, t.rows, iters, out); });
                double rows_n = (double)grid * 4 * iters * 10 * 2;
                printf("gather %-34s %s %2d waves/CU burst 10: %7.1f GB/s  (%.2f G rows/s)\n", t.name,
                       unc ? "uncached" : "cached  ", wpc, rows_n * 512 / ms / 1e6, rows_n / ms / 1e6);
                ms = time_ms([&] { atomic_kernel<10><<<grid, 256>>>(tab, t.rows, 128, iters, 0); });
                rows_n = (double)grid * 4 * iters * 10;
                printf("atomic %-34s %s %2d waves/CU burst 10: %7.1f GB/s payload (%.2f G row-ops/s = %.2f G 128-B "
                       "line-ops/s)\n", t.name, unc ? "uncached" : "cached  ", wpc, rows_n * 512 / ms / 1e6,
                       rows_n / ms / 1e6, 4 * rows_n / ms / 1e6);
            }
            CHECK(hipFree(tab));
        }
    }
    return 0;
}
