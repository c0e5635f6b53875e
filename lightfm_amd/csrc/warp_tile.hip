// warp_tile.hip -- host side of the lane-group WARP tile kernel (warp_tile_kernel.hpp): the
// record packing kernel, tile geometry and the dispatcher over the instantiated variants.
#include "device.hpp"
#include "kernels.hpp"

namespace lfm {

// One 16-byte record per example instead of four 4-byte arrays: the epoch kernel then makes
// one random access per interaction into the (shuffled) COO instead of four.
__global__ void pack_records_kernel(const int32_t *user_ids, const int32_t *item_ids, const float *Y,
                                    const float *weight, int64_t n, int4 *out)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = t; j < n; j += st)
        out[j] = make_int4(user_ids[j], item_ids[j], __float_as_int(Y[j]), __float_as_int(weight[j]));
}

hipError_t launch_pack_records(const int32_t *user_ids, const int32_t *item_ids, const float *Y,
                               const float *weight, int64_t n, void *out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    int grid = (int)std::min<int64_t>(8192, (n + 255) / 256);
    pack_records_kernel<<<grid, 256, 0, st>>>(user_ids, item_ids, Y, weight, n, (int4 *)out);
    return hipGetLastError();
}

// Tile geometry for `ng` interactions per wavefront pass (1, 2 or 4): lanes per row = 64 / ng,
// a lane carries vec = 1, 2 or 4 consecutive floats of a row.  Returns the LDS bytes per
// 256-thread workgroup, or 0 if (d, max_sampled, ng) is outside what the kernel supports.
// dma4 (ng = 4, vec = 4 only): the candidate-major LDS-DMA layout of warp_tile_kernel.hpp -- rg rows of
// (4 * 64 + 4) floats plus four user rows of 64 + 4.
size_t warp_tile_geometry(int d, int max_sampled, int ng, int *rows, int *stride, int *vec, bool dma4)
{
    if (d < 4 || (d & 3) != 0 || max_sampled < 1 || (ng != 1 && ng != 2 && ng != 4)) return 0;
    const int lpr = WAVE / ng;
    int v = 0;
    for (int c : {1, 2, 4})
        if (!v && lpr * c >= d && lpr * c >= WAVE) v = c;
    if (!v) return 0;  // d too wide for this many interactions per pass
    const int ts = d + 4;  // 16-byte aligned rows, consecutive rows 4 dwords apart in bank phase
    int rg = std::min(max_sampled, lpr - 1) + 1;
    if (dma4) {
        if (ng != 4 || v != 4) return 0;
        *rows = rg;
        *stride = ts;
        *vec = v;
        return (size_t)WAVES_PER_BLOCK * ((size_t)rg * (4 * WAVE + 4) + 4 * (WAVE + 4)) * sizeof(float);
    }
    // at least two workgroups per CU (160 KiB LDS)
    while (rg > 2 && (size_t)WAVES_PER_BLOCK * (ng * rg + ng) * ts * sizeof(float) > 78 * 1024) --rg;
    *rows = rg;
    *stride = ts;
    *vec = v;
    return (size_t)WAVES_PER_BLOCK * (ng * rg + ng) * ts * sizeof(float);
}

hipError_t launch_tile_lpr16(const FitArgs &, int, int, size_t, hipStream_t, int, bool, int *, bool);
hipError_t launch_tile_lpr32(const FitArgs &, int, int, size_t, hipStream_t, int, bool, int *);
hipError_t launch_tile_lpr64(const FitArgs &, int, int, size_t, hipStream_t, int, bool, int *);

hipError_t launch_fit_warp_tile(const FitArgs &a, int ng, int vec, int grid, size_t smem, hipStream_t st,
                                int cus, bool timed, int *grid_used, bool dma4)
{
    switch (ng) {
    case 4: return launch_tile_lpr16(a, vec, grid, smem, st, cus, timed, grid_used, dma4);
    case 2: return launch_tile_lpr32(a, vec, grid, smem, st, cus, timed, grid_used);
    case 1: return launch_tile_lpr64(a, vec, grid, smem, st, cus, timed, grid_used);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace lfm
