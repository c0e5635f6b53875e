// warp_tile_ahead.hip -- instantiation and launcher of the gather-ahead variant of the lane-group tile
// kernel (warp_tile_ahead.hpp): max_sampled = 10 candidates in one batch, the shape every BASELINE
// configuration trains with (LightFM's default max_sampled).
#include <stdlib.h>

#include "warp_tile_ahead.hpp"
#include "warp_tile_narrow.hpp"

namespace lfm {

// LIGHTFM_AMD_TILE_NARROW=1 (off by default): rows of up to 16 floats take the VEC = 1 instantiation -- a quarter of the LDS per
// interaction, 16 384-20 480 instead of 12 288 interactions in flight.  Built for the reference's default width and measured
// SLOWER (C2 at d = 10: 1.66 / 1.62 G/s at 4 / 5 workgroups per CU against 1.71 G/s; d = 16: 1.78 against 1.82;
// profiles/r06_narrow_tile_ab.txt): at this width the kernel is bound by neither latency nor bytes but by the atomic units'
// LINE operations -- a 12-float row costs the line operation a 32-float half row costs (~10 G/s chip-wide), and an update is
// 10 of them (4 embedding rows, 6 bias cells): 0.71 G updates/s x 10 = 0.7 of that rate.  What would help is fewer line
// operations per update (W, G, b and bG of a narrow row in ONE 128-byte line), not more wavefronts.
static bool narrow_rows_enabled()
{
    static const bool on = [] { const char *e = getenv("LIGHTFM_AMD_TILE_NARROW"); return e && atoi(e) != 0; }();
    return on;
}

// 0 if (d, max_sampled, first batch) is outside the variant's scope, else its LDS bytes per workgroup
size_t warp_tile_ahead_smem(int d, int max_sampled, int first_batch)
{
    if (d < 4 || d > 64 || (d & 3) != 0 || max_sampled != 10 || first_batch != 10) return 0;
    return d <= 16 && narrow_rows_enabled() ? tile_ahead_smem<10, 1>() : tile_ahead_smem<10>();
}

// The narrow-model kernel (warp_tile_narrow.hpp: two interactions per 16-lane group, eight per wavefront pass): LDS bytes per
// workgroup, 0 outside its scope (LIGHTFM_AMD_TILE_PAIRS=0 keeps narrow models on the wide kernel)
size_t warp_tile_narrow_smem(int d, int max_sampled, int first_batch, int64_t n_items)
{
    static const bool on = [] { const char *e = getenv("LIGHTFM_AMD_TILE_PAIRS"); return !e || atoi(e) != 0; }();
    if (!on || d < 4 || d > 16 || (d & 3) != 0 || max_sampled != 10 || first_batch != 10 || n_items * (int64_t)d >= (1ll << 30)) return 0;
    return tile_narrow_smem();
}

hipError_t launch_fit_warp_tile_narrow(const FitArgs &a, int grid, hipStream_t st, int cus, int per_cu_cap, int *grid_used)
{
    void (*kernel)(FitArgs) = a.user_store ? fit_warp_tile_narrow_kernel<10, true> : fit_warp_tile_narrow_kernel<10, false>;
    if (a.rp[0]) kernel = a.user_store ? fit_warp_tile_narrow_kernel<10, true, true> : fit_warp_tile_narrow_kernel<10, false, true>;  // row pairs
    if (a.rp[0] && a.rp_bias) kernel = a.user_store ? fit_warp_tile_narrow_kernel<10, true, true, true> : fit_warp_tile_narrow_kernel<10, false, true, true>;  // ... with the bias cells
    const size_t smem = tile_narrow_smem();
    if (cus > 0) {
        int per_cu = occupancy_cached(kernel, 256, smem);
        if (per_cu_cap > 0) per_cu = std::min(per_cu, per_cu_cap);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    kernel<<<grid, 256, smem, st>>>(a);
    return hipGetLastError();
}

hipError_t launch_fit_warp_tile_ahead(const FitArgs &a, int grid, hipStream_t st, int cus, int *grid_used)
{
    void (*kernel)(FitArgs) = a.shards.n > 0 ? fit_warp_tile_ahead_kernel<10, true> : fit_warp_tile_ahead_kernel<10, false>;
    if (a.shards.n == 0 && a.user_store) kernel = fit_warp_tile_ahead_kernel<10, false, true>;  // user rows by plain stores
    size_t smem = tile_ahead_smem<10>();
    if (a.shards.n == 0 && a.m.d <= 16 && narrow_rows_enabled()) {  // rows of up to 16 floats: a quarter of the LDS per interaction
        kernel = a.user_store ? fit_warp_tile_ahead_kernel<10, false, true, 1> : fit_warp_tile_ahead_kernel<10, false, false, 1>;
        smem = tile_ahead_smem<10, 1>();
    }
    if (cus > 0) {
        const int per_cu = occupancy_cached(kernel, 256, smem);
        if (per_cu > 0) grid = std::min(grid, per_cu * cus);
    }
    if (grid_used) *grid_used = grid;
    kernel<<<grid, 256, smem, st>>>(a);
    return hipGetLastError();
}

}  // namespace lfm
