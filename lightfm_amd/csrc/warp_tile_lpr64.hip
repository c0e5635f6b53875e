// warp_tile_lpr64.hip -- instantiations of the lane-group tile kernel with 64 lanes per row
// (1 interaction per wavefront pass); see warp_tile_kernel.hpp.
#include "warp_tile_kernel.hpp"

namespace lfm {

hipError_t launch_tile_lpr64(const FitArgs &a, int vec, int grid, size_t smem, hipStream_t st, int cus,
                             bool timed, int *grid_used)
{
    switch (vec) {
    case 1: return launch_tile_variant<64, 1>(a, grid, smem, st, cus, timed, grid_used);
    case 2: return launch_tile_variant<64, 2>(a, grid, smem, st, cus, timed, grid_used);
    case 4: return launch_tile_variant<64, 4>(a, grid, smem, st, cus, timed, grid_used);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace lfm
