// warp_tile_lpr16.hip -- instantiations of the lane-group tile kernel with 16 lanes per row
// (4 interactions per wavefront pass); see warp_tile_kernel.hpp.
#include "warp_tile_kernel.hpp"

namespace lfm {

hipError_t launch_tile_lpr16(const FitArgs &a, int vec, int grid, size_t smem, hipStream_t st, int cus,
                             bool timed, int *grid_used, bool dma4)
{
    switch (vec) {
    case 4:
        if (dma4) return launch_tile_variant<16, 4, true>(a, grid, smem, st, cus, timed, grid_used);
        return launch_tile_variant<16, 4>(a, grid, smem, st, cus, timed, grid_used);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace lfm
