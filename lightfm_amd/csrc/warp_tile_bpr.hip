// warp_tile_bpr.hip -- fit_bpr (PYX:1074-1182) and fit_logistic (PYX:694-781) of identity models wider than the lane-group
// kernels of logistic_tile.hip take: the BPR / logistic instantiations of the lane-group tile kernel (warp_tile_kernel.hpp,
// LOSS = LFM_LOSS_BPR_ID / LFM_LOSS_LOGISTIC_ID) -- four interactions per
// wavefront pass with rows memory -> LDS by LDS-DMA for d <= 64, two for d <= 128, one for d <= 256.  A unit of its own: the
// instantiations compile beside the WARP ones.
#include "warp_tile_kernel.hpp"

namespace lfm {

hipError_t launch_fit_bpr_wide_tile(const FitArgs &a, int ng, int vec, int grid, size_t smem, hipStream_t st, int cus,
                                    int *grid_used, bool dma4, bool logistic)
{
    if (vec != 4) return hipErrorInvalidValue;
    switch (ng) {
    case 4:
        if (dma4) return launch_tile_bpr_variant<16, 4, true>(a, grid, smem, st, cus, grid_used, logistic);
        return launch_tile_bpr_variant<16, 4>(a, grid, smem, st, cus, grid_used, logistic);
    case 2: return launch_tile_bpr_variant<32, 4>(a, grid, smem, st, cus, grid_used, logistic);
    case 1: return launch_tile_bpr_variant<64, 4>(a, grid, smem, st, cus, grid_used, logistic);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace lfm
