// feat_kernels_hot.hip -- the HOT instantiations of the row-stream epoch kernels (feat_kernel.hpp): models with a hot set
// (shared item-feature rows accumulated in LDS slices, hot_slices.hip).  A translation unit of its own so that the
// instantiations compile in parallel with feat_kernels.hip.
#include "feat_kernel.hpp"

namespace lfm {

hipError_t launch_fit_feat_hot(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                               int *grid_used)
{
    if (a.m.d <= 64) return launch_feat_hot_nc<1>(loss, a, grid, block, smem, st, cus, grid_used);
    if (a.m.d <= 128) return launch_feat_hot_nc<2>(loss, a, grid, block, smem, st, cus, grid_used);
    return hipErrorInvalidValue;
}

}  // namespace lfm
