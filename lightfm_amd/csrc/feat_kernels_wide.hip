// feat_kernels_wide.hip -- the row-stream epoch kernels (feat_kernel.hpp) for 128 < d <= 256: four components per lane
// (NC = 4).  A translation unit of its own so that the instantiations compile in parallel with feat_kernels.hip.
#include "feat_kernel.hpp"

namespace lfm {

hipError_t launch_fit_feat_wide(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                                int *grid_used)
{
    return launch_feat_nc<4>(loss, a, grid, block, smem, st, cus, grid_used, false);
}

}  // namespace lfm
