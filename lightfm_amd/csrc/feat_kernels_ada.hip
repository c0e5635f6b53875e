// feat_kernels_ada.hip -- the ADA (adadelta) instantiations of the row-stream epoch kernels (feat_kernel.hpp): a translation
// unit of its own so that they compile in parallel with feat_kernels.hip.
#include "feat_kernel.hpp"

namespace lfm {

hipError_t launch_fit_feat_ada(int loss, const FitArgs &a, int grid, int block, size_t smem, hipStream_t st, int cus,
                               int *grid_used)
{
    if (a.m.d <= 64) return launch_feat_ada_nc<1>(loss, a, grid, block, smem, st, cus, grid_used);
    if (a.m.d <= 128) return launch_feat_ada_nc<2>(loss, a, grid, block, smem, st, cus, grid_used);
    return hipErrorInvalidValue;
}

}  // namespace lfm
