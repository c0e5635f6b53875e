// fit_kernels_wide.hip -- the one-interaction-per-wavefront epoch kernels (fit_kernels.hip) for 512 < d <= 1 024: sixteen
// coordinates per lane (NC = 16).  A translation unit of its own so that the instantiations compile beside fit_kernels.hip.
#define LFM_FIT_WIDE_UNIT 1
#include "fit_kernels.hip"
