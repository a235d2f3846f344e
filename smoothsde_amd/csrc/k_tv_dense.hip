// k_tv_dense.hip -- full-covariance lanes (per-row H_array, general P0) of the lane = gradient direction recursion
// kernel: instantiations of k_tv_filter.hpp with TvDenseOps (ssde_tv.hpp).  Own translation unit for compile time.
#include "k_tv_filter.hpp"

namespace ssde {

hipError_t launch_tv_filter_dense(const TvArgs& a, bool want_grad, hipStream_t s) {
    SSDE_TV_LAUNCH_FILTER(TvDenseOps)
}

}  // namespace ssde
