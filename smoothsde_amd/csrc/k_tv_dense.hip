// k_tv_dense.hip -- full-covariance lanes (per-row H_array, general P0) of the lane = gradient direction recursion
// kernel: instantiations of k_tv_filter.hpp with TvDenseOps (ssde_tv.hpp).  Own translation unit for compile time.
#include "k_tv_filter.hpp"

namespace ssde {

hipError_t launch_tv_filter_dense(const TvArgs& a, bool want_grad, hipStream_t s) {
    if (a.model == M_ESEAL) {                       // scalar lipid-mass filter (TvEsealOps), no report variant
        if (a.n_items == 0) return hipSuccess;
        dim3 grid((a.n_items + WG_WAVES - 1) / WG_WAVES), block(WG_WAVES * WAVE);
        if (want_grad) hipLaunchKernelGGL((tv_filter_kernel<TvEsealOps, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((tv_filter_kernel<TvEsealOps, false, false>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    SSDE_TV_LAUNCH_FILTER(TvDenseOps)
}

}  // namespace ssde
