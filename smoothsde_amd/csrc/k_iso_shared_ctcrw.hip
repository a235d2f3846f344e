// k_iso_shared_ctcrw.hip -- the shared-covariance kernels of CTCRW (k_iso_shared.inc), one translation unit per model.
#define SSDE_SHARED_HAS_P2 1
#include "k_iso_shared.inc"

namespace ssde {
hipError_t launch_iso_shared_ctcrw(int d, const IsoArgs& a, const ReduceArgs& r, dim3 grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, bool deep) {
    return launch_shared_model<M_CTCRW>(d, a, r, grid, s, ev0, ev1, deep);
}
}  // namespace ssde
