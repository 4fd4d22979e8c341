// smhip_side.hip - explicit instantiation of one group of kernels, selected with -DSM_SIDE_GROUP=<g>
// (SM_SIDE_KERNELS_<g> in sm_pipeline.hpp; groups 3..6: the transform kernels for run-time planned lengths).
#include "smhip_device.hpp"

namespace smhip {
#define SM_INST(...) template __global__ void sm_kernel<__VA_ARGS__>(const typename __VA_ARGS__::Params);
#if SM_SIDE_GROUP == 0
SM_SIDE_KERNELS_0(SM_INST)
#elif SM_SIDE_GROUP == 1
SM_SIDE_KERNELS_1(SM_INST)
#elif SM_SIDE_GROUP == 2
SM_SIDE_KERNELS_2(SM_INST)
#elif SM_SIDE_GROUP == 3
SM_INST(KF1<DynPlan>) SM_INST(KF1Q<DynPlan>) SM_INST(KI2<DynPlan>) SM_INST(KPair1d<DynPlan>)
#elif SM_SIDE_GROUP == 4
SM_INST(KF2<DynPlan>) SM_INST(KF2Q<DynPlan>) SM_INST(KF2S<DynPlan>) SM_INST(KF2SQ<DynPlan>)
#elif SM_SIDE_GROUP == 5
SM_INST(KI1x1<DynPlan>) SM_INST(KI1x2<DynPlan>) SM_INST(KI1x1Q<DynPlan>) SM_INST(KI1x2Q<DynPlan>)
#elif SM_SIDE_GROUP == 6
SM_INST(KF1B<DynPlan>) SM_INST(KI2B<DynPlan>)
#else
#error "SM_SIDE_GROUP must be 0..6"
#endif
}  // namespace smhip
