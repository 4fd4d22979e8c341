// smhip_inst.hip - explicit instantiation of the five transform kernels for ONE
// static plan, selected with -DSM_PLAN_INDEX=<i> (order of SM_STATIC_PLANS).
#include "smhip_device.hpp"

namespace smhip {
#define SM_INST(...) template __global__ void sm_kernel<__VA_ARGS__>(const typename __VA_ARGS__::Params);
template <int I> struct PlanAt;
#define SM_COUNT_PLAN(...) , __VA_ARGS__
template <int I, class... Ps> struct Pick;
template <class P0, class... Ps> struct Pick<0, P0, Ps...> { using type = P0; };
template <int I, class P0, class... Ps> struct Pick<I, P0, Ps...> { using type = typename Pick<I - 1, Ps...>::type; };
using ThisPlan = Pick<SM_PLAN_INDEX SM_STATIC_PLANS(SM_COUNT_PLAN)>::type;
SM_FFT_KERNELS_OF(SM_INST, ThisPlan)
}  // namespace smhip
