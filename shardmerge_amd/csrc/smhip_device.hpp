// smhip_device.hpp - DeviceExec (one real gfx950 work-group per block) and the
// generic __global__ entry point; shared by smhip_hip.hip and smhip_inst.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "sm_pipeline.hpp"

namespace smhip {

struct DeviceExec {
    template <class S> struct State { using value_type = S; S s; };
    __device__ __forceinline__ int bid() const { return blockIdx.x; }
    __device__ __forceinline__ int nthreads() const { return blockDim.x; }
    __device__ __forceinline__ int nblocks() const { return gridDim.x; }
    __device__ __forceinline__ float* lds() {
        extern __shared__ __attribute__((aligned(16))) float sm_dyn_lds[];
        return sm_dyn_lds;
    }
    template <class S> __device__ __forceinline__ void init(State<S>&) {}
    template <class S, class F> __device__ __forceinline__ void each(State<S>& st, F&& f) { f((int)threadIdx.x, st.s); }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ void lds_atomic_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
    __device__ __forceinline__ uint32_t lds_atomic_add_ret(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
    __device__ __forceinline__ uint32_t global_atomic_add_ret_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
    __device__ __forceinline__ void global_atomic_add(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
    __device__ __forceinline__ void global_atomic_add_u32(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
    __device__ __forceinline__ void global_atomic_or_u32(uint32_t* p, uint32_t v) { atomicOr(p, v); }

    // dst(lane)[i] = src(lane + DELTA)[i], i < N dwords, inside the 64-lane wave (lanes past its end keep their own)
    template <int DELTA, int N, class S, class Src, class Dst>
    __device__ __forceinline__ void wave_shift_down(State<S>& st, Src src, Dst dst) {
        const uint32_t* a = src(st.s);
        uint32_t* b = dst(st.s);
#pragma unroll
        for (int i = 0; i < N; ++i) b[i] = (uint32_t)__shfl_down((int)a[i], DELTA, 64);
    }
    // Lanes t and t ^ 1 trade one float each, NP times: of the pair (a, b) that get(s, k) names, the even lane gives
    // its b and takes the partner's a into it, the odd lane gives its a and takes the partner's b into it (one select,
    // one quad-permute DPP move, two selects; no LDS).  get(S&, integral_constant<int, k>) -> {float& a, float& b}.
    template <int NP, class S, class Get>
    __device__ __forceinline__ void lane_pair_trade(State<S>& st, Get get) {
        const bool odd = (threadIdx.x & 1u) != 0;
        static_for<0, NP>([&](auto k_c) {
            auto pr = get(st.s, k_c);
            const float give = odd ? pr.a : pr.b;
            const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            pr.a = odd ? got : pr.a;
            pr.b = odd ? pr.b : got;
        });
    }
    // sum NV doubles (State::red) over the work-group; f(total) runs on thread 0.
    // Uses the first LDS_SCRATCH_FLOATS of LDS; ends with a barrier.
    template <int NV, class S, class F>
    __device__ __forceinline__ void block_sum(State<S>& st, F&& f) {
        double v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = st.s.red[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q] += __shfl_down(v[q], off, 64);
        }
        double* scratch = (double*)lds();
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int nw = (blockDim.x + 63) >> 6;
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NV; ++q) scratch[wave * NV + q] = v[q];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) tot[q] = 0.0;
            for (int w = 0; w < nw; ++w) {
#pragma unroll
                for (int q = 0; q < NV; ++q) tot[q] += scratch[w * NV + q];
            }
            f((const double*)tot);
        }
        __syncthreads();
    }
};

// Each kernel tag states its largest work-group (max_threads) and the waves per SIMD it
// is compiled for.  Default (1024, 4): at most 128 VGPRs.  The forward row pass is the
// one kernel that measured faster on MI355X with 3 waves per SIMD (168 VGPRs, no scratch
// spills, -10..-17 % time); the column passes lose 30 % that way and stay at 4.
template <class K>
__global__ void __launch_bounds__(K::max_threads, K::waves) sm_kernel(const typename K::Params p) {
    DeviceExec ex;
    K::run(ex, p);
}

// static-plan transform kernels are compiled in separate translation units
// (smhip_inst.hip, one per plan) so that the build parallelises
#define SM_FFT_KERNELS_OF(X, ...)                      \
    X(KF1<__VA_ARGS__>) X(KF2<__VA_ARGS__>) X(KI1x1<__VA_ARGS__>) X(KI1x2<__VA_ARGS__>) X(KI2<__VA_ARGS__>) X(KF2S<__VA_ARGS__>) \
    X(KF1Q<__VA_ARGS__>) X(KF2Q<__VA_ARGS__>) X(KF2SQ<__VA_ARGS__>) X(KI1x1Q<__VA_ARGS__>) X(KI1x2Q<__VA_ARGS__>) \
    X(KPair1d<__VA_ARGS__>) X(KF1B<__VA_ARGS__>) X(KI2B<__VA_ARGS__>)

}  // namespace smhip
