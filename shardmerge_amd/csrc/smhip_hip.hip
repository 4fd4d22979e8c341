// smhip_hip.hip - the gfx950 build: DeviceExec (one real work-group per block),
// HipBackend (launches on the caller's stream, hipMalloc'd workspace, HIP-event
// profiling) and the C ABI of include/shardmerge_hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see Makefile).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "sm_pipeline.hpp"

#define SM_VERSION_STRING "shardmerge-hip 0.1 (gfx950)"

namespace smhip {

struct DeviceExec {
    template <class S> struct State { using value_type = S; S s; };
    __device__ __forceinline__ int bid() const { return blockIdx.x; }
    __device__ __forceinline__ int nthreads() const { return blockDim.x; }
    __device__ __forceinline__ float* lds() {
        extern __shared__ __attribute__((aligned(16))) float sm_dyn_lds[];
        return sm_dyn_lds;
    }
    template <class S> __device__ __forceinline__ void init(State<S>&) {}
    template <class S, class F> __device__ __forceinline__ void each(State<S>& st, F&& f) { f((int)threadIdx.x, st.s); }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ void lds_atomic_add(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
    __device__ __forceinline__ void global_atomic_add(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
    __device__ __forceinline__ void global_atomic_add_u32(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
    __device__ __forceinline__ void global_atomic_or_u32(uint32_t* p, uint32_t v) { atomicOr(p, v); }

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

template <class K>
__global__ void __launch_bounds__(1024) sm_kernel(const typename K::Params p) {
    DeviceExec ex;
    K::run(ex, p);
}

struct HipBackend {
    int device;
    hipError_t first_err = hipSuccess;
    std::string err_msg;
    bool profiling = false;
    std::vector<ProfEntry> prof;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    explicit HipBackend(int dev) : device(dev) {
        check(hipSetDevice(dev), "hipSetDevice");
        if (ok()) {
            check(hipEventCreate(&ev0), "hipEventCreate");
            check(hipEventCreate(&ev1), "hipEventCreate");
        }
    }
    ~HipBackend() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
    void check(hipError_t e, const char* what) {
        if (e != hipSuccess && first_err == hipSuccess) {
            first_err = e;
            err_msg = std::string(what) + ": " + hipGetErrorString(e);
        }
    }
    bool ok() const { return first_err == hipSuccess; }
    std::string error() const { return err_msg; }

    void* alloc(size_t n) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, n ? n : 16);
        if (e != hipSuccess) { check(e, "hipMalloc"); return nullptr; }
        return p;
    }
    void free(void* p) { check(hipFree(p), "hipFree"); }
    void memset(void* p, int v, size_t n, void* s) { check(hipMemsetAsync(p, v, n, (hipStream_t)s), "hipMemsetAsync"); }
    void sync(void* s) { check(hipStreamSynchronize((hipStream_t)s), "hipStreamSynchronize"); }
    void d2h(void* dst, const void* src, size_t n, void* s) {
        check(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (hipStream_t)s), "hipMemcpyAsync d2h");
        sync(s);
    }
    void h2d(void* dst, const void* src, size_t n, void* s) {
        check(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)s), "hipMemcpyAsync h2d");
        sync(s);     // the source is a host temporary
    }

    template <class K>
    void launch(int grid, int block, size_t lds_bytes, const typename K::Params& p, void* s) {
        if (!ok()) return;
        static size_t lds_set = 0;                 // per kernel instantiation
        if (lds_bytes > lds_set) {
            check(hipFuncSetAttribute((const void*)sm_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes),
                  "hipFuncSetAttribute");
            lds_set = lds_bytes;
        }
        hipStream_t st = (hipStream_t)s;
        if (profiling) check(hipEventRecord(ev0, st), "hipEventRecord");
        hipLaunchKernelGGL(sm_kernel<K>, dim3(grid), dim3(block), lds_bytes, st, p);
        check(hipGetLastError(), K::name());
        if (profiling) {
            check(hipEventRecord(ev1, st), "hipEventRecord");
            check(hipEventSynchronize(ev1), "hipEventSynchronize");
            float ms = 0.f;
            check(hipEventElapsedTime(&ms, ev0, ev1), "hipEventElapsedTime");
            ProfEntry* e = nullptr;
            for (auto& q : prof) if (q.name == K::name()) e = &q;
            if (!e) { prof.push_back(ProfEntry{K::name(), 0, 0.0}); e = &prof.back(); }
            e->launches++; e->ms += ms;
        }
    }
    void profile_enable(bool on) { profiling = on; }
    void profile_reset() { prof.clear(); }
    int profile_count() const { return (int)prof.size(); }
    bool profile_get(int i, const char** name, uint64_t* launches, double* ms) const {
        if (i < 0 || i >= (int)prof.size()) return false;
        if (name) *name = prof[i].name.c_str();
        if (launches) *launches = prof[i].launches;
        if (ms) *ms = prof[i].ms;
        return true;
    }
};

}  // namespace smhip

#define SM_BACKEND smhip::HipBackend
#include "sm_capi.inc"
