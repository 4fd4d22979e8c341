// smhip_hip.hip - the gfx950 build: DeviceExec (one real work-group per block),
// HipBackend (launches on the caller's stream, hipMalloc'd workspace, HIP-event
// profiling) and the C ABI of include/shardmerge_hip.h.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see Makefile).
#include <hip/hip_runtime.h>
#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "smhip_device.hpp"

#define SM_VERSION_STRING "shardmerge-hip 0.1 (gfx950)"

namespace smhip {

#define SM_DECL_EXTERN(...) extern template __global__ void sm_kernel<__VA_ARGS__>(const typename __VA_ARGS__::Params);
#define SM_DECL_PLAN(...) SM_FFT_KERNELS_OF(SM_DECL_EXTERN, __VA_ARGS__)
SM_STATIC_PLANS(SM_DECL_PLAN)
#undef SM_DECL_PLAN
SM_SIDE_KERNELS_0(SM_DECL_EXTERN)
SM_SIDE_KERNELS_1(SM_DECL_EXTERN)
SM_SIDE_KERNELS_2(SM_DECL_EXTERN)
SM_FFT_KERNELS_OF(SM_DECL_EXTERN, DynPlan)
#undef SM_DECL_EXTERN

struct HipBackend {
    int device;
    hipError_t first_err = hipSuccess;
    std::string err_msg;
    bool profiling = false;
    bool trace_launches = getenv("SMHIP_TRACE_LAUNCHES") != nullptr;
    std::vector<ProfEntry> prof;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t aux = nullptr;            // side stream (norm_mode = reference_cpu: the norm emulation beside the row pass)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

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
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (aux) (void)hipStreamDestroy(aux);
    }
    // a second stream of this context: work between fork(s, aux) and join(aux, s) runs beside what the caller's
    // stream s does in between, and is complete (in stream order) for everything s does after the join
    void* aux_stream() {
        if (!aux && ok()) {
            // (SMHIP_AUX_PRIORITY=1 / 2: highest / lowest stream priority, an experiment knob - measured on MI355X the
            //  walker gets onto the CUs between the row pass's work-groups at any priority, tools/ab_aux.sh)
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
            const char* pr = getenv("SMHIP_AUX_PRIORITY");
            const int prio = (pr && pr[0] == '1') ? hi : ((pr && pr[0] == '2') ? lo : 0);
            if (hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, prio) != hipSuccess) { aux = nullptr; (void)hipGetLastError(); return nullptr; }
            check(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming), "hipEventCreate");
            check(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming), "hipEventCreate");
        }
        return (void*)aux;
    }
    void fork(void* s, void* a) {
        check(hipEventRecord(ev_fork, (hipStream_t)s), "hipEventRecord");
        check(hipStreamWaitEvent((hipStream_t)a, ev_fork, 0), "hipStreamWaitEvent");
    }
    void join(void* a, void* s) {
        check(hipEventRecord(ev_join, (hipStream_t)a), "hipEventRecord");
        check(hipStreamWaitEvent((hipStream_t)s, ev_join, 0), "hipStreamWaitEvent");
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
    // pinned host memory the device can write (same pointer on both sides)
    void* alloc_host(size_t n) {
        void* p = nullptr;
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e != hipSuccess) { check(e, "hipHostMalloc"); return nullptr; }
        ::memset(p, 0, n);
        return p;
    }
    void free_host(void* p) { check(hipHostFree(p), "hipHostFree"); }
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
        static std::atomic<size_t> lds_set{0};     // per kernel instantiation; contexts of several threads share it
        if (lds_bytes > lds_set.load(std::memory_order_relaxed)) {
            check(hipFuncSetAttribute((const void*)sm_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes),
                  "hipFuncSetAttribute");
            size_t seen = lds_set.load(std::memory_order_relaxed);
            while (seen < lds_bytes && !lds_set.compare_exchange_weak(seen, lds_bytes, std::memory_order_relaxed)) {}
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
            if (trace_launches) fprintf(stderr, "[smhip] %-20s grid %6d block %4d lds %6zu  %9.1f us\n", K::name(), grid, block, lds_bytes, ms * 1e3);
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
