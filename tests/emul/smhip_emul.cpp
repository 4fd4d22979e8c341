// smhip_emul.cpp - CPU work-group emulator of the HIP kernels (TEST INFRASTRUCTURE).
//
// Compiles the very same kernel bodies (shardmerge_amd/csrc/sm_kernels.hpp) and
// the same host orchestration (sm_pipeline.hpp) with g++, running each
// work-group's threads one after another between barriers.  It exists so that
// the "not gpu" test tier can check indexing, planning and the tournament logic
// of the product code in a container without a GPU.  It is never loaded by the
// shardmerge_amd package (which binds libshardmerge_hip.so only and fails
// loudly without it); only tests/ build and load it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../shardmerge_amd/csrc/sm_pipeline.hpp"

#define SM_VERSION_STRING "shardmerge-emul 0.1 (cpu work-group emulator, tests only)"

namespace smhip {

struct HostExec {
    template <class S> struct State { using value_type = S; std::vector<S> v; };
    int bid_, nt_;
    std::vector<float>* lds_;
    int nblk_ = 1;
    int nblocks() const { return nblk_; }
    int bid() const { return bid_; }
    int nthreads() const { return nt_; }
    float* lds() { return lds_->data(); }
    template <class S> void init(State<S>& st) { st.v.resize(nt_); }
    template <class S, class F> void each(State<S>& st, F&& f) { for (int t = 0; t < nt_; ++t) f(t, st.v[t]); }
    void sync() {}
    void lds_atomic_add(uint32_t* p, uint32_t v) { *p += v; }
    uint32_t lds_atomic_add_ret(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
    uint32_t global_atomic_add_ret_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
    void global_atomic_add(unsigned long long* p, unsigned long long v) { *p += v; }
    void global_atomic_add_u32(uint32_t* p, uint32_t v) { *p += v; }
    void global_atomic_or_u32(uint32_t* p, uint32_t v) { *p |= v; }
    template <int DELTA, int N, class S, class Src, class Dst> void wave_shift_down(State<S>& st, Src src, Dst dst) {
        for (int t = 0; t < nt_; ++t) {
            const int q = t + DELTA;
            const int from = (q < nt_ && q / 64 == t / 64) ? q : t;
            const uint32_t* a = src(st.v[from]);
            uint32_t* b = dst(st.v[t]);
            for (int i = 0; i < N; ++i) b[i] = a[i];
        }
    }
    template <int NP, class S, class Get> void lane_pair_trade(State<S>& st, Get get) {
        for (int t = 0; t + 1 < nt_; t += 2) {                      // even lane t, odd lane t + 1
            static_for<0, NP>([&](auto k_c) {
                auto ev = get(st.v[t], k_c);
                auto od = get(st.v[t + 1], k_c);
                const float give_e = ev.b, give_o = od.a;
                ev.b = give_o; od.a = give_e;
            });
        }
    }
    template <int NV, class S, class F> void block_sum(State<S>& st, F&& f) {
        double tot[NV];
        for (int q = 0; q < NV; ++q) tot[q] = 0;
        for (int t = 0; t < nt_; ++t)
            for (int q = 0; q < NV; ++q) tot[q] += st.v[t].red[q];
        f((const double*)tot);
    }
};

struct HostBackend {
    std::vector<ProfEntry> prof;
    bool profiling = false;
    explicit HostBackend(int) {}
    bool ok() const { return true; }
    std::string error() const { return ""; }
    void* alloc(size_t n) { void* p = nullptr; if (posix_memalign(&p, 256, n ? n : 16)) return nullptr; ::memset(p, 0xCD, n); return p; }
    void free(void* p) { ::free(p); }
    void* alloc_host(size_t n) { void* p = calloc(1, n ? n : 16); return p; }
    void free_host(void* p) { ::free(p); }
    void memset(void* p, int v, size_t n, void*) { ::memset(p, v, n); }
    void sync(void*) {}
    // (a single sequential "stream": the side stream is the same one)
    void* aux_stream() { static int token; return &token; }
    void fork(void*, void*) {}
    void join(void*, void*) {}
    void d2h(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); }
    void h2d(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); }
    template <class K>
    void launch(int grid, int block, size_t lds_bytes, const typename K::Params& p, void*) {
        if (lds_bytes > 160 * 1024) { fprintf(stderr, "emul: %s wants %zu bytes of LDS (> 160 KiB)\n", K::name(), lds_bytes); abort(); }
        if (block > 1024 || block < 1) { fprintf(stderr, "emul: %s block size %d\n", K::name(), block); abort(); }
        std::vector<float> lds(lds_bytes / 4 + 16);
        for (int b = 0; b < grid; ++b) {
            // poison LDS so that reads of never-written words show up as NaNs
            for (auto& x : lds) x = std::nanf("");
            HostExec ex{b, block, &lds, grid};
            K::run(ex, p);
        }
        if (profiling) {
            ProfEntry* e = nullptr;
            for (auto& q : prof) if (q.name == K::name()) e = &q;
            if (!e) { prof.push_back(ProfEntry{K::name(), 0, 0.0}); e = &prof.back(); }
            e->launches++;
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

#define SM_BACKEND smhip::HostBackend
#include "../../shardmerge_amd/csrc/sm_capi.inc"
