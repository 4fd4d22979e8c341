// fftproto.hip - go / no-go probe for the NEXT engine (round 4; not product code): an 8192-point complex FFT per
// 256-thread group, (a) as the product's engine does it (fft_engine.hpp: Stockham 32 x 16 x 16, four exchanges through
// LDS with two work-group barriers each and per component) and (b) with WAVE-LOCAL sub-transforms:
//     8192 = 4 waves x 2048;  2048 = 32 (in registers) x 64 (across the wave's lanes, 2 x 32: registers + one lane-pair step)
//   step 1  lane l of wave w holds x[4 (l + 64 j) + w], j < 32: a 32-point DFT over j in registers, times W_2048^(l k1)
//   step 2  transpose through a WAVE-PRIVATE piece of LDS (a wave's LDS operations complete in order: no barrier):
//           lane (k1, par) takes l = par + 2 i, i < 32, of problem k1
//   step 3  32-point DFT over i in registers, times W_64^(par k2'), butterfly with the partner lane (one DPP move per value)
//   step 4  ONE work-group exchange (the only barrier pair): thread t takes F_w[k], w < 4, for k = t + 256 r, r < 8,
//           multiplies by W_8192^(w k) and does the radix-4 step: X[k + 2048 q]
// 128 LDS stores + 128 loads per lane and 2 barriers against 256 + 256 and 16; the same two 32-point register DFTs
// and a radix-4 against 32 x 16 x 16.
// The probe: contiguous loads, one transform, contiguous stores per group, G = 2 groups per work-group as in the
// product's column pass; ONE round of two work-groups per CU (every phase exposed) and sixteen (steady state), beside
// a copy-only launch of the same traffic; both transforms are checked against an fp64 FFT on the host.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I shardmerge_amd/csrc tools/fftproto.hip -o exp_libs/fftproto
#include "fft_engine.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <complex>
#include <vector>
using namespace smhip;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using Plan = SPlan<8192, 256, false, 4, 32, 16, 16>;
constexpr int N = 8192, T = 256;

struct DevExec {                      // the slice of the product's Exec interface wg_fft uses
    template <class S> struct State { using value_type = S; S s; };
    template <class S> __device__ void init(State<S>&) {}
    template <class S, class F> __device__ __forceinline__ void each(State<S>& st, F&& f) { f((int)threadIdx.x, st.s); }
    __device__ __forceinline__ void sync() { __syncthreads(); }
};
struct St { float xr[EREG]; float xi[EREG]; };

// (a) the product's engine: natural order in (element t + 256 q in slot q), natural order out
template <int G, bool DO>
__global__ void __launch_bounds__(G * T, 4) k_engine(const cf2* __restrict__ in, cf2* __restrict__ out, const cf2* __restrict__ tw) {
    extern __shared__ float lds[];
    DevExec ex;
    DevExec::State<St> st;
    const int g = threadIdx.x / T, t = threadIdx.x % T;
    const size_t base = ((size_t)blockIdx.x * G + g) * N;
#pragma unroll
    for (int q = 0; q < 32; ++q) { const cf2 v = in[base + t + q * T]; st.s.xr[q] = v.x; st.s.xi[q] = v.y; }
    FftPlanDev pl;
    pl.N = N; pl.T = T; pl.npass = 3; pl.radix[0] = 32; pl.radix[1] = 16; pl.radix[2] = 16; pl.tw = tw; pl.lds_floats = Plan::lds_floats; pl.cx = 0;
    if constexpr (DO) {
        wg_fft<Plan, 1>(ex, st, pl, lds,
            [&](int tid, St& s, auto comp_c) {
                constexpr int comp = decltype(comp_c)::value;
                float* l = lds + (tid / T) * Plan::lds_floats;
                const float* x = comp_of<comp>(s);
#pragma unroll
                for (int q = 0; q < 32; ++q) l[lpad(tid % T + q * T)] = x[q];
            },
            [&](int tid, St& s, auto comp_c) {
                constexpr int comp = decltype(comp_c)::value;
                const float* l = lds + (tid / T) * Plan::lds_floats;
                float* o = comp_of<comp>(s);
#pragma unroll
                for (int q = 0; q < 32; ++q) o[q] = l[lpad(tid % T + q * T)];
            });
    }
    int t_ = t;
    SM_OPAQUE(t_);                                // (the store addresses are computed here, not carried across the transform)
#pragma unroll
    for (int q = 0; q < 32; ++q) { cf2 v = {st.s.xr[q], st.s.xi[q]}; out[base + t_ + q * T] = v; }
}

// (b) wave-local.  LDS per group: the work-group exchange needs 4 x (2048 + pad) floats per component; the wave-private
// transposes use the same space (32 x 66 floats per wave and component).
constexpr int WPITCH = 66;                        // floats per k1 row of a wave's transpose image: bank 2 k1 + par
constexpr int XPITCH = 2048 + 64;                 // floats per wave in the work-group exchange
constexpr int WL_LDS = 4 * XPITCH;                // floats per group (one component at a time)
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
template <int G>
__global__ void __launch_bounds__(G * T, 4) k_wavelocal(const cf2* __restrict__ in, cf2* __restrict__ out, const cf2* __restrict__ tw2048,
                                                        const cf2* __restrict__ tw8192) {
    extern __shared__ float lds[];
    const int g = threadIdx.x / T, t = threadIdx.x % T;
    const int w = t >> 6, l = t & 63;
    const int k1p = l >> 1, par = l & 1;          // after the transpose: problem k1p, half par
    float* gl = lds + g * WL_LDS;
    float* wl = gl + w * XPITCH;                  // this wave's private image (32 * 66 <= XPITCH)
    const size_t base = ((size_t)blockIdx.x * G + g) * N;
    float xr[32], xi[32];
    // natural-order input x[n]: lane l of wave w takes n = 4 (l + 64 j) + w
#pragma unroll
    for (int j = 0; j < 32; ++j) { const cf2 v = in[base + 4 * (l + 64 * j) + w]; xr[j] = v.x; xi[j] = v.y; }
    {
        // step 1: DFT over j, twiddle W_2048^(l k1)
        Dft<32>::run(xr, xi);
        apply_twiddles<32>(xr, xi, tw2048, l, 0);
        // step 2: wave-private transpose, one component at a time
        float yr[32], yi[32];
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) wl[k1 * WPITCH + l] = xr[k1];
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this wave's stores have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 32; ++i) yr[i] = wl[k1p * WPITCH + par + 2 * i];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) wl[k1 * WPITCH + l] = xi[k1];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 32; ++i) yi[i] = wl[k1p * WPITCH + par + 2 * i];
        // step 3: DFT over i, W_64^(par k2'), lane-pair butterfly
        Dft<32>::run(yr, yi);
        static_for<1, 32>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            // W_64^k = W_32^(k/2) for even k; odd k: the half-way root
            const float c = (float)__builtin_cos(-2.0 * 3.14159265358979323846 * k / 64.0);
            const float s = (float)__builtin_sin(-2.0 * 3.14159265358979323846 * k / 64.0);
            const float cr = par ? c : 1.f, si = par ? s : 0.f;
            const float tr = yr[k] * cr - yi[k] * si;
            yi[k] = yr[k] * si + yi[k] * cr;
            yr[k] = tr;
        });
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float pr = dpp_xor1(yr[k]), pi = dpp_xor1(yi[k]);
            // par 0 keeps D0 + D1 (k2 = k), par 1 keeps D0 - D1 (k2 = k + 32): own value is D_par, partner's D_(1-par)
            yr[k] = par ? pr - yr[k] : yr[k] + pr;
            yi[k] = par ? pi - yi[k] : yi[k] + pi;
        }
        // lane (k1p, par), slot k2' holds F_w[k1p + 32 k2' + 1024 par]
        // step 4: the work-group exchange, one component at a time; reader t takes k = t + 256 r
        __syncthreads();                          // (the transposes of every wave are done with the space)
        float zr[32], zi[32];
        float* mine = gl + w * XPITCH + par * 32; // the par = 1 half sits 32 words further: different banks than its partner
#pragma unroll
        for (int k = 0; k < 32; ++k) mine[k1p + 32 * k + 1024 * par] = yr[k];
        __syncthreads();
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
            const int k = t + 256 * r8;
            const int off = k + ((k >= 1024) ? 32 : 0);
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) zr[r8 * 4 + ww] = gl[ww * XPITCH + off];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; ++k) mine[k1p + 32 * k + 1024 * par] = yi[k];
        __syncthreads();
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
            const int k = t + 256 * r8;
            const int off = k + ((k >= 1024) ? 32 : 0);
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) zi[r8 * 4 + ww] = gl[ww * XPITCH + off];
        }
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
            const int k = t + 256 * r8;
            const cf2 w1 = tw8192[k], w2 = tw8192[2 * k], w3 = tw8192[3 * k];
            float br[4] = {zr[r8 * 4], zr[r8 * 4 + 1], zr[r8 * 4 + 2], zr[r8 * 4 + 3]};
            float bi[4] = {zi[r8 * 4], zi[r8 * 4 + 1], zi[r8 * 4 + 2], zi[r8 * 4 + 3]};
            cmul(br[1], bi[1], w1.x, w1.y); cmul(br[2], bi[2], w2.x, w2.y); cmul(br[3], bi[3], w3.x, w3.y);
            Dft<4>::run(br, bi);
#pragma unroll
            for (int q = 0; q < 4; ++q) { xr[r8 * 4 + q] = br[q]; xi[r8 * 4 + q] = bi[q]; }
        }
    }
    // slot r8 * 4 + q of thread t holds X[t + 256 r8 + 2048 q]
    int t_ = t;
    SM_OPAQUE(t_);
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
        for (int q = 0; q < 4; ++q) { cf2 v = {xr[r8 * 4 + q], xi[r8 * 4 + q]}; out[base + t_ + 256 * r8 + 2048 * q] = v; }
}

static void host_fft(std::vector<std::complex<double>>& a) {
    const size_t n = a.size();
    if (n == 1) return;
    std::vector<std::complex<double>> e(n / 2), o(n / 2);
    for (size_t i = 0; i < n / 2; ++i) { e[i] = a[2 * i]; o[i] = a[2 * i + 1]; }
    host_fft(e); host_fft(o);
    for (size_t k = 0; k < n / 2; ++k) {
        const std::complex<double> w = std::polar(1.0, -2.0 * M_PI * (double)k / (double)n) * o[k];
        a[k] = e[k] + w; a[k + n / 2] = e[k] - w;
    }
}

template <class Launch>
static float time_it(Launch&& go, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    go(); go();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main() {
    constexpr int G = 2;
    const int ngroups = 256 * 2 * G * 16;         // 16 rounds of 2 work-groups per CU
    std::vector<cf2> h((size_t)ngroups * N);
    srand(1);
    for (auto& v : h) { v.x = (float)rand() / RAND_MAX - 0.5f; v.y = (float)rand() / RAND_MAX - 0.5f; }
    std::vector<cf2> t2(2048), t8(8192 * 3 + 1);
    for (int j = 0; j < 2048; ++j) { t2[j].x = (float)cos(-2.0 * M_PI * j / 2048); t2[j].y = (float)sin(-2.0 * M_PI * j / 2048); }
    for (size_t j = 0; j < t8.size(); ++j) { t8[j].x = (float)cos(-2.0 * M_PI * (double)j / 8192); t8[j].y = (float)sin(-2.0 * M_PI * (double)j / 8192); }
    cf2 *din, *dout, *dt2, *dt8;
    CK(hipMalloc(&din, h.size() * sizeof(cf2))); CK(hipMalloc(&dout, h.size() * sizeof(cf2)));
    CK(hipMalloc(&dt2, t2.size() * sizeof(cf2))); CK(hipMalloc(&dt8, t8.size() * sizeof(cf2)));
    CK(hipMemcpy(din, h.data(), h.size() * sizeof(cf2), hipMemcpyHostToDevice));
    CK(hipMemcpy(dt2, t2.data(), t2.size() * sizeof(cf2), hipMemcpyHostToDevice));
    CK(hipMemcpy(dt8, t8.data(), t8.size() * sizeof(cf2), hipMemcpyHostToDevice));
    const size_t lds_a = (size_t)G * Plan::lds_floats * 4, lds_b = (size_t)G * WL_LDS * 4;
    const int grid = ngroups / G;

    // correctness of group 0 against fp64
    std::vector<std::complex<double>> ref(N);
    for (int i = 0; i < N; ++i) ref[i] = {h[i].x, h[i].y};
    host_fft(ref);
    double mx = 0; for (auto& v : ref) mx = fmax(mx, std::abs(v));
    std::vector<cf2> o(N);
    k_engine<G, true><<<grid, G * T, lds_a>>>(din, dout, dt8);
    CK(hipMemcpy(o.data(), dout, N * sizeof(cf2), hipMemcpyDeviceToHost));
    double ea = 0; for (int i = 0; i < N; ++i) ea = fmax(ea, std::abs(std::complex<double>(o[i].x, o[i].y) - ref[i]));
    k_wavelocal<G><<<grid, G * T, lds_b>>>(din, dout, dt2, dt8);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, N * sizeof(cf2), hipMemcpyDeviceToHost));
    double eb = 0; for (int i = 0; i < N; ++i) eb = fmax(eb, std::abs(std::complex<double>(o[i].x, o[i].y) - ref[i]));
    printf("max |X - fp64| / max |X|: engine %.2e, wave-local %.2e\n", ea / mx, eb / mx);

    // one round (two work-groups per CU: every work-group loads, transforms, stores at the same time - the transform
    // phase is fully exposed) and sixteen rounds (the product's steady state: phases of different work-groups overlap)
    for (int rounds : {1, 16}) {
        const int gr = 256 * 2 * rounds;
        const float c = time_it([&] { k_engine<G, false><<<gr, G * T, lds_a>>>(din, dout, dt8); }, 20);
        const float a = time_it([&] { k_engine<G, true><<<gr, G * T, lds_a>>>(din, dout, dt8); }, 20);
        const float b = time_it([&] { k_wavelocal<G><<<gr, G * T, lds_b>>>(din, dout, dt2, dt8); }, 20);
        printf("%2d round(s) of 2 x %d-thread work-groups per CU (%d transforms, %.0f MB in + out): copy only %7.1f us, engine %7.1f us, wave-local %7.1f us\n",
               rounds, G * T, gr * G, 2.0 * gr * G * N * 8 / 1e6, c * 1e3, a * 1e3, b * 1e3);
    }
    return (ea / mx < 1e-5 && eb / mx < 1e-5) ? 0 : 1;
}
