// membench2.hip - access-pattern probe for candidate T1 layouts between the row pass and the
// single-signal column pass (round 4).  T1 holds M row pairs x P float4 (one float4 = one bin of
// rows 2m, 2m+1); layout ILV: element (m, k) at ((m / ILV) * P + k) * ILV + m % ILV, i.e. ILV
// row pairs of a bin are adjacent (ILV * 16-byte pieces for a column reader).
//   colread<ILV, G>: the column pass without its transform: a work-group of G*256 threads takes
//       G adjacent bin columns, every thread 16 float4 (all in flight), then writes the two
//       planes contiguously (2 x 32 KB per column).
//   rowwrite<ILV, SEQ>: the row pass without its transform: a 256-thread work-group reads bf16
//       rows and writes row pairs; SEQ = 1: ONE work-group writes the ILV pairs of a group one
//       after the other (16 bytes at ILV*16 stride per sweep, `delay` ns of sleep between sweeps
//       standing for the transform), SEQ = 0: ILV co-scheduled work-groups (xcd-adjacent) each
//       write one pair of the group.
//   hipcc --offload-arch=gfx950 -O3 tools/membench2.hip -o tools/membench2 && tools/membench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int G) {
    const int per = 8 * G;
    const int y = bid / per, i = (bid % per) / 8, x = bid % 8;
    return (y * 8 + x) * G + i;
}

template <int ILV, int G>
__global__ void __launch_bounds__(G * 256) colread(const float4* __restrict__ t1, float* __restrict__ re, float* __restrict__ im, int M, int P, int ncols) {
    constexpr int XG = (G * ILV >= 8) ? 1 : 8 / (G * ILV);     // work-groups sharing a 128-byte line
    const int lbid = xcd_remap(blockIdx.x, XG);
    const int kbase = lbid * G;
    if (kbase >= ncols) return;
    const int b = threadIdx.x % G, lane = threadIdx.x / G;      // lane < 256
    float4 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int m = ILV * (lane + (q / ILV) * 256) + q % ILV;           // row pair, < 4096 = M
        v[q] = t1[((size_t)(m / ILV) * P + kbase + b) * ILV + m % ILV];
    }
    // "transform": nothing; then the contiguous plane stores of column kbase + g, g = tid / 256
    const int g = threadIdx.x / 256, t = threadIdx.x % 256;
    float* dre = re + (size_t)(kbase + g) * (2 * M);
    float* dim = im + (size_t)(kbase + g) * (2 * M);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k0 = 4 * (t + u * 256);
        float4 a = {v[2 * u].x, v[2 * u].z, v[2 * u + 1].x, v[2 * u + 1].z};
        float4 c = {v[2 * u].y, v[2 * u].w, v[2 * u + 1].y, v[2 * u + 1].w};
        *(float4*)(dre + k0) = a;
        *(float4*)(dim + k0) = c;
    }
}

// colread with the plane stores of a LAST radix pass written directly (no final exchange): thread t holds outputs
// t + 256 u, u < 32, and stores them as dwords - a wave-instruction writes 256 contiguous bytes (two lines) where the
// exchanged form writes 1 KB (eight lines) with dwordx4
template <int G>
__global__ void __launch_bounds__(G * 256) colread_dwstore(const float4* __restrict__ t1, float* __restrict__ re, float* __restrict__ im, int M, int P, int ncols) {
    constexpr int XG = (G >= 8) ? 1 : 8 / G;
    const int lbid = xcd_remap(blockIdx.x, XG);
    const int kbase = lbid * G;
    if (kbase >= ncols) return;
    const int b = threadIdx.x % G, lane = threadIdx.x / G;
    float4 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = t1[(size_t)(lane + q * 256) * P + kbase + b];
    const int g = threadIdx.x / 256, t = threadIdx.x % 256;
    float* dre = re + (size_t)(kbase + g) * (2 * M);
    float* dim = im + (size_t)(kbase + g) * (2 * M);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        dre[t + 256 * (2 * u)] = v[u].x; dre[t + 256 * (2 * u + 1)] = v[u].z;
        dim[t + 256 * (2 * u)] = v[u].y; dim[t + 256 * (2 * u + 1)] = v[u].w;
    }
}
template <int G>
void run_colread_dw(const float4* t1, float* re, float* im, int M, int P, int ncols) {
    constexpr int XG = (G >= 8) ? 1 : 8 / G;
    const int nwg = ncols / G;
    const int grid = ((nwg + 8 * XG - 1) / (8 * XG)) * (8 * XG);
    const float ms = time_it([&] { hipLaunchKernelGGL((colread_dwstore<G>), dim3(grid), dim3(G * 256), 0, 0, t1, re, im, M, P, ncols); });
    printf("colread  ILV=1 G=%d dword plane stores (no final exchange): %7.1f us  read+write %6.0f GB/s\n", G, ms * 1e3, 2.0 * M * ncols * 16 / (ms * 1e-3) / 1e9);
}

template <int ILV, int SEQ>
__global__ void __launch_bounds__(256) rowwrite(const uint4* __restrict__ x, float4* __restrict__ t1, int M, int P, int Cb, int delay_ns) {
    // M row pairs; input: 2 rows of 8192 bf16 per pair = 32 KB = 2048 uint4 -> 8 per thread
    const int ngroups = M / ILV;
    int grp, j0, j1;
    if (SEQ) { grp = blockIdx.x; j0 = 0; j1 = ILV; if (grp >= ngroups) return; }
    else {
        const int l = xcd_remap(blockIdx.x, ILV);
        grp = l / ILV; j0 = l % ILV; j1 = j0 + 1;
        if (grp >= ngroups) return;
    }
    for (int j = j0; j < j1; ++j) {
        const int m = grp * ILV + j;
        uint4 r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = x[(size_t)m * 2048 + threadIdx.x + q * 256];
        if (delay_ns > 0) {
            const unsigned long long t0 = wall_clock64();            // 100 MHz
            while (wall_clock64() - t0 < (unsigned long long)delay_ns / 10) __builtin_amdgcn_s_sleep(8);
        }
        float4* dst = t1 + (size_t)grp * P * ILV + j;
#pragma unroll
        for (int u = 0; u < 17; ++u) {
            const int k = threadIdx.x + u * 256;
            if (k < Cb) {
                const uint4 w = r[u % 8];
                float4 v = {__uint_as_float(w.x << 16), __uint_as_float(w.y << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w)};
                dst[(size_t)k * ILV] = v;
            }
        }
    }
}

template <class F> float time_it(F launch, int reps = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

template <int ILV, int G>
void run_colread(const float4* t1, float* re, float* im, int M, int P, int ncols) {
    constexpr int XG = (G * ILV >= 8) ? 1 : 8 / (G * ILV);
    const int nwg = ncols / G;
    const int grid = ((nwg + 8 * XG - 1) / (8 * XG)) * (8 * XG);
    const float ms = time_it([&] { hipLaunchKernelGGL((colread<ILV, G>), dim3(grid), dim3(G * 256), 0, 0, t1, re, im, M, P, ncols); });
    const double rd = (double)M * ncols * 16, wr = (double)M * ncols * 16;
    printf("colread  ILV=%d G=%d (pieces %3d B, %4d thr): %7.1f us  read+write %6.0f GB/s\n", ILV, G, ILV * G * 16, G * 256, ms * 1e3, (rd + wr) / (ms * 1e-3) / 1e9);
}
template <int ILV, int SEQ>
void run_rowwrite(const uint4* x, float4* t1, int M, int P, int Cb, int delay_ns) {
    const int nwg = SEQ ? M / ILV : M;
    const int grid = SEQ ? nwg : ((nwg + 8 * ILV - 1) / (8 * ILV)) * (8 * ILV);
    const float ms = time_it([&] { hipLaunchKernelGGL((rowwrite<ILV, SEQ>), dim3(grid), dim3(256), 0, 0, x, t1, M, P, Cb, delay_ns); });
    const double rd = (double)M * 32768, wr = (double)M * Cb * 16;
    printf("rowwrite ILV=%d %s delay %5d ns: %7.1f us  read+write %6.0f GB/s\n", ILV, SEQ ? "sequential (one WG)  " : "co-scheduled WGs     ", delay_ns, ms * 1e3, (rd + wr) / (ms * 1e-3) / 1e9);
}

int main() {
    const int M = 4096, P = 4104, Cb = 4097, ncols = 4096;
    float4* t1; float *re, *im; uint4* x;
    CK(hipMalloc(&t1, (size_t)M * P * 16 + 4096)); CK(hipMalloc(&re, (size_t)ncols * 2 * M * 4)); CK(hipMalloc(&im, (size_t)ncols * 2 * M * 4));
    CK(hipMalloc(&x, (size_t)M * 32768));
    CK(hipMemset(t1, 0, (size_t)M * P * 16)); CK(hipMemset(x, 1, (size_t)M * 32768));
    printf("--- column pass without transform: 268 MB strided read + 268 MB contiguous write\n");
    run_colread<1, 1>(t1, re, im, M, P, ncols);
    run_colread<1, 2>(t1, re, im, M, P, ncols);
    run_colread<1, 4>(t1, re, im, M, P, ncols);
    run_colread_dw<1>(t1, re, im, M, P, ncols);
    run_colread_dw<2>(t1, re, im, M, P, ncols);
    run_colread<2, 1>(t1, re, im, M, P, ncols);
    run_colread<2, 2>(t1, re, im, M, P, ncols);
    run_colread<4, 1>(t1, re, im, M, P, ncols);
    run_colread<4, 2>(t1, re, im, M, P, ncols);
    run_colread<8, 1>(t1, re, im, M, P, ncols);
    run_colread<8, 2>(t1, re, im, M, P, ncols);
    run_colread<16, 1>(t1, re, im, M, P, ncols);
    printf("--- row pass without transform: 134 MB contiguous read + 268 MB write\n");
    for (int d : {0, 4000}) {
        run_rowwrite<1, 1>(x, t1, M, P, Cb, d);
        run_rowwrite<2, 1>(x, t1, M, P, Cb, d);
        run_rowwrite<4, 1>(x, t1, M, P, Cb, d);
        run_rowwrite<8, 1>(x, t1, M, P, Cb, d);
        run_rowwrite<16, 1>(x, t1, M, P, Cb, d);
        run_rowwrite<2, 0>(x, t1, M, P, Cb, d);
        run_rowwrite<4, 0>(x, t1, M, P, Cb, d);
        run_rowwrite<8, 0>(x, t1, M, P, Cb, d);
    }
    return 0;
}
