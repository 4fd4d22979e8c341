// membench.hip - HBM access-granularity probe for the column passes (F2 reads,
// I1 writes).  Each work-group walks ALL rows of a [R][P] float4 array and
// touches W bytes per row (W = 16..256); with "xcd" mapping the work-groups that
// share 128-byte lines are placed on one XCD (blocks b, b+8, ... share an L2).
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// V = float4s per row per work-group (W = 16*V bytes)
template <int V, bool XCD, bool WRITE>
__global__ void __launch_bounds__(512) strided(float4* a, int R, int P, int nblk, float* sink) {
    int bid = blockIdx.x;
    if (XCD) {           // the 128/(16V) work-groups sharing a line get equal bid % 8
        constexpr int G = (8 / V) > 0 ? (8 / V) : 1;     // work-groups per 128-byte line
        int per = 8 * G;
        int y = bid / per, i = (bid % per) / 8, x = bid % 8;
        bid = (y * 8 + x) * G + i;
        if (bid >= nblk) return;
    }
    const int col = bid * V;
    float acc = 0.f;
    for (int r = threadIdx.x; r < R; r += blockDim.x * 4) {
        float4 v[4][V];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int rr = r + q * blockDim.x;
            if (rr < R) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (WRITE) { float4 w = {(float)rr, (float)j, 1.f, 2.f}; a[(size_t)rr * P + col + j] = w; }
                    else v[q][j] = a[(size_t)rr * P + col + j];
                }
            }
        }
        if (!WRITE) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < V; ++j) if (r + q * (int)blockDim.x < R) acc += v[q][j].x + v[q][j].w;
        }
    }
    if (!WRITE && acc == 12345.678f) sink[0] = acc;
}

// same bytes as strided<V>, but the V float4s of a row piece go to V adjacent LANES (one load
// instruction touches 64/V lines) instead of V loads of one lane (each touching 64 lines)
template <int V, bool XCD>
__global__ void __launch_bounds__(512) strided_lanes(float4* a, int R, int P, int nblk, float* sink) {
    int bid = blockIdx.x;
    if (XCD) {
        constexpr int G = (8 / V) > 0 ? (8 / V) : 1;
        int per = 8 * G;
        int y = bid / per, i = (bid % per) / 8, x = bid % 8;
        bid = (y * 8 + x) * G + i;
        if (bid >= nblk) return;
    }
    const int col = bid * V;
    float acc = 0.f;
    const int piece = threadIdx.x % V, r0 = threadIdx.x / V, rstep = blockDim.x / V;
    for (int r = r0; r < R; r += rstep * 16) {
        float4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int rr = r + q * rstep;
            v[q] = a[(size_t)(rr < R ? rr : 0) * P + col + piece];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q].x + v[q].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int V, bool XCD>
int run_lanes(float4* a, int R, int P, int cols, float* sink, const char* label) {
    const int nblk = cols / V;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = ((nblk + 63) / 64) * 64;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((strided_lanes<V, XCD>), dim3(grid), dim3(512), 0, 0, a, R, P, nblk, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((strided_lanes<V, XCD>), dim3(grid), dim3(512), 0, 0, a, R, P, nblk, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bytes = (double)R * cols * 16.0 * reps;
    printf("%-28s W=%4d B  %8.1f GB/s\n", label, V * 16, bytes / (ms * 1e-3) / 1e9);
    return 0;
}

__global__ void copy16(const float4* a, float4* b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <int V, bool XCD, bool WRITE>
int run(float4* a, int R, int P, int cols, float* sink, const char* label) {
    const int nblk = cols / V;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = ((nblk + 63) / 64) * 64;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((strided<V, XCD, WRITE>), dim3(grid), dim3(512), 0, 0, a, R, P, nblk, sink);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((strided<V, XCD, WRITE>), dim3(grid), dim3(512), 0, 0, a, R, P, nblk, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bytes = (double)R * cols * 16.0 * reps;
    printf("%-28s W=%4d B  %8.1f GB/s\n", label, V * 16, bytes / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    const int R = 8192, cols = 4096, P = 4104;
    float4 *a, *b; float* sink;
    CK(hipMalloc(&a, (size_t)R * P * 16)); CK(hipMalloc(&b, (size_t)R * P * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0, (size_t)R * P * 16));
    {   // baseline: coalesced copy
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        size_t n = (size_t)R * P;
        hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("coalesced copy (r+w)                 %8.1f GB/s\n", 2.0 * n * 16 * 5 / (ms * 1e-3) / 1e9);
    }
    run<1, false, false>(a, R, P, cols, sink, "read  strided, plain map");
    run<1, true, false>(a, R, P, cols, sink, "read  strided, xcd map");
    run<2, false, false>(a, R, P, cols, sink, "read  strided, plain map");
    run<2, true, false>(a, R, P, cols, sink, "read  strided, xcd map");
    run<4, false, false>(a, R, P, cols, sink, "read  strided, plain map");
    run<4, true, false>(a, R, P, cols, sink, "read  strided, xcd map");
    run<8, false, false>(a, R, P, cols, sink, "read  strided, plain map");
    run<16, false, false>(a, R, P, cols, sink, "read  strided, plain map");
    run_lanes<1, true>(a, R, P, cols, sink, "read lanes/row, xcd (16 deep)");
    run_lanes<2, true>(a, R, P, cols, sink, "read lanes/row, xcd (16 deep)");
    run_lanes<4, true>(a, R, P, cols, sink, "read lanes/row, xcd (16 deep)");
    run_lanes<8, false>(a, R, P, cols, sink, "read lanes/row, plain");
    run<1, false, true>(a, R, P, cols, sink, "write strided, plain map");
    run<1, true, true>(a, R, P, cols, sink, "write strided, xcd map");
    run<2, false, true>(a, R, P, cols, sink, "write strided, plain map");
    run<2, true, true>(a, R, P, cols, sink, "write strided, xcd map");
    run<4, false, true>(a, R, P, cols, sink, "write strided, plain map");
    run<4, true, true>(a, R, P, cols, sink, "write strided, xcd map");
    run<8, false, true>(a, R, P, cols, sink, "write strided, plain map");
    run<16, false, true>(a, R, P, cols, sink, "write strided, plain map");
    return 0;
}
