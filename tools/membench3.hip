// membench3.hip - what a plane-streaming pass (selection, blend, rescale ...) can read: two fp32 planes of
// 4097 x 28672 elements (470 MB each), read once, reduced to a per-work-group partial (round 4).
//   MODE 0 "chunk":  work-group b owns ONE contiguous chunk (the product's traversal: start = b * chunks * 256 quads)
//   MODE 1 "sweep":  the grid sweeps the planes together: step c of work-group b reads block c * grid + b
//   U = 16-byte loads per plane in flight per thread; WPC = resident work-groups per CU (grid = 256 * WPC)
//   WRITE = 1: also writes one plane (the blend's traffic: 2 reads + 1 write)
//   hipcc --offload-arch=gfx950 -O3 tools/membench3.hip -o exp_libs/membench3 && exp_libs/membench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int U, int WRITE>
__global__ void __launch_bounds__(256) stream2(const float4* __restrict__ X, const float4* __restrict__ Y, float4* __restrict__ Z,
                                               float* __restrict__ out, size_t nquad, int chunks) {
    const int tid = threadIdx.x, nt = 256;
    float acc = 0.f;
    for (int c0 = 0; c0 < chunks; c0 += U) {
        float4 a[U], b[U];
        size_t q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u;
            q[u] = MODE == 0 ? ((size_t)blockIdx.x * chunks + c) * nt + tid : ((size_t)c * gridDim.x + blockIdx.x) * nt + tid;
            a[u] = X[q[u] < nquad ? q[u] : nquad - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) b[u] = Y[q[u] < nquad ? q[u] : nquad - 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c0 + u < chunks && q[u] < nquad) {
                acc += a[u].x * b[u].x + a[u].y * b[u].y + a[u].z * b[u].z + a[u].w * b[u].w;
                if (WRITE) { float4 r = {a[u].x + b[u].x, a[u].y + b[u].y, a[u].z + b[u].z, a[u].w + b[u].w}; Z[q[u]] = r; }
            }
        }
    }
    // (one value per thread: the reduction is not what is measured)
    if (acc == 12345.678f) out[blockIdx.x * nt + tid] = acc;
}

template <int MODE, int U, int WRITE>
static void run(const float4* X, const float4* Y, float4* Z, float* out, size_t nquad, int wpc) {
    const int grid = 256 * wpc;
    int chunks = (int)((nquad + (size_t)grid * 256 - 1) / ((size_t)grid * 256));
    chunks = (chunks + U - 1) / U * U;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) stream2<MODE, U, WRITE><<<grid, 256>>>(X, Y, Z, out, nquad, chunks);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) stream2<MODE, U, WRITE><<<grid, 256>>>(X, Y, Z, out, nquad, chunks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)nquad * 16 * (2 + WRITE);
    printf("%s U=%d WGs/CU=%2d %s: %7.1f us  %6.0f GB/s\n", MODE ? "sweep" : "chunk", U, wpc, WRITE ? "2 reads + 1 write" : "2 reads          ", ms * 1e3, bytes / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t n = (size_t)4097 * 28672, nquad = n / 4;
    float4 *X, *Y, *Z; float* out;
    CK(hipMalloc(&X, n * 4)); CK(hipMalloc(&Y, n * 4)); CK(hipMalloc(&Z, n * 4)); CK(hipMalloc(&out, 256 * 16 * 256 * 4));
    CK(hipMemset(X, 0, n * 4)); CK(hipMemset(Y, 0, n * 4)); CK(hipMemset(Z, 0, n * 4));
    for (int wpc : {3, 5, 8}) {
        run<0, 4, 0>(X, Y, Z, out, nquad, wpc);
        run<1, 4, 0>(X, Y, Z, out, nquad, wpc);
        run<0, 8, 0>(X, Y, Z, out, nquad, wpc);
        run<1, 8, 0>(X, Y, Z, out, nquad, wpc);
        run<0, 2, 0>(X, Y, Z, out, nquad, wpc);
        run<1, 2, 0>(X, Y, Z, out, nquad, wpc);
    }
    for (int wpc : {5, 8}) {
        run<0, 4, 1>(X, Y, Z, out, nquad, wpc);
        run<1, 4, 1>(X, Y, Z, out, nquad, wpc);
    }
    return 0;
}
