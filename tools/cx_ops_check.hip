// Device check of the complex-packed butterflies (fft_engine.hpp, PACK mode 2) against the single-float ones:
// every radix the planner uses, on random data, plus the twiddle application.  Prints the largest difference
// relative to the largest output (a few 1e-7: the two forms round differently) and fails above 2e-6.
//   hipcc -O3 --offload-arch=gfx950 -I shardmerge_amd/csrc tools/cx_ops_check.hip -o gpurun_out/cx_ops_check
#include "fft_engine.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace smhip;

template <int R>
__global__ void k_check(const float* in, const cf2* tw, float* out_s, float* out_c, int kidx) {
    const int t = threadIdx.x;
    float re[R], im[R];
    vf2 z[R];
    for (int i = 0; i < R; ++i) { re[i] = in[(t * R + i) * 2]; im[i] = in[(t * R + i) * 2 + 1]; z[i] = mk2(re[i], im[i]); }
    if constexpr (R <= 32 && R != 28) {
        apply_twiddles<R>(re, im, tw, kidx + t, 0);
        apply_twiddles_cx<R>(z, tw, kidx + t);
    }
    Dft<R>::run(re, im);
    CDft<R>::run(z);
    for (int i = 0; i < R; ++i) {
        out_s[(t * R + i) * 2] = re[i]; out_s[(t * R + i) * 2 + 1] = im[i];
        out_c[(t * R + i) * 2] = z[i].x; out_c[(t * R + i) * 2 + 1] = z[i].y;
    }
}

template <int R>
static int run(const cf2* dtw, const std::vector<cf2>& htw, int NTW) {
    const int T = 64;
    std::vector<float> h(T * R * 2);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *din, *ds, *dc;
    hipMalloc(&din, h.size() * 4); hipMalloc(&ds, h.size() * 4); hipMalloc(&dc, h.size() * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k_check<R><<<1, T>>>(din, dtw, ds, dc, 3);
    std::vector<float> a(h.size()), b(h.size());
    hipMemcpy(a.data(), ds, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), dc, h.size() * 4, hipMemcpyDeviceToHost);
    // fp64 reference of the same thing
    double worst_s = 0, worst_c = 0, mx = 0;
    for (int t = 0; t < T; ++t) {
        for (int k = 0; k < R; ++k) {
            double sr = 0, si = 0;
            for (int i = 0; i < R; ++i) {
                double xr = h[(t * R + i) * 2], xi = h[(t * R + i) * 2 + 1];
                if (R <= 32 && R != 28) {
                    const double ang = -2.0 * M_PI * (double)(((3 + t) * i) % NTW) / NTW;
                    const double wr = cos(ang), wi = sin(ang);
                    const double r2 = xr * wr - xi * wi; xi = xr * wi + xi * wr; xr = r2;
                }
                const double ang = -2.0 * M_PI * (double)((i * k) % R) / R;
                sr += xr * cos(ang) - xi * sin(ang); si += xr * sin(ang) + xi * cos(ang);
            }
            mx = fmax(mx, fmax(fabs(sr), fabs(si)));
            worst_s = fmax(worst_s, fmax(fabs(a[(t * R + k) * 2] - sr), fabs(a[(t * R + k) * 2 + 1] - si)));
            worst_c = fmax(worst_c, fmax(fabs(b[(t * R + k) * 2] - sr), fabs(b[(t * R + k) * 2 + 1] - si)));
        }
    }
    printf("radix %2d: single floats %.2e, complex-packed %.2e (of max |X| = %.2f)\n", R, worst_s / mx, worst_c / mx, mx);
    hipFree(din); hipFree(ds); hipFree(dc);
    return worst_c / mx < 2e-6 ? 0 : 1;
}

int main() {
    const int NTW = 4096;
    std::vector<cf2> htw(NTW);
    for (int j = 0; j < NTW; ++j) { htw[j].x = (float)cos(-2.0 * M_PI * j / NTW); htw[j].y = (float)sin(-2.0 * M_PI * j / NTW); }
    cf2* dtw; hipMalloc(&dtw, NTW * sizeof(cf2));
    hipMemcpy(dtw, htw.data(), NTW * sizeof(cf2), hipMemcpyHostToDevice);
    int bad = 0;
    bad += run<2>(dtw, htw, NTW); bad += run<3>(dtw, htw, NTW); bad += run<4>(dtw, htw, NTW); bad += run<5>(dtw, htw, NTW);
    bad += run<7>(dtw, htw, NTW); bad += run<8>(dtw, htw, NTW); bad += run<11>(dtw, htw, NTW); bad += run<13>(dtw, htw, NTW);
    bad += run<16>(dtw, htw, NTW); bad += run<28>(dtw, htw, NTW); bad += run<32>(dtw, htw, NTW);
    printf(bad ? "FAILED\n" : "ok\n");
    return bad;
}
