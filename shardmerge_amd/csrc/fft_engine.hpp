// fft_engine.hpp - work-group FFT engine for gfx950 (MI355X).
//
// One FFT of length N is computed by a group of T threads (T a multiple of the
// 64-lane wavefront).  The data lives in REGISTERS (<= EMAX complex values per
// thread, i.e. the 512 KB register file of a CU is the working store, not the
// 160 KB LDS); between Stockham radix passes the values are exchanged through
// LDS one component (re, then im) at a time, so an N-point transform needs only
// 4*N bytes of LDS (+1/32 padding against bank conflicts).
//
// Everything here is written once and compiled twice: by hipcc for the GPU and
// by g++ for the CPU work-group emulator used by the "not gpu" tests
// (tests/emul).  No torch, no rocFFT.
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SM_HD __host__ __device__ __forceinline__
#else
#define SM_HD inline
#endif
#define SM_CONST static constexpr
// make an integer opaque to the optimizer at this point: values derived from it cannot be hoisted
// above it, i.e. they are recomputed where they are used instead of living (or being spilled)
// across the whole kernel
#if defined(__HIP_DEVICE_COMPILE__)
#define SM_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define SM_OPAQUE(x) ((void)0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SM_LAUNDER(x) asm("" : "+v"(x))
#else
#define SM_LAUNDER(x) ((void)0)
#endif
// keep the instruction scheduler from interleaving independent butterflies: one
// butterfly's temporaries die before the next one's are born (register pressure)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SM_NO_SCHED_BARRIER)
#define SM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SM_SCHED_FENCE() ((void)0)
#endif

namespace smhip {

#include "twiddle_consts.inc"

constexpr int EMAX = 32;        // complex values a thread may hold in a pass
constexpr int EREG = EMAX + 4;  // register array length (storers need a little slack)
constexpr int MAX_PASSES = 8;

struct cf2 { float x, y; };

// Two floats side by side.  The butterflies below are templates over the value type:
// with V = vf2 one instruction stream does two isomorphic butterflies of a thread at once
// (lane = butterfly), which the gfx950 packed-f32 ALU ops (v_pk_add/mul/fma_f32) execute
// at the price of one - with no lane shuffles, because the two lanes never mix.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float vf2 __attribute__((ext_vector_type(2)));
SM_HD vf2 mk2(float a, float b) { vf2 v; v.x = a; v.y = b; return v; }
#else
struct vf2 { float x, y; };
SM_HD vf2 mk2(float a, float b) { vf2 v; v.x = a; v.y = b; return v; }
SM_HD vf2 operator+(vf2 a, vf2 b) { return mk2(a.x + b.x, a.y + b.y); }
SM_HD vf2 operator-(vf2 a, vf2 b) { return mk2(a.x - b.x, a.y - b.y); }
SM_HD vf2 operator*(vf2 a, vf2 b) { return mk2(a.x * b.x, a.y * b.y); }
SM_HD vf2 operator*(vf2 a, float b) { return mk2(a.x * b, a.y * b); }
SM_HD vf2 operator*(float a, vf2 b) { return mk2(a * b.x, a * b.y); }
SM_HD vf2 operator-(vf2 a) { return mk2(-a.x, -a.y); }
SM_HD vf2& operator+=(vf2& a, vf2 b) { a = a + b; return a; }
#endif

// plan of one N-point transform, passed to kernels by value
struct FftPlanDev {
    int N;                 // transform length
    int T;                 // threads per transform (multiple of 64)
    int npass;
    int radix[MAX_PASSES];
    const cf2* tw;         // tw[j] = exp(-2*pi*i*j/N), j < N (device memory)
    int lds_floats;        // padded LDS floats per transform (for the plan's exchange mode)
    int cx;                // 1: complex (float2) exchanges, 0: one component at a time
};

template <int I, int N, class F>
SM_HD void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <class V> SM_HD V vzero();
template <> SM_HD float vzero<float>() { return 0.f; }
template <> SM_HD vf2 vzero<vf2>() { return mk2(0.f, 0.f); }

// LDS padding: one spare word per 32 breaks the power-of-two strides of the
// radix scatters (stride-16 words would otherwise be a 16-way bank conflict).
SM_HD int lpad(int o) { return o + (o >> 5); }

// ---- multiply by W32^E (compile-time exponent) --------------------------------
template <int E, class V>
SM_HD void mul_w32(V& r, V& i) {
    constexpr int e = ((E % 32) + 32) % 32;
    if constexpr (e == 0) {
    } else if constexpr (e == 8) {          // * (-i)
        V t = r; r = i; i = -t;
    } else if constexpr (e == 16) {
        r = -r; i = -i;
    } else if constexpr (e == 24) {         // * (+i)
        V t = r; r = -i; i = t;
    } else {
        constexpr float c = W32_RE[e];
        constexpr float s = W32_IM[e];
        V t = r * c - i * s;
        i = r * s + i * c;
        r = t;
    }
}

// ---- small in-register DFTs (forward, natural order in and out) ----------------
template <int R> struct Dft;

template <> struct Dft<1> { template <class V> static SM_HD void run(V*, V*) {} };

template <> struct Dft<2> {
    template <class V> static SM_HD void run(V* re, V* im) {
        V ar = re[0], ai = im[0];
        re[0] = ar + re[1]; im[0] = ai + im[1];
        re[1] = ar - re[1]; im[1] = ai - im[1];
    }
};

template <> struct Dft<4> {
    template <class V> static SM_HD void run(V* re, V* im) {
        V t0r = re[0] + re[2], t0i = im[0] + im[2];
        V t1r = re[0] - re[2], t1i = im[0] - im[2];
        V t2r = re[1] + re[3], t2i = im[1] + im[3];
        V dr = re[1] - re[3], di = im[1] - im[3];
        V t3r = di, t3i = -dr;                           // (x1-x3) * (-i)
        re[0] = t0r + t2r; im[0] = t0i + t2i;
        re[2] = t0r - t2r; im[2] = t0i - t2i;
        re[1] = t1r + t3r; im[1] = t1i + t3i;
        re[3] = t1r - t3r; im[3] = t1i - t3i;
    }
};

// N = A*B Cooley-Tukey in registers: A-point DFTs over the slow index, twiddle,
// B-point DFTs, outputs written to natural positions k1 + A*k2.
template <int A, int B>
struct DftComposite {
    template <class V> static SM_HD void run(V* re, V* im) {
        constexpr int N = A * B;
        V yr[N], yi[N];
        static_for<0, B>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            V tr[A], ti[A];
            static_for<0, A>([&](auto a_) {
                constexpr int a = decltype(a_)::value;
                tr[a] = re[B * a + b]; ti[a] = im[B * a + b];
            });
            Dft<A>::run(tr, ti);
            static_for<0, A>([&](auto k_) {
                constexpr int k1 = decltype(k_)::value;
                V r = tr[k1], i = ti[k1];
                mul_w32<(b * k1) * (32 / N)>(r, i);
                yr[k1 * B + b] = r; yi[k1 * B + b] = i;
            });
        });
        static_for<0, A>([&](auto k_) {
            constexpr int k1 = decltype(k_)::value;
            V tr[B], ti[B];
            static_for<0, B>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                tr[b] = yr[k1 * B + b]; ti[b] = yi[k1 * B + b];
            });
            Dft<B>::run(tr, ti);
            static_for<0, B>([&](auto k2_) {
                constexpr int k2 = decltype(k2_)::value;
                re[k1 + A * k2] = tr[k2]; im[k1 + A * k2] = ti[k2];
            });
        });
    }
};

template <> struct Dft<8> { template <class V> static SM_HD void run(V* re, V* im) { DftComposite<2, 4>::run(re, im); } };
template <> struct Dft<16> { template <class V> static SM_HD void run(V* re, V* im) { DftComposite<4, 4>::run(re, im); } };
// 32 = 2 x 16 decimation in time with the even- and the odd-indexed 16-point transforms
// side by side in the two lanes; only the last radix-2 stage (and the odd half's W32^k)
// is done on single floats.  (A thread holds ONE radix-32 butterfly, so there is no
// second butterfly to pair with as for the smaller radices.)
template <> struct Dft<32> {
    static SM_HD void run(float* re, float* im) {
        vf2 er[16], ei[16];
        static_for<0, 16>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            er[i] = mk2(re[2 * i], re[2 * i + 1]); ei[i] = mk2(im[2 * i], im[2 * i + 1]);
        });
        Dft<16>::run(er, ei);
        static_for<0, 16>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            float orr = er[k].y, oi = ei[k].y;
            mul_w32<k>(orr, oi);
            re[k] = er[k].x + orr; im[k] = ei[k].x + oi;
            re[k + 16] = er[k].x - orr; im[k + 16] = ei[k].x - oi;
        });
    }
};

template <int P> struct OddTab;
template <> struct OddTab<3> { SM_HD static constexpr float c(int m) { return ODD3_COS[m]; } SM_HD static constexpr float s(int m) { return ODD3_SIN[m]; } };
template <> struct OddTab<5> { SM_HD static constexpr float c(int m) { return ODD5_COS[m]; } SM_HD static constexpr float s(int m) { return ODD5_SIN[m]; } };
template <> struct OddTab<7> { SM_HD static constexpr float c(int m) { return ODD7_COS[m]; } SM_HD static constexpr float s(int m) { return ODD7_SIN[m]; } };
template <> struct OddTab<11> { SM_HD static constexpr float c(int m) { return ODD11_COS[m]; } SM_HD static constexpr float s(int m) { return ODD11_SIN[m]; } };
template <> struct OddTab<13> { SM_HD static constexpr float c(int m) { return ODD13_COS[m]; } SM_HD static constexpr float s(int m) { return ODD13_SIN[m]; } };

// odd prime P: X_k = P_k - i Q_k, X_{P-k} = P_k + i Q_k with
// P_k = x0 + sum_j cos(2 pi j k / P)(x_j + x_{P-j}),  Q_k = sum_j sin(2 pi j k / P)(x_j - x_{P-j})
template <int P>
struct DftOdd {
    template <class V> static SM_HD void run(V* re, V* im) {
        constexpr int H = (P - 1) / 2;
        V sr[H + 1], si[H + 1], dr[H + 1], di[H + 1];
        V x0r = re[0], x0i = im[0];
        V accr = x0r, acci = x0i;
        static_for<1, H + 1>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            sr[j] = re[j] + re[P - j]; si[j] = im[j] + im[P - j];
            dr[j] = re[j] - re[P - j]; di[j] = im[j] - im[P - j];
            accr = accr + sr[j]; acci = acci + si[j];
        });
        re[0] = accr; im[0] = acci;
        static_for<1, H + 1>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            V pr = x0r, pi = x0i, qr = vzero<V>(), qi = vzero<V>();
            static_for<1, H + 1>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                constexpr float c = OddTab<P>::c((j * k) % P);
                constexpr float s = OddTab<P>::s((j * k) % P);
                pr = pr + sr[j] * c; pi = pi + si[j] * c;
                qr = qr + dr[j] * s; qi = qi + di[j] * s;
            });
            re[k] = pr + qi; im[k] = pi - qr;
            re[P - k] = pr - qi; im[P - k] = pi + qr;
        });
    }
};
// multiply by W28^E (compile-time exponent)
template <int E>
SM_HD void mul_w28(float& r, float& i) {
    constexpr int e = ((E % 28) + 28) % 28;
    if constexpr (e == 0) {
    } else if constexpr (e == 7) { float t = r; r = i; i = -t;            // * (-i)
    } else if constexpr (e == 14) { r = -r; i = -i;
    } else if constexpr (e == 21) { float t = r; r = -i; i = t;           // * (+i)
    } else {
        constexpr float c = W28_RE[e];
        constexpr float s = W28_IM[e];
        float t = r * c - i * s;
        i = r * s + i * c;
        r = t;
    }
}

template <> struct Dft<3> { template <class V> static SM_HD void run(V* re, V* im) { DftOdd<3>::run(re, im); } };
template <> struct Dft<5> { template <class V> static SM_HD void run(V* re, V* im) { DftOdd<5>::run(re, im); } };
template <> struct Dft<7> { template <class V> static SM_HD void run(V* re, V* im) { DftOdd<7>::run(re, im); } };
template <> struct Dft<11> { template <class V> static SM_HD void run(V* re, V* im) { DftOdd<11>::run(re, im); } };
template <> struct Dft<13> { template <class V> static SM_HD void run(V* re, V* im) { DftOdd<13>::run(re, im); } };

// 28 = 4 x 7 (the 7 * 2^k lengths of Llama / Mixtral MLP tensors: 14336 = 32*16*28)
template <> struct Dft<28> {
    static SM_HD void run(float* re, float* im) {
        constexpr int A = 4, B = 7, N = 28;
        float yr[N], yi[N];
        static_for<0, B>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            float tr[A], ti[A];
            static_for<0, A>([&](auto a_) { constexpr int a = decltype(a_)::value; tr[a] = re[B * a + b]; ti[a] = im[B * a + b]; });
            Dft<A>::run(tr, ti);
            static_for<0, A>([&](auto k_) {
                constexpr int k1 = decltype(k_)::value;
                float r = tr[k1], i = ti[k1];
                mul_w28<b * k1>(r, i);
                yr[k1 * B + b] = r; yi[k1 * B + b] = i;
            });
        });
        static_for<0, A>([&](auto k_) {
            constexpr int k1 = decltype(k_)::value;
            float tr[B], ti[B];
            static_for<0, B>([&](auto b_) { constexpr int b = decltype(b_)::value; tr[b] = yr[k1 * B + b]; ti[b] = yi[k1 * B + b]; });
            Dft<B>::run(tr, ti);
            static_for<0, B>([&](auto k2_) { constexpr int k2 = decltype(k2_)::value; re[k1 + A * k2] = tr[k2]; im[k1 + A * k2] = ti[k2]; });
        });
    }
};

// ---- complex-packed arithmetic (PACK mode 2): one vf2 holds (re, im) of ONE value -------------
// The pair-packed form above (two butterflies side by side) doubles the twiddle registers of a pass;
// the 128-VGPR kernels (512/1024-thread work-groups) spill with it and ran on single floats.  Here the two
// lanes of a packed op are the two COMPONENTS: a complex add is one v_pk_add_f32, a complex multiply
// v_pk_mul_f32 + v_pk_fma_f32, a multiply by -i costs nothing (the consuming add swaps and negates
// through op_sel / neg_lo / neg_hi) - half the instructions of the scalar form at ITS register count.
// clang folds a pure lane swap into op_sel but not a swap with one lane negated: those three ops are
// spelled out.
SM_HD vf2 cx_mul(vf2 x, vf2 w) {                 // x * w
#if defined(__HIP_DEVICE_COMPILE__)
    const vf2 t = x * mk2(w.x, w.x);
    vf2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(x), "v"(w), "v"(t));
    return r;
#else
    const float tr = x.x * w.x, ti = x.y * w.x;
    return mk2(std::fmaf(-x.y, w.y, tr), std::fmaf(x.x, w.y, ti));
#endif
}
SM_HD vf2 cx_add_mi(vf2 a, vf2 b) {              // a + (-i) b = (a.re + b.im, a.im - b.re)
#if defined(__HIP_DEVICE_COMPILE__)
    vf2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return mk2(a.x + b.y, a.y - b.x);
#endif
}
SM_HD vf2 cx_sub_mi(vf2 a, vf2 b) {              // a - (-i) b = (a.re - b.im, a.im + b.re)
#if defined(__HIP_DEVICE_COMPILE__)
    vf2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return mk2(a.x - b.y, a.y + b.x);
#endif
}
// x * (c + i s), c and s compile-time constants (an SGPR pair on the device)
SM_HD vf2 cx_mul_cs(vf2 x, float c, float s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_fma(mk2(x.y, x.x), mk2(-s, s), x * c);
#else
    return mk2(std::fmaf(x.y, -s, x.x * c), std::fmaf(x.x, s, x.y * c));
#endif
}
// x * W_M^E (M = 32 or 28; compile-time exponent)
template <int E, int M>
SM_HD vf2 cx_mul_w(vf2 x) {
    constexpr int e = ((E % M) + M) % M;
    if constexpr (e == 0) return x;
    else if constexpr (4 * e == M) return cx_add_mi(vzero<vf2>(), x);          // * (-i)
    else if constexpr (2 * e == M) return -x;
    else if constexpr (4 * e == 3 * M) return cx_sub_mi(vzero<vf2>(), x);      // * (+i)
    else if constexpr (M == 32) return cx_mul_cs(x, W32_RE[e], W32_IM[e]);
    else return cx_mul_cs(x, W28_RE[e], W28_IM[e]);
}

template <int R> struct CDft;
template <> struct CDft<1> { static SM_HD void run(vf2*) {} };
template <> struct CDft<2> {
    static SM_HD void run(vf2* z) { const vf2 a = z[0]; z[0] = a + z[1]; z[1] = a - z[1]; }
};
template <> struct CDft<4> {
    static SM_HD void run(vf2* z) {
        const vf2 t0 = z[0] + z[2], t1 = z[0] - z[2], t2 = z[1] + z[3], d = z[1] - z[3];
        z[0] = t0 + t2; z[2] = t0 - t2;
        z[1] = cx_add_mi(t1, d); z[3] = cx_sub_mi(t1, d);
    }
};
// N = A*B Cooley-Tukey as DftComposite; M: the twiddle table's modulus (32 for the power-of-two sizes, 28 for 4 x 7)
template <int A, int B, int M>
struct CDftComposite {
    static SM_HD void run(vf2* z) {
        constexpr int N = A * B;
        static_assert(M % N == 0, "twiddle table");
        vf2 y[N];
        static_for<0, B>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            vf2 t[A];
            static_for<0, A>([&](auto a_) { constexpr int a = decltype(a_)::value; t[a] = z[B * a + b]; });
            CDft<A>::run(t);
            static_for<0, A>([&](auto k_) {
                constexpr int k1 = decltype(k_)::value;
                y[k1 * B + b] = cx_mul_w<(b * k1) * (M / N), M>(t[k1]);
            });
        });
        static_for<0, A>([&](auto k_) {
            constexpr int k1 = decltype(k_)::value;
            vf2 t[B];
            static_for<0, B>([&](auto b_) { constexpr int b = decltype(b_)::value; t[b] = y[k1 * B + b]; });
            CDft<B>::run(t);
            static_for<0, B>([&](auto k2_) { constexpr int k2 = decltype(k2_)::value; z[k1 + A * k2] = t[k2]; });
        });
    }
};
template <> struct CDft<8> { static SM_HD void run(vf2* z) { CDftComposite<2, 4, 32>::run(z); } };
template <> struct CDft<16> { static SM_HD void run(vf2* z) { CDftComposite<4, 4, 32>::run(z); } };
template <> struct CDft<32> { static SM_HD void run(vf2* z) { CDftComposite<4, 8, 32>::run(z); } };
template <int P>
struct CDftOdd {
    static SM_HD void run(vf2* z) {
        constexpr int H = (P - 1) / 2;
        vf2 sm[H + 1], df[H + 1];
        const vf2 x0 = z[0];
        vf2 acc = x0;
        static_for<1, H + 1>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            sm[j] = z[j] + z[P - j]; df[j] = z[j] - z[P - j];
            acc = acc + sm[j];
        });
        z[0] = acc;
        static_for<1, H + 1>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            vf2 pp = x0, qq = vzero<vf2>();
            static_for<1, H + 1>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                constexpr float c = OddTab<P>::c((j * k) % P);
                constexpr float s = OddTab<P>::s((j * k) % P);
                pp = pp + sm[j] * c; qq = qq + df[j] * s;
            });
            z[k] = cx_add_mi(pp, qq); z[P - k] = cx_sub_mi(pp, qq);
        });
    }
};
template <> struct CDft<3> { static SM_HD void run(vf2* z) { CDftOdd<3>::run(z); } };
template <> struct CDft<5> { static SM_HD void run(vf2* z) { CDftOdd<5>::run(z); } };
template <> struct CDft<7> { static SM_HD void run(vf2* z) { CDftOdd<7>::run(z); } };
template <> struct CDft<11> { static SM_HD void run(vf2* z) { CDftOdd<11>::run(z); } };
template <> struct CDft<13> { static SM_HD void run(vf2* z) { CDftOdd<13>::run(z); } };
template <> struct CDft<28> { static SM_HD void run(vf2* z) { CDftComposite<4, 7, 28>::run(z); } };

// radices the planner may use (keep in sync with plan_fft in smhip_host.cpp)
#define SM_RADIX_SWITCH(r, ...)                                  \
    switch (r) {                                                 \
        case 1:  { constexpr int RX = 1;  __VA_ARGS__; } break;  \
        case 2:  { constexpr int RX = 2;  __VA_ARGS__; } break;  \
        case 3:  { constexpr int RX = 3;  __VA_ARGS__; } break;  \
        case 4:  { constexpr int RX = 4;  __VA_ARGS__; } break;  \
        case 5:  { constexpr int RX = 5;  __VA_ARGS__; } break;  \
        case 7:  { constexpr int RX = 7;  __VA_ARGS__; } break;  \
        case 8:  { constexpr int RX = 8;  __VA_ARGS__; } break;  \
        case 11: { constexpr int RX = 11; __VA_ARGS__; } break;  \
        case 13: { constexpr int RX = 13; __VA_ARGS__; } break;  \
        case 16: { constexpr int RX = 16; __VA_ARGS__; } break;  \
        case 28: { constexpr int RX = 28; __VA_ARGS__; } break;  \
        case 32: { constexpr int RX = 32; __VA_ARGS__; } break;  \
        default: break;                                          \
    }

// ---- Stockham pass pieces, per thread ---------------------------------------
// Pass with radix R and sub-transform size Ns: butterfly j (0 <= j < N/R) reads
// in[j + i*N/R], multiplies by W_{Ns*R}^{i*(j mod Ns)}, does an R-point DFT and
// writes out[(j - j mod Ns)*R + (j mod Ns) + i*Ns].  Thread t owns butterflies
// j = t + m*T; their values sit in x[m*R + i].

// lpad(b + c) = lpad(b) + lpad(c) whenever b is a multiple of 32, so every LDS
// address below is "one thread-dependent base + a compile-time constant": with a
// static plan the constants fold into the ds_read/ds_write immediate offsets
// (no per-element address registers, no per-element address arithmetic).

template <int R>
SM_HD void pass_gather(float* x, const float* lds, int N, int T, int t) {
    constexpr int MB = EMAX / R;
    const int nb = N / R;
    if ((nb & 31) == 0) {
        const int step = nb + (nb >> 5);               // lpad(i*nb) = i*step
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int j = t + m * T;
            if (j < nb) {
                const float* b = lds + lpad(j);
#pragma unroll
                for (int i = 0; i < R; ++i) x[m * R + i] = b[i * step];
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int j = t + m * T;
            if (j < nb) {
#pragma unroll
                for (int i = 0; i < R; ++i) x[m * R + i] = lds[lpad(j + i * nb)];
            }
        }
    }
}

template <int R>
SM_HD void pass_scatter(const float* x, float* lds, int N, int Ns, int T, int t) {
    constexpr int MB = EMAX / R;
    const int nb = N / R;
    const bool big = (Ns & 31) == 0;                                   // i*Ns is a multiple of 32
    const bool small = Ns < 32 && (32 % Ns) == 0 && ((Ns * R) & 31) == 0;   // (j-k)*R is a multiple of 32, k + i*Ns never carries
    if (big || small) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int j = t + m * T;
            if (j < nb) {
                const int k = j % Ns;
                const int base0 = (j - k) * R;
                float* b = lds + (big ? lpad(base0 + k) : lpad(base0) + k);
#pragma unroll
                for (int i = 0; i < R; ++i) b[i * Ns + ((i * Ns) >> 5)] = x[m * R + i];
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int j = t + m * T;
            if (j < nb) {
                const int k = j % Ns;
                const int base = (j - k) * R + k;
#pragma unroll
                for (int i = 0; i < R; ++i) lds[lpad(base + i * Ns)] = x[m * R + i];
            }
        }
    }
}

template <class V>
SM_HD void cmul(V& ar, V& ai, V br, V bi) {
    const V r = ar * br - ai * bi;
    ai = ar * bi + ai * br;
    ar = r;
}

// twiddle table entry for one butterfly (float) or for the two of a packed pair (vf2)
SM_HD void tw_load(const cf2* tw, int idx, int, float& wr, float& wi) { const cf2 w = tw[idx]; wr = w.x; wi = w.y; }
SM_HD void tw_load(const cf2* tw, int idx0, int idx1, vf2& wr, vf2& wi) {
    const cf2 w0 = tw[idx0], w1 = tw[idx1];
    wr = mk2(w0.x, w1.x); wi = mk2(w0.y, w1.y);
}
template <class V> SM_HD V vone();
template <> SM_HD float vone<float>() { return 1.f; }
template <> SM_HD vf2 vone<vf2>() { return mk2(1.f, 1.f); }

// multiply x[i] by w^i, i = 1..R-1, w = W_{Ns*R}^k.  Only the powers w^(2^b) are
// loaded from the table (exact to half an ulp); the others are products of at
// most three of them, so a radix-16 butterfly costs 4 table loads instead of 15
// and keeps 16 instead of 30 registers of twiddles alive.  (A base-4 digit
// variant with 12 registers measured slower on MI355X: more multiplies and, in
// this code, more spills.)  kidx1 is the second lane's index when V = vf2.
template <int R, class V>
SM_HD void apply_twiddles(V* xr, V* xi, const cf2* tw, int kidx0, int kidx1) {
    constexpr int LOGR = R <= 2 ? 1 : R <= 4 ? 2 : R <= 8 ? 3 : R <= 16 ? 4 : 5;
    constexpr int HALF = 1 << (LOGR - 1);              // top power of two used
    V pr[LOGR], pi[LOGR];
#pragma unroll
    for (int b = 0; b < LOGR; ++b) {
        if ((1 << b) < R) tw_load(tw, kidx0 << b, kidx1 << b, pr[b], pi[b]);
        else { pr[b] = vone<V>(); pi[b] = vzero<V>(); }
    }
    // low powers w^1 .. w^(HALF-1)
    V lr[HALF], li[HALF];
    lr[0] = vone<V>(); li[0] = vzero<V>();
    static_for<1, HALF>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int top = (i >= 16) ? 4 : (i >= 8) ? 3 : (i >= 4) ? 2 : (i >= 2) ? 1 : 0;
        constexpr int rest = i - (1 << top);
        if constexpr (rest == 0) { lr[i] = pr[top]; li[i] = pi[top]; }
        else { lr[i] = lr[rest]; li[i] = li[rest]; cmul(lr[i], li[i], pr[top], pi[top]); }
    });
    static_for<1, R>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int lo = i & (HALF - 1);
        if constexpr (lo != 0) cmul(xr[i], xi[i], lr[lo], li[lo]);
        if constexpr (i >= HALF) cmul(xr[i], xi[i], pr[LOGR - 1], pi[LOGR - 1]);
    });
}

// the complex-packed form of apply_twiddles: the same table entries and the same products
template <int R>
SM_HD void apply_twiddles_cx(vf2* z, const cf2* tw, int kidx) {
    constexpr int LOGR = R <= 2 ? 1 : R <= 4 ? 2 : R <= 8 ? 3 : R <= 16 ? 4 : 5;
    constexpr int HALF = 1 << (LOGR - 1);
    vf2 pw[LOGR];
#pragma unroll
    for (int b = 0; b < LOGR; ++b) {
        if ((1 << b) < R) { const cf2 w = tw[kidx << b]; pw[b] = mk2(w.x, w.y); }
        else pw[b] = mk2(1.f, 0.f);
    }
    vf2 lw[HALF];
    lw[0] = mk2(1.f, 0.f);
    static_for<1, HALF>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int top = (i >= 16) ? 4 : (i >= 8) ? 3 : (i >= 4) ? 2 : (i >= 2) ? 1 : 0;
        constexpr int rest = i - (1 << top);
        if constexpr (rest == 0) lw[i] = pw[top];
        else lw[i] = cx_mul(lw[rest], pw[top]);
    });
    static_for<1, R>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int lo = i & (HALF - 1);
        if constexpr (lo != 0) z[i] = cx_mul(z[i], lw[lo]);
        if constexpr (i >= HALF) z[i] = cx_mul(z[i], pw[LOGR - 1]);
    });
}

// PACK: 0 - single floats; 1 - two butterflies of a thread side by side in a vf2; 2 - complex-packed (re, im)
template <int R, int PACK = 1>
SM_HD void pass_compute(float* xr, float* xi, int N, int Ns, int T, int t, const cf2* tw) {
    constexpr int MB = EMAX / R;
    const int nb = N / R;
    const int tstep = N / (Ns * R);
    if constexpr (PACK == 2) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int j = t + m * T;
            if (j < nb) {
                vf2 z[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    // (laundered: LLVM's VectorCombine widens "load float, insert at lane 0" into a <2 x float> load of
                    //  the register array, which then stays in scratch memory)
                    float a = xr[m * R + i], b = xi[m * R + i];
                    SM_LAUNDER(a); SM_LAUNDER(b);
                    z[i] = mk2(a, b);
                }
                if (Ns > 1) {
                    const int k = j % Ns;
                    if constexpr (R <= 32 && R != 28) {
                        apply_twiddles_cx<R>(z, tw, k * tstep);
                    } else {
#pragma unroll
                        for (int i = 1; i < R; ++i) {
                            const cf2 w = tw[i * k * tstep];
                            z[i] = cx_mul(z[i], mk2(w.x, w.y));
                        }
                    }
                }
                CDft<R>::run(z);
#pragma unroll
                for (int i = 0; i < R; ++i) { xr[m * R + i] = z[i].x; xi[m * R + i] = z[i].y; }
            }
        }
    } else if constexpr (PACK == 1 && MB >= 2 && MB % 2 == 0 && R <= 16) {
        // the thread's butterflies j = t + m*T go through the same arithmetic: two at a time,
        // side by side in the lanes of a vf2 (the second of a pair may lie beyond the last
        // butterfly: its lane computes on stale values that nobody stores)
#pragma unroll
        for (int m = 0; m < MB; m += 2) {
            const int j0 = t + m * T, j1 = j0 + T;
            if (j0 < nb) {
                vf2 re[R], im[R];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    re[i] = mk2(xr[m * R + i], xr[(m + 1) * R + i]);
                    im[i] = mk2(xi[m * R + i], xi[(m + 1) * R + i]);
                }
                if (Ns > 1) apply_twiddles<R>(re, im, tw, (j0 % Ns) * tstep, (j1 % Ns) * tstep);
                Dft<R>::run(re, im);
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    xr[m * R + i] = re[i].x; xr[(m + 1) * R + i] = re[i].y;
                    xi[m * R + i] = im[i].x; xi[(m + 1) * R + i] = im[i].y;
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int j = t + m * T;
            if (j < nb) {
                if (Ns > 1) {
                    const int k = j % Ns;
                    if constexpr (R <= 32 && R != 28) {        // every radix a plan uses with twiddles
                        apply_twiddles<R>(xr + m * R, xi + m * R, tw, k * tstep, 0);
                    } else {
#pragma unroll
                        for (int i = 1; i < R; ++i) {
                            const cf2 w = tw[i * k * tstep];
                            cmul(xr[m * R + i], xi[m * R + i], w.x, w.y);
                        }
                    }
                }
                if constexpr (R == 32 && PACK == 0) DftComposite<4, 8>::run(xr + m * R, xi + m * R);
                else Dft<R>::run(xr + m * R, xi + m * R);
            }
        }
    }
}

// component selector with a compile-time index: keeps every register-array
// access statically indexed (a runtime `comp ? xi : xr` pointer sends the arrays
// to scratch memory).
template <int COMP, class S> SM_HD float* comp_of(S& s) { if constexpr (COMP == 0) return s.xr; else return s.xi; }

// ---- plans -------------------------------------------------------------------
// DynPlan: length, thread count and radix list are run-time values (any
// supported length; the radix of each pass is dispatched with a switch).
// SPlan<N, T, R...>: everything is a compile-time constant, the passes are
// unrolled into straight-line code and all index arithmetic folds.  The hot
// lengths (powers of two up to 16384, 7*2^11, 7*2^12) get an SPlan instantiation.
struct DynPlan { static constexpr bool is_static = false; static constexpr bool cx = false; static constexpr int waves = 4; };
// CX_ (complex float2 exchanges) and WAVES_ (occupancy target) are experiment knobs
// that measured slower on MI355X; both are ignored by the engine now.
template <int N_, int T_, bool CX_, int WAVES_, int... RS>
struct SPlan {
    static constexpr bool is_static = true;
    static constexpr bool cx = false;
    static constexpr int waves = WAVES_;
    static constexpr int N = N_, T = T_, npass = (int)sizeof...(RS);
    static constexpr int radix(int i) { constexpr int r[] = {RS...}; return r[i]; }
    static constexpr int ns(int p) { int v = 1; for (int i = 0; i < p; ++i) v *= radix(i); return v; }
    static constexpr int lds_floats = ((N_ + (N_ >> 5) + 1 + 31) / 32) * 32;
    static_assert(ns(npass) == N_, "radices must multiply to N");
};
template <class P> SM_HD int plan_N(const FftPlanDev& pl) { if constexpr (P::is_static) return P::N; else return pl.N; }
template <class P> SM_HD int plan_T(const FftPlanDev& pl) { if constexpr (P::is_static) return P::T; else return pl.T; }
template <class P> SM_HD int plan_lds(const FftPlanDev& pl) { if constexpr (P::is_static) return P::lds_floats; else return pl.lds_floats; }

// Run a whole transform for every group of a work-group.
//   nat_scatter(tid, state, comp_c): write the loaded, natural-order values of
//        component comp (0 = re, 1 = im; passed as std::integral_constant) into
//        LDS (element n of group g at lds + g*lds_floats + lpad(n)).
//   fin_gather(tid, state, comp_c): read from LDS (natural order X[k]) what the
//        storer needs of component comp.
// State must expose float xr[EREG], xi[EREG].
// PACK (see pass_compute): 0 keeps every butterfly on single floats, 1 pairs a thread's butterflies (twice the twiddle
// registers: the inverse column pass and the 128-VGPR row plans spill with it), 2 packs (re, im)
// diagnostic builds (tools/build_variant.sh + tools/ab_kprof.sh): -DSM_DIAG_NOFFT=<mask> drops the radix passes of
// the kernels whose bit is set (1 f1, 2 f2, 4 f2s, 8 i1, 16 i2): their memory traffic and ONE exchange remain
#ifndef SM_DIAG_NOFFT
#define SM_DIAG_NOFFT 0
#endif
struct FloatPairRef { float& a; float& b; };      // what a lane_pair_trade getter returns
struct NoHook { template <class C> SM_HD void operator()(C) const {} };
// after_nat(comp_c): called once per component between the barrier behind the natural scatter and the next one - LDS
// holds the natural-order values of that component and may be READ (a work-group level hook: it may run collectives)
// SKIPS (static plans): bit 0 - the caller has put the values in FIRST-PASS layout already (x[m*R0 + i] = in[j + i*N/R0],
// j = t + m*T < N/R0): no natural scatter / first gather, nat_scatter is never called; bit 1 - the caller takes the
// result in LAST-PASS layout (x[m*RL + i] = out[j + i*N/RL], j = t + m*T < N/RL): no final scatter / gather, fin_gather
// is never called.  Each saves one of the transform's exchanges through LDS (two barriers and 2 x N dwords per component).
template <class P, int PACK = 1, int DIAG = 0, int SKIPS = 0, class Ex, class StT, class NatScatter, class FinGather, class AfterNat = NoHook>
SM_HD void wg_fft(Ex& ex, StT& st, const FftPlanDev& pl, float* lds, NatScatter nat_scatter, FinGather fin_gather, AfterNat after_nat = AfterNat{}) {
    using S = typename StT::value_type;
    const int N = plan_N<P>(pl), T = plan_T<P>(pl), LF = plan_lds<P>(pl);
    if constexpr ((SM_DIAG_NOFFT & DIAG) != 0) {
        static_for<0, 2>([&](auto comp_c) {
            ex.each(st, [&](int tid, S& s) { nat_scatter(tid, s, comp_c); });
            ex.sync();
            after_nat(comp_c);
            ex.each(st, [&](int tid, S& s) { fin_gather(tid, s, comp_c); });
            ex.sync();
        });
        (void)N; (void)T; (void)LF;
        return;
    }
    static_assert(SKIPS == 0 || P::is_static, "layout skips need a static plan");
    if constexpr (P::is_static) {
        if constexpr (!(SKIPS & 1))
        static_for<0, 2>([&](auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            ex.each(st, [&](int tid, S& s) { nat_scatter(tid, s, comp_c); });
            ex.sync();
            if constexpr (!std::is_same<AfterNat, NoHook>::value) {
                SM_SCHED_FENCE();
                after_nat(comp_c);              // (before the gather: the pass's 32 values are not live yet)
                SM_SCHED_FENCE();
            }
            ex.each(st, [&](int tid, S& s) {
                pass_gather<P::radix(0)>(comp_of<comp>(s), lds + (tid / T) * LF, N, T, tid % T);
            });
            ex.sync();
        });
        static_for<0, P::npass>([&](auto p_c) {
            constexpr int p = decltype(p_c)::value;
            constexpr int r = P::radix(p);
            constexpr int Ns = P::ns(p);
            constexpr bool last = (p + 1 == P::npass);
            ex.each(st, [&](int tid, S& s) { pass_compute<r, PACK>(s.xr, s.xi, N, Ns, T, tid % T, pl.tw); });
            if constexpr (!(last && (SKIPS & 2)))
            static_for<0, 2>([&](auto comp_c) {
                constexpr int comp = decltype(comp_c)::value;
                ex.each(st, [&](int tid, S& s) {
                    pass_scatter<r>(comp_of<comp>(s), lds + (tid / T) * LF, N, Ns, T, tid % T);
                });
                ex.sync();
                if constexpr (!last) {
                    ex.each(st, [&](int tid, S& s) {
                        pass_gather<P::radix(last ? p : p + 1)>(comp_of<comp>(s), lds + (tid / T) * LF, N, T, tid % T);
                    });
                } else {
                    ex.each(st, [&](int tid, S& s) { fin_gather(tid, s, comp_c); });
                }
                ex.sync();
            });
        });
    } else {
        // natural -> first pass layout, one component at a time
        static_for<0, 2>([&](auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            ex.each(st, [&](int tid, S& s) { nat_scatter(tid, s, comp_c); });
            ex.sync();
            ex.each(st, [&](int tid, S& s) {
                const float* l = lds + (tid / T) * LF;
                SM_RADIX_SWITCH(pl.radix[0], pass_gather<RX>(comp_of<comp>(s), l, N, T, tid % T));
            });
            after_nat(comp_c);
            ex.sync();
        });
        int Ns = 1;
        for (int p = 0; p < pl.npass; ++p) {
            const int r = pl.radix[p];
            ex.each(st, [&](int tid, S& s) {
                SM_RADIX_SWITCH(r, (pass_compute<RX, PACK>(s.xr, s.xi, N, Ns, T, tid % T, pl.tw)));
            });
            const bool last = (p + 1 == pl.npass);
            static_for<0, 2>([&](auto comp_c) {
                constexpr int comp = decltype(comp_c)::value;
                ex.each(st, [&](int tid, S& s) {
                    float* l = lds + (tid / T) * LF;
                    SM_RADIX_SWITCH(r, pass_scatter<RX>(comp_of<comp>(s), l, N, Ns, T, tid % T));
                });
                ex.sync();
                if (!last) {
                    const int rn = pl.radix[p + 1];
                    ex.each(st, [&](int tid, S& s) {
                        const float* l = lds + (tid / T) * LF;
                        SM_RADIX_SWITCH(rn, pass_gather<RX>(comp_of<comp>(s), l, N, T, tid % T));
                    });
                } else {
                    ex.each(st, [&](int tid, S& s) { fin_gather(tid, s, comp_c); });
                }
                ex.sync();
            });
            Ns *= r;
        }
    }
}

}  // namespace smhip
