// sm_bluestein.hpp - row passes for a row length C the engine cannot plan (a prime factor > 13:
// Falcon-7B's 4544 = 71 * 64, 4672 = 73 * 64).  The reference transforms any shape
// (functions.py:45-58: torch.fft.fft2); the column side already splits such a length into p * M
// (k_dftp), and a tensor with ONE rough length is merged transposed so that it becomes the column
// length.  A tensor whose lengths are BOTH rough needs a row transform of arbitrary length:
// Bluestein's chirp-z form of the DFT on the work-group engine,
//
//     X[k] = w[k] * sum_n (x[n] w[n]) * conj(w[k - n]),      w[n] = exp(-i pi n^2 / C)
//
// i.e. a circular convolution of length L >= 2C - 1 (L a power of two with a static plan: two
// L-point transforms per row and three element-wise complex products, everything in registers / LDS
// of one work-group, no extra pass over HBM).  The tables come from the host in double precision
// (n^2 is reduced mod 2C in integers): chirp[n] = w[n], n < C, and filt = FFT_L(h) / L with
// h[m] = conj(w[m]) for |m| < C (indices mod L).  The inverse transform uses
// IDFT(Y) = conj(DFT(conj(Y))) and therefore the same tables.
//
//   k_f1b   forward row pass: the contract of k_f1 (two real rows as one complex sequence, Hermitian
//           split, T1 layout, sum-of-squares partials)
//   k_i2b   inverse row pass: the contract of k_i2 (row pairs out of G, scale, NaN / Inf policy,
//           add-back, bf16 / f32 output, norm partials)
//
// Rough lengths are rare and these kernels are written for correctness first: element-wise global
// accesses, one row (pair) per work-group, 3-4x the arithmetic of a planned length.
#pragma once
#include "sm_kernels.hpp"

namespace smhip {

struct BluesteinTab {
    const cf2* chirp;       // [C]
    const cf2* filt;        // [L], 1/L folded in
};
struct F1BParams { F1Params f; BluesteinTab bt; };      // f.plan: the L-point plan; f.C: the row length
struct I2BParams { I2Params i; BluesteinTab bt; };

// plans the convolution may run on: the power-of-two static plans, or a run-time plan
template <class P> constexpr bool bluestein_plan() {
    if constexpr (P::is_static) return (P::N & (P::N - 1)) == 0;
    else return true;
}

// in: z[n] at slot q of thread tid, n = tid + q * T (zero for n >= C).  out: X[k] = sum_n z[n] exp(-2 pi i n k / C)
// in the same slots, k < C (slots past C hold garbage).  One transform per work-group.
template <class P, class Ex, class StT>
SM_HD void bluestein_dft(Ex& ex, StT& st, const FftPlanDev& pl, float* lds, const BluesteinTab& bt, int C) {
    using S = typename StT::value_type;
    const int T = plan_T<P>(pl), L = plan_N<P>(pl);
    ex.each(st, [&](int tid, S& s) {
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int n = tid + q * T;
            const cf2 w = bt.chirp[n < C ? n : 0];
            const float xr = s.xr[q], xi = s.xi[q];
            const bool in = n < C;
            s.xr[q] = in ? xr * w.x - xi * w.y : 0.f;
            s.xi[q] = in ? xr * w.y + xi * w.x : 0.f;
        }
    });
    auto nat_scatter = [&](int tid, S& s, auto comp_c) {
        constexpr int comp = decltype(comp_c)::value;
        const float* x = comp_of<comp>(s);
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int n = tid + q * T;
            if (n < L) lds[lpad(n)] = x[q];
        }
    };
    auto nat_gather = [&](int tid, S& s, auto comp_c) {
        constexpr int comp = decltype(comp_c)::value;
        float* o = comp_of<comp>(s);
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int k = tid + q * T;
            if (k < L) o[q] = lds[lpad(k)];
        }
    };
    wg_fft<P, f1_pack<P>()>(ex, st, pl, lds, nat_scatter, nat_gather);
    // times the filter's spectrum; conjugated: the second forward transform is the inverse one
    ex.each(st, [&](int tid, S& s) {
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int k = tid + q * T;
            const cf2 h = bt.filt[k < L ? k : 0];
            const float xr = s.xr[q], xi = s.xi[q];
            s.xr[q] = xr * h.x - xi * h.y;
            s.xi[q] = -(xr * h.y + xi * h.x);
        }
    });
    wg_fft<P, f1_pack<P>()>(ex, st, pl, lds, nat_scatter, nat_gather);
    ex.each(st, [&](int tid, S& s) {
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int k = tid + q * T;
            const cf2 w = bt.chirp[k < C ? k : 0];
            const float vr = s.xr[q], vi = -s.xi[q];
            s.xr[q] = vr * w.x - vi * w.y;
            s.xi[q] = vr * w.y + vi * w.x;
        }
    });
}

// ---- forward row pass ------------------------------------------------------------------------------
template <class P, class Ex>
SM_HD void k_f1b(Ex& ex, const F1BParams& pp) {
    if constexpr (!bluestein_plan<P>()) { return; } else {
    const F1Params& p = pp.f;
    typename Ex::template State<FftState> st;
    ex.init(st);
    float* lds = ex.lds() + LDS_SCRATCH_FLOATS;
    const FftPlanDev& pl = p.plan;
    const int T = plan_T<P>(pl), C = p.C;
    const int row = (p.ilv > 1) ? xcd_remap(ex.bid(), p.ilv) : ex.bid();      // see k_f1 (one row per work-group here)
    const int pbid = ex.bid();
    const bool valid = row < p.R;

    ex.each(st, [&](int tid, FftState& s) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int n = tid + q * T;
            float va = 0.f, vb = 0.f;
            if (valid && n < C) {
                const size_t off = (size_t)row * p.row_stride + n;
                va = load_sig1(p.a, off);
                vb = load_sig1(p.b, off);
            }
            s.xr[q] = va; s.xi[q] = vb;
            sa += (double)va * va; sb += (double)vb * vb;
        }
        s.red[0] = sa; s.red[1] = sb;
    });
    ex.template block_sum<2>(st, [&](const double* tot) {
        p.partials[2 * (size_t)pbid] = tot[0];
        p.partials[2 * (size_t)pbid + 1] = tot[1];
    });
    ex.sync();

    bluestein_dft<P>(ex, st, pl, lds, pp.bt, C);

    // Z = A + i B (A, B the spectra of the two real rows): pairs (k, C - k) through LDS, one component at a time
    static_for<0, 2>([&](auto comp_c) {
        constexpr int comp = decltype(comp_c)::value;
        ex.each(st, [&](int tid, FftState& s) {
            const float* x = comp_of<comp>(s);
#pragma unroll
            for (int q = 0; q < EMAX; ++q) {
                const int k = tid + q * T;
                if (k < C) lds[lpad(k)] = x[q];
            }
        });
        ex.sync();
        ex.each(st, [&](int tid, FftState& s) {
            float* o = comp_of<comp>(s);
#pragma unroll
            for (int u = 0; u < EMAX / 2 + 1; ++u) {
                const int k = tid + u * T;
                if (k < p.Cb) {
                    const int k2 = (k == 0) ? 0 : C - k;
                    const float v1 = lds[lpad(k)], v2 = lds[lpad(k2)];
                    if (comp == 0) { o[2 * u] = 0.5f * (v1 + v2); o[2 * u + 1] = 0.5f * (v2 - v1); }   // A.re, B.im
                    else           { o[2 * u] = 0.5f * (v1 - v2); o[2 * u + 1] = 0.5f * (v1 + v2); }   // A.im, B.re
                    if (!p.b.x) o[2 * u + 1] = 0.f;                                                    // see k_f1
                }
            }
        });
        ex.sync();
    });

    ex.each(st, [&](int tid, FftState& s) {
        if (!valid) return;
        cf4* dst = p.t1 + (size_t)(row / p.ilv) * p.pitch4 * p.ilv + (row % p.ilv);
#pragma unroll
        for (int u = 0; u < EMAX / 2 + 1; ++u) {
            const int k = tid + u * T;
            if (k < p.Cb) {
                cf4 v;
                v.x = s.xr[2 * u]; v.y = s.xi[2 * u]; v.z = s.xi[2 * u + 1]; v.w = s.xr[2 * u + 1];
                dst[(size_t)k * p.ilv] = v;
            }
        }
    });
    }
}

// ---- inverse row pass ------------------------------------------------------------------------------
template <class P, class Ex>
SM_HD void k_i2b(Ex& ex, const I2BParams& pp) {
    if constexpr (!bluestein_plan<P>()) { return; } else {
    const I2Params& p = pp.i;
    typename Ex::template State<FftState> st;
    ex.init(st);
    float* lds = ex.lds() + LDS_SCRATCH_FLOATS;
    const FftPlanDev& pl = p.plan;
    const int T = plan_T<P>(pl), C = p.C;
    const int bid = ex.bid();
    const int r0 = 2 * bid, r1 = r0 + 1;
    const bool v0 = r0 < p.R, v1 = r1 < p.R;

    ex.each(st, [&](int tid, FftState& s) {
        const cf4* GP = (const cf4*)p.G + (size_t)(v0 ? r0 / 2 : 0) * p.pitchG;
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int k = tid + q * T;
            const bool in = k < C;
            const bool low = 2 * k <= C;
            const int kk = in ? (low ? k : C - k) : 0;
            const cf4 a = GP[kk];
            cf2 g0 = {a.x, a.y}, g1 = {a.z, a.w};
            if (!v0) { g0.x = 0.f; g0.y = 0.f; }
            if (!v1) { g1.x = 0.f; g1.y = 0.f; }
            if (kk == 0 || 2 * kk == C) { g0.y = 0.f; g1.y = 0.f; }      // c2r: DC / Nyquist are real
            // Y[k] = G0[k] + i G1[k];  Y[C - k] = conj(G0[k]) + i conj(G1[k]);  the transform takes conj(Y)
            const float yr = low ? g0.x - g1.y : g0.x + g1.y;
            const float yi = low ? g0.y + g1.x : g1.x - g0.y;
            s.xr[q] = in ? yr : 0.f;
            s.xi[q] = in ? -yi : 0.f;
        }
    });

    bluestein_dft<P>(ex, st, pl, lds, pp.bt, C);

    // y = conj(D): row r0 = Re D, row r1 = -Im D
    ex.each(st, [&](int tid, FftState& s) {
        uint32_t nan1 = 0, inf1 = 0, nan2 = 0, inf2 = 0;
        double ss = 0.0;
        static_for<0, 2>([&](auto h_c) {
            constexpr int h = decltype(h_c)::value;
            const int row = r0 + h;
            if (row >= p.R) return;
#pragma unroll
            for (int q = 0; q < EMAX; ++q) {
                const int n = tid + q * T;
                if (n < C) {
                    const size_t off = (size_t)row * C + n;
                    float v;
                    i2_finish(p, h ? -s.xi[q] : s.xr[q], off, nan1, inf1, nan2, inf2, v);
                    ss += (double)v * v;
                    if (p.out_mode == OUT_BF16) ((uint16_t*)p.out)[off] = f_to_bf16(v);
                    else ((float*)p.out)[off] = v;
                }
            }
        });
        if (nan1) ex.global_atomic_add_u32(&p.flags[0], nan1);
        if (inf1) ex.global_atomic_or_u32(&p.flags[1], 1u);
        if (nan2) ex.global_atomic_add_u32(&p.flags[2], nan2);
        if (inf2) ex.global_atomic_or_u32(&p.flags[3], 1u);
        s.red[0] = ss; s.red[1] = 0.0;
    });
    if (p.norm_partials) {
        ex.sync();
        ex.template block_sum<2>(st, [&](const double* tot) {
            p.norm_partials[2 * (size_t)bid] = tot[0];
            p.norm_partials[2 * (size_t)bid + 1] = 0.0;
        });
    }
    }
}

}  // namespace smhip
