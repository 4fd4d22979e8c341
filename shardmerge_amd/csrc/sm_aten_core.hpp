// sm_aten_core.hpp - the pieces of the torch.norm emulation (sm_aten_norm.hpp has the story) that the ROW PASS needs
// as well: since round 4 the forward row pass summarises the deltas it has just formed (k_f1 / k_f1q, AtenFuse below)
// instead of a separate kernel reading every finetune and base a second time.
#pragma once
#include <cmath>
#include <cstring>

#include "fft_engine.hpp"

namespace smhip {

SM_HD float aten_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
SM_HD uint32_t aten_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

constexpr int ATEN_LANES = 8;                  // Vectorized<float>::size() of ATen's AVX2 build
constexpr int ATEN_THREADS = 256;
constexpr int ATEN_ROWS_PER_THREAD = 32;       // rows of 8 elements per thread
constexpr int ATEN_CHUNK_ROWS = ATEN_THREADS * ATEN_ROWS_PER_THREAD;      // 8192 rows = 65536 elements
constexpr int ATEN_MAX_SIGS = 16;
constexpr int ATEN_NO_EXP = -32768;

// f(m) = m + A + (m & 1 ? B1 : B0); pA = A & 1 rides in bit 31 of B1p; A = +inf: no summary ("stop")
struct AtenSum { double A; uint32_t B0, B1p; };
SM_HD AtenSum aten_sum_identity() { AtenSum r; r.A = 0.0; r.B0 = 0u; r.B1p = 0u; return r; }
SM_HD AtenSum aten_sum_stop() { AtenSum r; r.A = INFINITY; r.B0 = 0u; r.B1p = 0u; return r; }
SM_HD AtenSum aten_compose(const AtenSum& l, const AtenSum& r) {      // l first, then r
    const uint32_t pl = l.B1p >> 31, lb1 = l.B1p & 0x7fffffffu;
    const uint32_t pr = r.B1p >> 31, rb1 = r.B1p & 0x7fffffffu;
    const uint32_t mid0 = (pl ^ l.B0) & 1u, mid1 = (1u ^ pl ^ lb1) & 1u;
    AtenSum o;
    o.A = l.A + r.A;
    o.B0 = l.B0 + (mid0 ? rb1 : r.B0);
    const uint32_t b1 = lb1 + (mid1 ? rb1 : r.B0);
    o.B1p = (b1 & 0x7fffffffu) | ((pl ^ pr) << 31);
    return o;
}
SM_HD double aten_apply(double m, uint32_t odd, const AtenSum& f) {
    return m + f.A + (double)(odd ? (f.B1p & 0x7fffffffu) : f.B0);
}
SM_HD double aten_pow2(int e) {                 // 2^e as a double, -1022 <= e <= 1023
    const unsigned long long b = (unsigned long long)(1023 + e) << 52;
    double d; memcpy(&d, &b, 8); return d;
}
constexpr int ATEN_GROUPS = 32;                 // groups of 8 threads = 256 rows per chunk
constexpr int ATEN_GROUP_ROWS = ATEN_CHUNK_ROWS / ATEN_GROUPS;
SM_HD int aten_exp_of(double v) {               // floor(log2 v) for a normal v > 0, else ATEN_NO_EXP
    unsigned long long b; memcpy(&b, &v, 8);
    const int ex = (int)((b >> 52) & 0x7ffu);
    if ((b >> 63) || ex == 0 || ex == 0x7ff) return ATEN_NO_EXP;
    const int e = ex - 1023;
    return (e < -100 || e > 127) ? ATEN_NO_EXP : e;
}
constexpr double ATEN_LAG_MARGIN = 0.08;        // how far below the estimated prefix the running sum is allowed for (its bias is
                                                // -3 % of the sum at 235 M elements; beyond the margin a chunk is walked cooperatively)
constexpr double ATEN_LEAD_MARGIN = 0.02;       // ... and how far above (lattice data round UP on balance; the estimate
                                                // comes from a sample)
// The same summary, EVALUATED instead of derived (round 4).  A step's increment of m = S / u depends on the running
// sum only through its binade (u) and its parity (a tie goes to the even neighbour), and the parity after the step
// follows from the parity before it: by induction the total increment D of a run depends on the START parity alone.
// Two real chains of fmas over the run - one from m = 2^23 (even), one from m = 2^23 + 1 (odd) in the predicted
// binade - therefore give D0 = A + B0 and D1 = A + B1 exactly, with the hardware's own rounding: one v_pk_fma_f32 per
// element and candidate where the derivation above takes ~20 instructions, half of them in double precision.  A chain
// that leaves the binade (an outlier, Inf, NaN) voids the summary, as an A >= 2^24 did.
SM_HD vf2 aten_chain_start(int e) {
    const uint32_t b = (uint32_t)(e + 127) << 23;
    return mk2(aten_u2f(b), aten_u2f(b + 1u));
}
SM_HD void aten_chain_add(vf2& s, float y) {
#if defined(__HIP_DEVICE_COMPILE__)
    const vf2 yy = mk2(y, y);
    s = __builtin_elementwise_fma(yy, yy, s);
#else
    s.x = std::fmaf(y, y, s.x); s.y = std::fmaf(y, y, s.y);
#endif
}
SM_HD AtenSum aten_chain_sum(vf2 s, int e) {
    const uint32_t b = (uint32_t)(e + 127) << 23;
    const uint32_t f0 = aten_f2u(s.x), f1 = aten_f2u(s.y);
    if ((f0 >> 23) != (b >> 23) || (f1 >> 23) != (b >> 23)) return aten_sum_stop();      // left the binade (or Inf / NaN)
    const uint32_t d0 = f0 - b, d1 = f1 - (b + 1u);
    const uint32_t a = d0 < d1 ? d0 : d1;
    AtenSum r; r.A = (double)a; r.B0 = d0 - a; r.B1p = ((d1 - a) & 0x7fffffffu) | ((a & 1u) << 31);
    return r;
}

// ---- the forward row pass summarises the deltas it forms (round 4) ------------------------------------------------
// torch.norm flattens the tensor row-major and element i goes to lane i % 8: with C % 8 == 0 the lane of an element is
// its column modulo 8.  After the natural scatter of a row pass, LDS holds one matrix row in natural order: wave w of
// the row's thread group takes the rows-of-8 [256 w, 256 w + 256) - one GROUP of the walker (ATEN_GROUP_ROWS) - thread
// (c, run) of the wave evaluates lane c over 32 consecutive rows-of-8 (two chains per candidate binade, 4 registers),
// three ordered compositions across the wave's 8 runs give the group's summary, which goes straight to `grp`.
// k_aten_rec composes the chunks' 32 groups into `rec` afterwards.  Needs C % 2048 == 0 and C / 2048 <= T / 64.
struct AtenFuse {
    const double* prefix;       // [nsig][nchunks][8]: estimated inclusive prefix of the lane sums (k_aten_pre, k_aten_scan); null: off
    AtenSum* grp;               // [nsig][nchunks][8][2][ATEN_GROUPS]
    unsigned long long nchunks;
    int sig0;                   // this launch's first signal in those arrays
};
#ifndef SM_FUSE_NORMS
#define SM_FUSE_NORMS 1            // 0: the row passes are compiled without the summaries (A/B builds)
#endif
SM_HD constexpr bool aten_fusable(int C, int T) { return SM_FUSE_NORMS && C % (8 * ATEN_GROUP_ROWS) == 0 && C / (8 * ATEN_GROUP_ROWS) <= T / 64; }

// In the kernel a summary travels as the pair (D0, D1) = total increment for an even / an odd start (8 bytes instead of
// AtenSum's 16): composition D_p = Dl_p + Dr_{(p + Dl_p) & 1}; anything >= 2^24 is void (m + D would leave the binade
// whatever m is) and is clamped there, so sums of voids cannot wrap.
constexpr uint32_t ATEN_D_VOID = 1u << 24;
SM_HD void aten_d_of_chain(vf2 s, int e, uint32_t& d0, uint32_t& d1) {
    const uint32_t b = (uint32_t)(e + 127) << 23;
    const uint32_t f0 = aten_f2u(s.x), f1 = aten_f2u(s.y);
    const bool in = (f0 >> 23) == (b >> 23) && (f1 >> 23) == (b >> 23);
    d0 = in ? f0 - b : ATEN_D_VOID; d1 = in ? f1 - (b + 1u) : ATEN_D_VOID;
}
SM_HD void aten_d_compose(uint32_t& l0, uint32_t& l1, uint32_t r0, uint32_t r1) {          // l first, then r
    const uint32_t n0 = l0 + ((l0 & 1u) ? r1 : r0), n1 = l1 + ((l1 & 1u) ? r0 : r1);     // (odd start + odd increment -> even)
    l0 = n0 < ATEN_D_VOID ? n0 : ATEN_D_VOID; l1 = n1 < ATEN_D_VOID ? n1 : ATEN_D_VOID;
}
SM_HD AtenSum aten_sum_of_d(uint32_t d0, uint32_t d1) {
    if (d0 >= ATEN_D_VOID || d1 >= ATEN_D_VOID) return aten_sum_stop();
    const uint32_t a = d0 < d1 ? d0 : d1;
    AtenSum r; r.A = (double)a; r.B0 = d0 - a; r.B1p = ((d1 - a) & 0x7fffffffu) | ((a & 1u) << 31);
    return r;
}

// the predicted binade for the group that thread `tid` of a row pass works on in component `comp` (ATEN_NO_EXP: none /
// not this thread's business): looked up at the START of the kernel, with the operand loads - a global load inside the
// hook would wait for every load in flight (the second operand's prefetch)
template <int C, int T, class Where>
SM_HD int aten_fused_ep(const AtenFuse& af, int tid, Where where) {
    constexpr int NG = C / (8 * ATEN_GROUP_ROWS);
    const int g = tid / T, t = tid % T;
    const int w = t >> 6, c = t & 7;
    int sig = 0; long long mrow = 0;
    if (w >= NG || !where(g, sig, mrow)) return ATEN_NO_EXP;
    const size_t G = (size_t)mrow * NG + w, chunk = G / ATEN_GROUPS;
    const size_t slot = ((size_t)(af.sig0 + sig) * af.nchunks + chunk) * 8 + c;
    return aten_exp_of(af.prefix[slot] * (1.0 + ATEN_LEAD_MARGIN));
}

// where(g, sig, mrow) -> bool: region g of LDS (lds + g * LF, T threads) holds matrix row `mrow` of signal `sig`
// (false: nothing to summarise there).  The state type has uint32_t fd[4], fr[4] ((D0, D1) of the two candidates).
// ep_of(S&) -> the thread's prediction for this component (aten_fused_ep, taken at the start of the kernel)
template <int C, int T, class Ex, class StT, class Where, class EpOf>
SM_HD void aten_fused_rows(Ex& ex, StT& st, const AtenFuse& af, const float* lds, int LF, Where where, EpOf ep_of) {
    using S = typename StT::value_type;
    constexpr int NG = C / (8 * ATEN_GROUP_ROWS);
    ex.each(st, [&](int tid, S& s) {
        int tid_ = tid;
        SM_OPAQUE(tid_);                               // (nothing of this hook is carried across the transform)
        const int g = tid_ / T, t = tid_ % T;
        const int w = t >> 6, ln = t & 63, c = ln & 7, run = ln >> 3;
        s.fd[0] = 0u; s.fd[1] = 0u; s.fd[2] = 0u; s.fd[3] = 0u;          // identity
        const int ep = ep_of(s);
        if (w >= NG || ep == ATEN_NO_EXP) return;      // (no prediction: the walker does not use this chunk's summaries)
        vf2 s0 = aten_chain_start(ep), s1 = aten_chain_start(ep - 1);
        const float* l = lds + g * LF;
        const int e0 = 8 * (ATEN_GROUP_ROWS * w + ATEN_ROWS_PER_THREAD * run) + c;
        const float* b = l + lpad(e0);                 // lpad(e0 + 8 k) = lpad(e0) + 8 k + (k >> 2): e0 is c (< 8) past a multiple of 32
        // (eight reads in flight at a time: all 32 at once cost the 128-register row passes their registers)
        static_for<0, ATEN_ROWS_PER_THREAD / 8>([&](auto kb_c) {
            constexpr int k0 = decltype(kb_c)::value * 8;
            float y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = b[8 * (k0 + k) + ((k0 + k) >> 2)];
#pragma unroll
            for (int k = 0; k < 8; ++k) { aten_chain_add(s0, y[k]); aten_chain_add(s1, y[k]); }
            SM_SCHED_FENCE();
        });
        aten_d_of_chain(s0, ep, s.fd[0], s.fd[1]);
        aten_d_of_chain(s1, ep - 1, s.fd[2], s.fd[3]);
    });
    static_for<0, 3>([&](auto step_c) {
        constexpr int DELTA = 8 << decltype(step_c)::value;
        ex.template wave_shift_down<DELTA, 4>(st, [](S& s) { return (const uint32_t*)s.fd; }, [](S& s) { return (uint32_t*)s.fr; });
        ex.each(st, [&](int, S& s) {
            aten_d_compose(s.fd[0], s.fd[1], s.fr[0], s.fr[1]);
            aten_d_compose(s.fd[2], s.fd[3], s.fr[2], s.fr[3]);
        });
    });
    ex.each(st, [&](int tid, S& s) {
        int tid_ = tid;
        SM_OPAQUE(tid_);
        const int g = tid_ / T, t = tid_ % T;
        const int w = t >> 6, ln = t & 63;
        int sig = 0; long long mrow = 0;
        if (ln >= 8 || w >= NG || !where(g, sig, mrow)) return;
        const size_t G = (size_t)mrow * NG + w, chunk = G / ATEN_GROUPS;
        const size_t slot = ((size_t)(af.sig0 + sig) * af.nchunks + chunk) * 8 + ln;
        af.grp[(slot * 2 + 0) * ATEN_GROUPS + G % ATEN_GROUPS] = aten_sum_of_d(s.fd[0], s.fd[1]);
        af.grp[(slot * 2 + 1) * ATEN_GROUPS + G % ATEN_GROUPS] = aten_sum_of_d(s.fd[2], s.fd[3]);
    });
}

}  // namespace smhip
