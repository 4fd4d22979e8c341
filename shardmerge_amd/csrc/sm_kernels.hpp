// sm_kernels.hpp - bodies of every kernel of the spectral-merge pipeline.
//
// Each body is a template over an execution policy `Ex`:
//   * DeviceExec (sm_device.hip)  -> a real gfx950 work-group, `each` runs the
//     lambda once for threadIdx.x and `sync` is s_barrier;
//   * HostExec (tests/emul)       -> a sequential CPU emulation of one
//     work-group, used by the "not gpu" tests to check indexing.
// Pipeline for one pair-merge of two real [R x C] tensors a, b:
//   F1  rows:    z = a + i b, C-point FFT per row, split by Hermitian symmetry
//                into half-spectra A_row, B_row -> T1[R][Cb] (float4 per bin)
//   F2  columns: R-point FFT per bin column of A and B -> planes [Cb][R]:
//                Re Fa, Im Fa, Re Fb (normalised), + level-1 histogram
//   SEL select:  exact k-th order statistic by 3-level radix histograms
//   RED / BLEND: masked slerp sums, class blend -> Re R plane (+ cull histogram)
//   I1  columns: inverse R-point FFT of (Re R culled, Im Fa) -> G[R][Cb]
//   I2  rows:    two rows per transform (two-for-one), inverse C-point FFT,
//                scale, NaN/Inf policy, add base, bf16 store
#pragma once
#include <cmath>
#include <cstring>

#include "fft_engine.hpp"
#include "sm_aten_core.hpp"

namespace smhip {

struct cf4 { float x, y, z, w; };
struct u32x4 { uint32_t x, y, z, w; };

enum { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };

// value = (load(x) - (base ? load(base) : 0)) * prescale
struct SigDesc {
    const void* x;
    const void* base;
    int dtype;
    float prescale;
};

SM_HD uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
SM_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
SM_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
SM_HD float bf16_to_f(uint32_t h) { return u2f(h << 16); }
SM_HD float f16_to_f(uint32_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint16_t h16 = (uint16_t)h;                 // v_cvt_f32_f16
    _Float16 hv;
    memcpy(&hv, &h16, 2);
    return (float)hv;
#endif
    const uint32_t s = (h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
    if (e == 0) {
        if (m == 0) return u2f(s);
        float v = (float)m * 5.9604644775390625e-08f;   // m * 2^-24
        return (s ? -v : v);
    }
    if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
    return u2f(s | ((e + 112) << 23) | (m << 13));
}
// round-to-nearest-even; caller guarantees v is not NaN
SM_HD uint16_t f_to_bf16(float v) {
    uint32_t u = f2u(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// two floats -> packed bf16 pair (lo = a), round-to-nearest-even; neither is NaN
SM_HD uint32_t pack_bf16x2(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);      // v_cvt_pk_bf16_f32
    uint32_t u;
    memcpy(&u, &h, 4);
    return u;
#else
    return (uint32_t)f_to_bf16(a) | ((uint32_t)f_to_bf16(b) << 16);
#endif
}
// a ROUNDED fp32 product / sum (the device compiler contracts a * b + c into an fma otherwise)
SM_HD float aten_fmul_(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = a * b; asm volatile("" : "+v"(r)); return r;
#else
    volatile float r = a * b; return r;
#endif
}
SM_HD float aten_fadd_(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = a + b; asm volatile("" : "+v"(r)); return r;
#else
    volatile float r = a + b; return r;
#endif
}
SM_HD bool is_nan(float v) { return v != v; }
SM_HD bool is_inf(float v) { return (f2u(v) & 0x7fffffffu) == 0x7f800000u; }

SM_HD float load_elem(const void* p, int dtype, size_t i) {
    if (dtype == DT_F32) return ((const float*)p)[i];
    const uint32_t h = ((const uint16_t*)p)[i];
    return dtype == DT_BF16 ? bf16_to_f(h) : f16_to_f(h);
}
// 8 consecutive elements starting at i (i % 8 == 0, pointer 16-byte aligned)
SM_HD void load_elem8(const void* p, int dtype, size_t i, float* out) {
    if (dtype == DT_F32) {
        const cf4 a = ((const cf4*)p)[i / 4], b = ((const cf4*)p)[i / 4 + 1];
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
        out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
    } else {
        const u32x4 v = ((const u32x4*)p)[i / 8];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (dtype == DT_BF16) {
                out[2 * c] = u2f(w[c] << 16);
                out[2 * c + 1] = u2f(w[c] & 0xffff0000u);
            } else {
                out[2 * c] = f16_to_f(w[c] & 0xffffu);
                out[2 * c + 1] = f16_to_f(w[c] >> 16);
            }
        }
    }
}
// 8 packed 16-bit values (bf16 or f16) -> float
SM_HD void decode16x8(const u32x4& v, int dtype, float* out) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (dtype == DT_BF16) {
            out[2 * c] = u2f(w[c] << 16);
            out[2 * c + 1] = u2f(w[c] & 0xffff0000u);
        } else {
            out[2 * c] = f16_to_f(w[c] & 0xffffu);
            out[2 * c + 1] = f16_to_f(w[c] >> 16);
        }
    }
}
SM_HD void load_sig8(const SigDesc& s, size_t i, float* out) {
    if (!s.x) {
#pragma unroll
        for (int c = 0; c < 8; ++c) out[c] = 0.f;
        return;
    }
    load_elem8(s.x, s.dtype, i, out);
    if (s.base) {
        float b[8];
        load_elem8(s.base, s.dtype, i, b);
#pragma unroll
        for (int c = 0; c < 8; ++c) out[c] -= b[c];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) out[c] *= s.prescale;
}
SM_HD float load_sig1(const SigDesc& s, size_t i) {
    if (!s.x) return 0.f;
    float v = load_elem(s.x, s.dtype, i);
    if (s.base) v -= load_elem(s.base, s.dtype, i);
    return v * s.prescale;
}

// weight of a half-spectrum bin column: columns 0 and C/2 (C even) hold all of
// their full-spectrum bins, every other column stands for itself and its
// conjugate twin (multiplicities for the order statistics).
// C < 0: the planes hold a full spectrum, every bin counts once.
SM_HD int bin_weight(int k2, int C) { return (C < 0 || k2 == 0 || (2 * k2 == C)) ? 1 : 2; }

// XCD-aware placement for the strided (column) passes.  Work-groups are dealt
// round-robin over the 8 XCDs (blocks b and b+8 share an L2): the G work-groups
// that touch the same 128-byte lines get block ids that are equal mod 8 and
// adjacent in dispatch order, so a line is fetched from HBM once per XCD-L2 and
// partial-line writes merge in that L2.  Speed only, never correctness.
// Returns the logical index for block id `bid`; the grid must be padded to a
// multiple of 8*G and logical indices >= count do nothing.
SM_HD int xcd_remap(int bid, int G) {
    const int per = 8 * G;
    const int y = bid / per, i = (bid % per) / 8, x = bid % 8;
    return (y * 8 + x) * G + i;
}

// multiplicity of plane element i (planes are [Cb][R]): 1 inside column 0 and
// column C/2, else 2 - two range tests instead of a division per element
struct WeightRanges {
    size_t hi0, loN, hiN;
    size_t period;      // > 0: the planes hold a BATCH of half spectra one after the other ((C/2+1)*R elements each)
    int full;
};
// Cb_total: bin columns in the planes; more than C/2 + 1 of them means a batch (rank > 2 tensors:
// the reference transforms the last two dims and takes every statistic over the whole tensor)
SM_HD WeightRanges weight_ranges(int R, int C, int Cb_total = 0) {
    WeightRanges w;
    w.full = C < 0;
    w.hi0 = (size_t)R;
    if (C >= 0 && (C % 2) == 0 && C > 0) { w.loN = (size_t)(C / 2) * R; w.hiN = w.loN + R; }
    else { w.loN = w.hiN = 0; }
    w.period = (C >= 0 && Cb_total > C / 2 + 1) ? (size_t)(C / 2 + 1) * R : 0;
    return w;
}
SM_HD uint32_t weight_at(const WeightRanges& w, size_t i) {
    if (w.period) i %= w.period;
    return (w.full || i < w.hi0 || (i >= w.loN && i < w.hiN)) ? 1u : 2u;
}

struct EmptyStateF { double red[2]; };
struct NormsState { double red[16]; };

struct FftState {
    float xr[EREG];
    float xi[EREG];
    double red[4];
    uint32_t fd[4], fr[4];     // the row pass's fused torch.norm summaries (aten_fused_rows); dead everywhere else
    int fep[2];                // ... and the predicted binades of this thread's group (per component)
};

constexpr int LDS_SCRATCH_FLOATS = 64 * 4;   // reduction scratch at the start of LDS
constexpr int HIST1_BINS = 2048;             // level 1: key >> 20
constexpr int HIST_LO_BINS = 1024;           // levels 2, 3: 10 bits each

// =====================================================================
// F1: forward row pass
// =====================================================================
// rows_first: the row passes of several raw deltas in ONE launch.  The deltas of a layer share their base, so the
// work-groups that transform the same rows of different signals are placed on one XCD, next to each other in dispatch
// order (xcd_remap): the base rows come from HBM once and out of that L2 afterwards (the row pass with its base rows
// always in cache runs 11 % faster: 462 against 520 us per 28672x8192 signal).
constexpr int F1_MAX_SIGS = 16;
struct F1Sigs {
    int n;                              // <= 1: off (the kernel uses a, b, t1, partials)
    const void* x[F1_MAX_SIGS];         // signal s: a.x (b.x = a.x + b_off bytes: row-pair mode)
    const void* base[F1_MAX_SIGS];      // its base (b.base likewise) or null
    cf4* t1[F1_MAX_SIGS];
    double* partials[F1_MAX_SIGS];
    long long b_off;
};
// work-group `wg` of a launch over sg.n signals -> its signal and its block id within that signal's grid (XG: the
// work-groups of ONE signal that share lines of T1 and must stay adjacent, see k_f1)
SM_HD int pick_signal(int n, int wg, int XG, int& sig) {
    sig = 0;
    if (n <= 1) return XG > 1 ? xcd_remap(wg, XG) : wg;
    const int G = XG * n;
    const int L = xcd_remap(wg, G);
    const int r = L % G;
    sig = r / XG;
    return (L / G) * XG + r % XG;
}
SM_HD int f1_pick_signal(const F1Sigs& sg, int wg, int XG, int& sig) { return pick_signal(sg.n, wg, XG, sig); }

struct F1Params {
    FftPlanDev plan;       // N = C
    SigDesc a, b;
    int R, C, Cb;
    int pitch4;            // T1 row pitch in float4
    int ilv;               // rows interleaved in T1 (1 or 2): element (r, k) at ((r/ilv)*pitch4 + k)*ilv + r%ilv
    int nb;                // rows (transforms) per work-group
    size_t row_stride;     // elements between consecutive transforms' inputs (C; 2C in row-pair mode,
                           // where "a" is row 2m and "b" row 2m+1 of ONE tensor: b's pointers are a's + C)
    int vec;               // 1: C % 8 == 0 and 16-byte aligned inputs
    cf4* t1;               // the column pass reads ilv*16 contiguous bytes per (row group, bin)
    double* partials;      // [grid][2]: sum a^2, sum b^2 of this work-group
    // k_f1q only (radix-4 column step folded into the row pass):
    int R2;                // transforms per slab: work-group w handles "rows" w + g*R2, g = 0..3
    int rowpair;           // 1: row-pair mode (a = row 2w, b = row 2w+1 of one tensor; row_stride = 2C)
    int Rcol;              // column length R (the folded twiddles are W_R)
    size_t slab_elems;     // float4 per k1 slab of T1
    const cf2* twR;        // exp(-2 pi i j / R), j < R
    F1Sigs sigs;           // k_f1 / k_f1q: several signals in one launch
    AtenFuse fuse;         // norm_mode = reference_cpu: the deltas' torch.norm summaries come out of this pass (prefix = null: off)
    int fuse_rowpair;      // 1: signal = the launch's signal index, matrix row = 2 * unit + component; 0: signal = component (K = 2)
};

template <class P> constexpr bool f1_full_batch() {
    if constexpr (P::is_static) return P::T <= 256; else return false;
}
// paired (packed-f32) butterflies cost registers: the 512/1024-thread row plans, compiled for
// 128 VGPRs, spill with them (14336: 76 bytes of scratch -> 0, -21 % time) and stay scalar
#ifndef SM_ROW_PACK_MAX_T
#define SM_ROW_PACK_MAX_T 256
#endif
template <class P> constexpr bool f1_opaque() {
    if constexpr (P::is_static) return P::T >= 512; else return false;
}
// ... (PACK mode of wg_fft: 0 single floats, 1 paired butterflies, 2 complex-packed - the scalar form's register count)
#ifndef SM_ROW_PACK_BIG
#define SM_ROW_PACK_BIG 0
#endif
template <class P> constexpr int f1_pack() {
    if constexpr (P::is_static) return P::T <= SM_ROW_PACK_MAX_T ? 1 : SM_ROW_PACK_BIG; else return 1;
}

template <class P, class Ex>
SM_HD void k_f1(Ex& ex, const F1Params& p) {
    typename Ex::template State<FftState> st;
    ex.init(st);
    float* lds = ex.lds() + LDS_SCRATCH_FLOATS;
    const FftPlanDev& pl = p.plan;
    const int T = plan_T<P>(pl), C = plan_N<P>(pl);
    const int LF = plan_lds<P>(pl);
    // the work-groups of an interleaved row group fill the same 128-byte lines: they sit
    // on one XCD, adjacent in dispatch order, so the partial-line writes merge in that L2
    int sig_;
    const int bid = f1_pick_signal(p.sigs, ex.bid(), (p.ilv > p.nb) ? p.ilv / p.nb : 1, sig_);
    const int pbid = p.sigs.n > 1 ? bid : ex.bid();
    SigDesc sgA = p.a, sgB = p.b;
    cf4* t1_ = p.t1;
    double* partials_ = p.partials;
    if (p.sigs.n > 1) {
        sgA.x = p.sigs.x[sig_]; sgA.base = p.sigs.base[sig_];
        sgB.x = (const char*)sgA.x + p.sigs.b_off; sgB.base = sgA.base ? (const char*)sgA.base + p.sigs.b_off : nullptr;
        t1_ = p.sigs.t1[sig_]; partials_ = p.sigs.partials[sig_];
    }
    // static plans are launched only when the 16-byte vector path applies (host checks),
    // so the element-wise path is not even compiled into them
    const bool vec = P::is_static ? true : (p.vec != 0);
    // norm_mode = reference_cpu, fused summaries: which (signal, matrix row) region g holds in component comp
    auto fuse_where = [&](int comp) {
        return [&, comp](int g, int& sig, long long& mrow) {
            const int row = bid * p.nb + g;
            if (row >= p.R) return false;
            if (p.fuse_rowpair) { sig = sig_; mrow = 2LL * row + comp; }
            else { sig = comp; mrow = row; if (comp == 1 && !sgB.x) return false; }
            return true;
        };
    };
    if constexpr (P::is_static) {
        if constexpr (aten_fusable(P::N, P::T)) {
            if (p.fuse.prefix) ex.each(st, [&](int tid, FftState& s) {
                s.fep[0] = aten_fused_ep<P::N, P::T>(p.fuse, tid, fuse_where(0));
                s.fep[1] = aten_fused_ep<P::N, P::T>(p.fuse, tid, fuse_where(1));
            });
        }
    }

    ex.each(st, [&](int tid, FftState& s) {
        const int g = tid / T, t = tid % T;
        const int row = bid * p.nb + g;
        const bool valid = row < p.R;
        double sa = 0.0, sb = 0.0;
        if (vec && sgA.dtype != DT_F32 && sgB.dtype != DT_F32) {
            // 16-bit inputs (the merge itself): all 16-byte loads of the row are issued back
            // to back - no load sits inside a divergent branch, out-of-range ones are
            // clamped to element 0 and masked afterwards - and only then decoded
            constexpr int NQ = EMAX / 8;
            u32x4 ra[NQ], rab[NQ], rb[NQ], rbb[NQ];
            bool ok[NQ];
            size_t off[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int n0 = 8 * (t + q * T);
                ok[q] = valid && n0 < C;
                off[q] = ok[q] ? ((size_t)row * p.row_stride + n0) / 8 : 0;
            }
            // an absent operand reads a's values instead (always there) and is masked out below
            const bool has_ab = sgA.base != nullptr, has_b = sgB.x != nullptr, has_bb = has_b && sgB.base != nullptr;
            const u32x4* pa = (const u32x4*)sgA.x;
            const u32x4* pab = has_ab ? (const u32x4*)sgA.base : pa;
            const u32x4* pb = has_b ? (const u32x4*)sgB.x : pa;
            const u32x4* pbb = has_bb ? (const u32x4*)sgB.base : pa;
            // the 128-VGPR variants (512/1024-thread work-groups, run-time plans) keep half of the
            // row's loads in flight at a time: all sixteen at once made them spill
            constexpr int QB = f1_full_batch<P>() ? NQ : NQ / 2;
            bool all_ok = true;
#pragma unroll
            for (int q = 0; q < NQ; ++q) all_ok = all_ok && ok[q];
            static_for<0, NQ / QB>([&](auto h_c) {
            constexpr int Q0 = decltype(h_c)::value * QB, Q1 = Q0 + QB;
#pragma unroll
            for (int q = Q0; q < Q1; ++q) ra[q] = pa[off[q]];
#pragma unroll
            for (int q = Q0; q < Q1; ++q) rab[q] = pab[off[q]];
#pragma unroll
            for (int q = Q0; q < Q1; ++q) rb[q] = pb[off[q]];
#pragma unroll
            for (int q = Q0; q < Q1; ++q) rbb[q] = pbb[off[q]];
            // MASKED: some loads of this thread were clamped (ragged row tail, padded grid)
            auto decode = [&](auto masked_c) {
                constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
                for (int q = Q0; q < Q1; ++q) {
                    float va[8], vb[8], ba[8], bb[8];
                    decode16x8(ra[q], sgA.dtype, va);
                    decode16x8(rab[q], sgA.dtype, ba);
                    decode16x8(rb[q], sgB.dtype, vb);
                    decode16x8(rbb[q], sgB.dtype, bb);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        if (!has_ab) ba[c] = 0.f;
                        if (!has_b) vb[c] = 0.f;
                        if (!has_bb) bb[c] = 0.f;
                    }
                    float pa = 0.f, pb = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float xa = (va[c] - ba[c]) * sgA.prescale;
                        float xb = (vb[c] - bb[c]) * sgB.prescale;
                        if (MASKED && !ok[q]) { xa = 0.f; xb = 0.f; }      // never NaN * 0 from a clamped load
                        s.xr[q * 8 + c] = xa; s.xi[q * 8 + c] = xb;
                        pa += xa * xa; pb += xb * xb;
                    }
                    sa += pa; sb += pb;
                }
            };
            if (all_ok) decode(std::false_type{});
            else decode(std::true_type{});
            });

        } else if (vec) {
            // any other dtype mix (later tournament rounds: an fp32 intermediate against a raw
            // bf16 delta): one signal at a time, its loads issued together, decoded afterwards
            constexpr int NQ = EMAX / 8;
            bool ok[NQ];
            size_t off[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int n0 = 8 * (t + q * T);
                ok[q] = valid && n0 < C;
                off[q] = ok[q] ? (size_t)row * p.row_stride + n0 : 0;
            }
            auto load_signal = [&](const SigDesc& sg, float* dst, double& ss) {
                if (!sg.x) {
#pragma unroll
                    for (int i = 0; i < 8 * NQ; ++i) dst[i] = 0.f;
                    return;
                }
                const bool has_base = sg.base != nullptr;
                float v[NQ][8], bv[NQ][8];
                if (sg.dtype == DT_F32) {
                    const cf4* px = (const cf4*)sg.x;
                    const cf4* pb = has_base ? (const cf4*)sg.base : px;
                    cf4 r[NQ][2], rb[NQ][2];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { r[q][0] = px[off[q] / 4]; r[q][1] = px[off[q] / 4 + 1]; }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { rb[q][0] = pb[off[q] / 4]; rb[q][1] = pb[off[q] / 4 + 1]; }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const cf4 x0 = r[q][0], x1 = r[q][1], b0 = rb[q][0], b1 = rb[q][1];
                        const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        const float bs[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                        for (int c = 0; c < 8; ++c) { v[q][c] = xs[c]; bv[q][c] = bs[c]; }
                    }
                } else {
                    const u32x4* px = (const u32x4*)sg.x;
                    const u32x4* pb = has_base ? (const u32x4*)sg.base : px;
                    u32x4 r[NQ], rb[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) r[q] = px[off[q] / 8];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) rb[q] = pb[off[q] / 8];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) { decode16x8(r[q], sg.dtype, v[q]); decode16x8(rb[q], sg.dtype, bv[q]); }
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float ps = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        float x = (v[q][c] - (has_base ? bv[q][c] : 0.f)) * sg.prescale;
                        if (!ok[q]) x = 0.f;
                        dst[q * 8 + c] = x;
                        ps += x * x;
                    }
                    ss += ps;
                }
            };
            load_signal(sgA, s.xr, sa);
            load_signal(sgB, s.xi, sb);
        } else {
#pragma unroll
            for (int q = 0; q < EMAX; ++q) {
                const int n = t + q * T;
                float va = 0.f, vb = 0.f;
                if (valid && n < C) {
                    const size_t off = (size_t)row * p.row_stride + n;
                    va = load_sig1(sgA, off);
                    vb = load_sig1(sgB, off);
                }
                s.xr[q] = va; s.xi[q] = vb;
                sa += (double)va * va; sb += (double)vb * vb;
            }
        }
        s.red[0] = sa; s.red[1] = sb;
    });
    ex.template block_sum<2>(st, [&](const double* tot) {
        partials_[2 * (size_t)pbid] = tot[0];
        partials_[2 * (size_t)pbid + 1] = tot[1];
    });

    wg_fft<P, f1_pack<P>(), 1>(ex, st, pl, lds,
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;          // natural scatter
            const int g = tid / T, t = tid % T;
            float* l = lds + g * LF;
            const float* x = comp_of<comp>(s);
            if (vec) {
#pragma unroll
                for (int q = 0; q < EMAX / 8; ++q) {
                    const int n0 = 8 * (t + q * T);
                    if (n0 < C) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) l[lpad(n0 + c)] = x[q * 8 + c];
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < EMAX; ++q) {
                    const int n = t + q * T;
                    if (n < C) l[lpad(n)] = x[q];
                }
            }
        },
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;          // final gather: pairs (k, C-k)
            int tid_ = tid;
            if constexpr (f1_opaque<P>()) SM_OPAQUE(tid_);         // 128-VGPR plans: addresses recomputed here, not carried from the top
            const int g = tid_ / T, t = tid_ % T;
            const float* l = lds + g * LF;
            float* o = comp_of<comp>(s);
#pragma unroll
            for (int u = 0; u < EMAX / 2 + 1; ++u) {
                const int k = t + u * T;
                if (k < p.Cb) {
                    const int k2 = (k == 0) ? 0 : C - k;
                    const float v1 = l[lpad(k)], v2 = l[lpad(k2)];
                    if (comp == 0) { o[2 * u] = 0.5f * (v1 + v2); o[2 * u + 1] = 0.5f * (v2 - v1); }   // A.re, B.im
                    else           { o[2 * u] = 0.5f * (v1 - v2); o[2 * u + 1] = 0.5f * (v1 + v2); }   // A.im, B.re
                    // an absent second signal is EXACTLY zero (its sign class matters: sign(0) = 0);
                    // the split above would leave rounding noise of random sign there
                    if (!sgB.x) o[2 * u + 1] = 0.f;
                }
            }
        },
        [&](auto comp_c) {
            // norm_mode = reference_cpu: the torch.norm summaries of the deltas that now sit in LDS in natural order
            constexpr int comp = decltype(comp_c)::value;
            if constexpr (P::is_static) {
                if constexpr (aten_fusable(P::N, P::T)) {
                    if (p.fuse.prefix)
                        aten_fused_rows<P::N, P::T>(ex, st, p.fuse, lds, LF, fuse_where(comp), [](FftState& s) { return s.fep[comp]; });
                }
            }
        });

    ex.each(st, [&](int tid, FftState& s) {
        int tid_ = tid;
        if constexpr (f1_opaque<P>()) SM_OPAQUE(tid_);
        const int g = tid_ / T, t = tid_ % T;
        const int row = bid * p.nb + g;
        if (row >= p.R) return;
        cf4* dst = t1_ + (size_t)(row / p.ilv) * p.pitch4 * p.ilv + (row % p.ilv);
#pragma unroll
        for (int u = 0; u < EMAX / 2 + 1; ++u) {
            const int k = t + u * T;
            if (k < p.Cb) {
                cf4 v;
                v.x = s.xr[2 * u]; v.y = s.xi[2 * u]; v.z = s.xi[2 * u + 1]; v.w = s.xr[2 * u + 1];
                dst[(size_t)k * p.ilv] = v;
            }
        }
    });
}

// =====================================================================
// F1Q: forward row pass with the first radix-4 step of the COLUMN transform folded in.
//
// A long column (28672 = 7 * 2^12 rows of a Llama-3-70B MLP tensor) does not fit the engine's
// efficient regime: one 1024-thread work-group per signal, lock-step, 8-byte strided reads -
// 1.8 TB/s.  Split R = 4 * R2 (decimation in time, n = n1 * R2 + n2, k = k1 + 4 * k2):
//     X[k1 + 4 k2] = sum_{n2} W_R2^{n2 k2} * ( W_R^{n2 k1} * sum_{n1} x[n1 R2 + n2] W_4^{n1 k1} )
// The inner radix-4 butterfly over rows n2, n2 + R2, n2 + 2 R2, n2 + 3 R2 and its twiddle are
// element-wise in the bin index, so the work-group that has just transformed those four ROWS
// does them (three twiddles per work-group), and the column pass runs plain R2-point
// transforms (7168 points: 256 threads, two signals per work-group) on 4 * Cb "virtual"
// columns (k1, bin).  The spectrum planes then hold each bin column in the order
// [k1][k2] instead of k - every consumer between the transforms is order-blind inside a
// column, and the inverse column pass reads that order back (k_i1, FOLD).
// T1 here is slab-major: element (n2, k1, bin) at k1 * slab_elems + ((n2 / ilv) * pitch4 + bin) * ilv + n2 % ilv
// (the column pass then strides by one real row pitch inside a 1/4-size slab, as the plain layout does).
// =====================================================================
#ifndef SM_F1Q_QB
#define SM_F1Q_QB 1          // 16-byte loads per operand in flight per batch (register budget: 128)
#endif
#ifndef SM_F1Q_PREFETCH_B
#define SM_F1Q_PREFETCH_B 1     // issue the second operand's loads before the first exchange (measured: -2..5 %)
#endif
#ifndef SM_F1Q_PACK
#define SM_F1Q_PACK 1
#endif
template <class P> constexpr bool f1q_eligible() {
    if constexpr (P::is_static) return 4 * P::T <= 1024 && (P::N / P::T) % 2 == 0 && P::N % 8 == 0; else return false;
}

struct F1QState {
    float xr[EREG];
    float xi[EREG];
    double red[4];
    u32x4 rb[EMAX / 8], rbb[EMAX / 8];      // the second operand's raw 16-bit loads, in flight during the first exchange
};

template <class P, class Ex>
SM_HD void k_f1q(Ex& ex, const F1Params& p) {
    if constexpr (!f1q_eligible<P>()) { return; } else {
    typename Ex::template State<F1QState> st;
    ex.init(st);
    float* lds = ex.lds() + LDS_SCRATCH_FLOATS;
    const FftPlanDev& pl = p.plan;
    constexpr int T = P::T, C = P::N, LF = P::lds_floats;
    constexpr int E2 = C / T / 2;                  // bins k = t + u*T, u < E2, plus the Nyquist bin (t = 0, u = E2)
    constexpr int NJ = (E2 + 3) / 4;               // main bins per thread: u = 4*jj + g
    constexpr int NQ = EMAX / 8;
    static_assert(NJ * 8 + 4 <= EREG, "register slots");
    int sig_;
    const int bid = f1_pick_signal(p.sigs, ex.bid(), p.ilv > 1 ? p.ilv : 1, sig_);
    const int pbid = p.sigs.n > 1 ? bid : ex.bid();
    SigDesc sa = p.a, sb = p.b;
    cf4* t1_ = p.t1;
    double* partials_ = p.partials;
    if (p.sigs.n > 1) {
        sa.x = p.sigs.x[sig_]; sa.base = p.sigs.base[sig_];
        sb.x = (const char*)sa.x + p.sigs.b_off; sb.base = sa.base ? (const char*)sa.base + p.sigs.b_off : nullptr;
        t1_ = p.sigs.t1[sig_]; partials_ = p.sigs.partials[sig_];
    }
    const int unit = bid;                           // n2 (or the row-pair index m2)
    const bool live = unit < p.R2;
    const bool bits16 = sa.dtype != DT_F32 && sb.dtype != DT_F32;
    const bool has_ab = sa.base != nullptr, has_b = sb.x != nullptr, has_bb = has_b && sb.base != nullptr;
    // twiddles W_R^{n2 k1}: slot A is column element n2A, slot B n2B (they differ in row-pair mode)
    const int n2A = p.rowpair ? 2 * unit : unit, n2B = p.rowpair ? 2 * unit + 1 : unit;
    cf4* const rowp = t1_ + (size_t)(unit / p.ilv) * p.pitch4 * p.ilv + (unit % p.ilv);
    const size_t slabstride = p.slab_elems;                      // slab-major: [k1][unit][bin]

    ex.each(st, [&](int, F1QState& s) { s.red[0] = 0.0; s.red[1] = 0.0; });
    // (no fused torch.norm summaries here, unlike k_f1: this kernel is one 1024-thread work-group per CU whose transform
    //  phase is fully exposed - measured on MI355X the summaries cost it +514 us per 28672 x 8192 K = 3 launch, more than
    //  the separate summary pass they would replace (454 us); the folded shapes keep k_aten_part)

    // Register budget: a 1024-thread work-group has 128 VGPRs per thread.  The operands are
    // therefore loaded one at a time, straight into the first LDS scatter (a in the pass of the
    // real component, b in the pass of the imaginary one; b's raw loads are issued before a's
    // exchange and land during it), and the last gather of the imaginary component does the
    // cross-row butterfly and the stores bin by bin instead of parking 36 more values.
    wg_fft<P, SM_F1Q_PACK, 1>(ex, st, pl, lds,
        [&](int tid, F1QState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;          // natural scatter = operand load
            int tid_ = tid;
            SM_OPAQUE(tid_);
            const int g = tid_ / T, t = tid_ % T;
            float* l = lds + g * LF;
            const size_t rowoff = (size_t)(unit + g * p.R2) * p.row_stride;
            auto okq = [&](int q) { return live && 8 * (t + q * T) < C; };
            auto offq = [&](int q) { return okq(q) ? rowoff + 8 * (t + q * T) : (size_t)0; };
            double ss = 0.0;
            if (bits16) {
                const u32x4* pa = (const u32x4*)sa.x;
                if (comp == 0) {
                    const u32x4* pab = has_ab ? (const u32x4*)sa.base : pa;
                    u32x4 ra[NQ], rab[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) ra[q] = pa[offq(q) / 8];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) rab[q] = pab[offq(q) / 8];
#if SM_F1Q_PREFETCH_B
                    const u32x4* pb = has_b ? (const u32x4*)sb.x : pa;
                    const u32x4* pbb = has_bb ? (const u32x4*)sb.base : pa;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) s.rb[q] = pb[offq(q) / 8];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) s.rbb[q] = pbb[offq(q) / 8];
#endif
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        float va[8], ba[8];
                        decode16x8(ra[q], sa.dtype, va);
                        decode16x8(rab[q], sa.dtype, ba);
                        const int n0 = 8 * (t + q * T);
                        float qa = 0.f;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            float xa = (va[c] - (has_ab ? ba[c] : 0.f)) * sa.prescale;
                            if (!okq(q)) xa = 0.f;
                            if (n0 < C) l[lpad(n0 + c)] = xa;
                            qa += xa * xa;
                        }
                        ss += qa;
                    }
                } else {
#if !SM_F1Q_PREFETCH_B
                    const u32x4* pb = has_b ? (const u32x4*)sb.x : pa;
                    const u32x4* pbb = has_bb ? (const u32x4*)sb.base : pa;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) s.rb[q] = pb[offq(q) / 8];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) s.rbb[q] = pbb[offq(q) / 8];
#endif
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        float vb[8], bb[8];
                        decode16x8(s.rb[q], sb.dtype, vb);
                        decode16x8(s.rbb[q], sb.dtype, bb);
                        const int n0 = 8 * (t + q * T);
                        float qb = 0.f;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            float xb = has_b ? (vb[c] - (has_bb ? bb[c] : 0.f)) * sb.prescale : 0.f;
                            if (!okq(q)) xb = 0.f;
                            if (n0 < C) l[lpad(n0 + c)] = xb;
                            qb += xb * xb;
                        }
                        ss += qb;
                    }
                }
            } else {
                const SigDesc& sg = comp == 0 ? sa : sb;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float v[8];
                    if (sg.x) load_sig8(sg, offq(q), v);
                    else { for (int c = 0; c < 8; ++c) v[c] = 0.f; }
                    const int n0 = 8 * (t + q * T);
                    float ps = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float x = okq(q) ? v[c] : 0.f;
                        if (n0 < C) l[lpad(n0 + c)] = x;
                        ps += x * x;
                    }
                    ss += ps;
                }
            }
            s.red[comp] += ss;
        },
        [&](int tid, F1QState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            // every thread takes a quarter of the bins, of ALL FOUR rows.  The real component's
            // pass parks (A.re, B.im) in xr[(jj*4 + g')*2 + {0,1}]; the imaginary component's
            // pass gets (A.im, B.re), does the radix-4 butterfly over the rows and stores.
            int tid_ = tid;
            SM_OPAQUE(tid_);                           // addresses below are recomputed here, not carried from the top
            const int g = tid_ / T, t = tid_ % T;
            auto emit = [&](int k, const float* are, const float* aim, const float* bre, const float* bim) {
                SM_SCHED_FENCE();                      // one bin's butterflies at a time (register pressure)
                float ar[4], ai[4], br[4], bi[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { ar[q] = are[q]; ai[q] = aim[q]; br[q] = has_b ? bre[q] : 0.f; bi[q] = has_b ? bim[q] : 0.f; }
                Dft<4>::run(ar, ai);
                Dft<4>::run(br, bi);
#pragma unroll
                for (int k1 = 0; k1 < 4; ++k1) {
                    if (k1) {
                        // work-group-uniform addresses: the twiddles come through the scalar cache, not VGPRs
                        const cf2 wa = p.twR[(size_t)n2A * k1], wb = p.twR[(size_t)n2B * k1];
                        cmul(ar[k1], ai[k1], wa.x, wa.y); cmul(br[k1], bi[k1], wb.x, wb.y);
                    }
                    cf4 v = {ar[k1], ai[k1], br[k1], bi[k1]};
                    if (live) rowp[(size_t)k * p.ilv + k1 * slabstride] = v;
                }
                SM_SCHED_FENCE();
            };
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int u = 4 * jj + g;
                if (u < E2) {
                    const int k = t + u * T;
                    const int k2 = (k == 0) ? 0 : C - k;
                    float aim[4], bre[4];
#pragma unroll
                    for (int gp = 0; gp < 4; ++gp) {
                        const float* l = lds + gp * LF;
                        const float v1 = l[lpad(k)], v2 = l[lpad(k2)];
                        const int sl = (jj * 4 + gp) * 2;
                        if (comp == 0) { s.xr[sl] = 0.5f * (v1 + v2); s.xr[sl + 1] = 0.5f * (v2 - v1); }   // A.re, B.im
                        else           { aim[gp] = 0.5f * (v1 - v2); bre[gp] = 0.5f * (v1 + v2); }           // A.im, B.re
                    }
                    if (comp == 1) {
                        float are[4], bim[4];
#pragma unroll
                        for (int gp = 0; gp < 4; ++gp) { are[gp] = s.xr[(jj * 4 + gp) * 2]; bim[gp] = s.xr[(jj * 4 + gp) * 2 + 1]; }
                        emit(k, are, aim, bre, bim);
                    }
                }
            }
            if (tid == 0) {                                  // the Nyquist bin k = C/2: its own twin
                if (comp == 0) {
#pragma unroll
                    for (int gp = 0; gp < 4; ++gp) s.xr[NJ * 8 + gp] = lds[gp * LF + lpad(C / 2)];
                } else {
                    float are[4], aim[4] = {0.f, 0.f, 0.f, 0.f}, bre[4], bim[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int gp = 0; gp < 4; ++gp) { are[gp] = s.xr[NJ * 8 + gp]; bre[gp] = lds[gp * LF + lpad(C / 2)]; }
                    emit(C / 2, are, aim, bre, bim);
                }
            }
        });

    ex.template block_sum<2>(st, [&](const double* tot) {
        partials_[2 * (size_t)pbid] = tot[0];
        partials_[2 * (size_t)pbid + 1] = tot[1];
    });
    }
}

// =====================================================================
// F2: forward column pass (one half-spectrum bin column of A and of B)
// =====================================================================
// One launch over the slices of a rank > 2 tensor or the row blocks of a split column length: the
// grid is `wgs` work-groups per slice (0: a single slice); the slices' T1 / G and planes lie
// t1_stride (float4, or float2 for G) and plane_stride (floats) apart.
struct SliceGrid { int wgs = 0; size_t t1_stride = 0, plane_stride = 0; };
template <class Ex> SM_HD int slice_of(Ex& ex, const SliceGrid& sl, int& bid) {
    bid = ex.bid();
    if (sl.wgs <= 0) return 0;
    const int s = bid / sl.wgs;
    bid -= s * sl.wgs;
    return s;
}

struct F2Params {
    FftPlanDev plan;       // N = R
    const cf4* t1;
    int pitch4;
    int ilv;               // see F1Params
    int R, C, Cb;
    int nsig;              // 2: A and B in one work-group; 1: one signal per work-group
    int swap;              // 1: slot B plays role "a" (the larger-norm input)
    float scale[2];        // per slot: 1/norm (or the arithmetic branch's scale)
    float* reA; float* imA; float* reB;     // planes [Cb][R]; role b writes only reB
    unsigned long long* hist;               // level-1 histogram (HIST1_BINS) or null
    // FOLD variants (see k_f1q): R is the sub-length R2, Cb counts the 4 * slab virtual columns
    int slab;              // virtual columns per k1 (the real T1 pitch in float4)
    int Cb_real;           // bins per slab that exist (C/2 + 1)
    int Rfull;             // 4 * R: a bin column of the planes
    size_t slab_elems;     // float4 per k1 slab of T1 (slab-major: the row stride stays one real pitch)
    SliceGrid sl;          // several slices (rank > 2 tensor / row blocks of a split column length) in one launch
};
// virtual column v of a folded column pass -> plane offset of its first element, real bin (or -1)
SM_HD int fold_bin(int v, int slab, int Cb_real, int R2, int Rfull, size_t& off) {
    const int k1 = v / slab, k = v % slab;
    off = (size_t)k * Rfull + (size_t)k1 * R2;
    return k < Cb_real ? k : -1;
}

// BINS adjacent bin columns per work-group (2*BINS transforms, 2*BINS*T threads): the
// row segment a work-group reads is BINS*16 bytes wide, so short columns get full
// 128-byte segments (BINS = 8) and 8192-long ones 32 bytes.
// Layout and grouping rules shared with the host (sm_pipeline.hpp: t1_interleave(), run_f2()).
// For a static plan the column length is known, so both are compile-time constants and the
// kernel contains one variant only (the union of all variants cost ~10 % through register
// allocation alone).
#ifndef SM_T1_ILV
#define SM_T1_ILV 2
#endif
#ifndef SM_T1_ILV_MIN_ROWS
#define SM_T1_ILV_MIN_ROWS 8192
#endif
#ifndef SM_F2_MAX_THREADS
#define SM_F2_MAX_THREADS 512
#endif
SM_HD constexpr int t1_interleave_rows(int R) { return R >= SM_T1_ILV_MIN_ROWS ? SM_T1_ILV : 1; }
SM_HD constexpr int f2_nsig_for(int T) { return (2 * T <= SM_F2_MAX_THREADS) ? 2 : 1; }

constexpr int FOLD_ILV = 2;      // rows n2, n2 + 1 interleaved in T1 on the folded path: 32-byte pieces
template <class P> constexpr bool fold_col_plan() {          // lengths used as R2 = R / 4
    if constexpr (P::is_static) return f2_nsig_for(P::T) == 2 && (P::N == 2048 || P::N == 3584 || P::N == 4096 || P::N == 7168);
    else return false;
}
template <class P, int BINS, bool FOLD = false, class Ex>
SM_HD void k_f2(Ex& ex, const F2Params& p) {
    if constexpr (FOLD && !fold_col_plan<P>()) { return; } else {
    typename Ex::template State<FftState> st;
    ex.init(st);
    float* lds = ex.lds() + LDS_SCRATCH_FLOATS;
    const FftPlanDev& pl = p.plan;
    const int T = plan_T<P>(pl), R = plan_N<P>(pl);
    const int LF = plan_lds<P>(pl);
    const int ng = P::is_static ? f2_nsig_for(T) : p.nsig;              // compile-time for static plans
    const int ilv = FOLD ? FOLD_ILV : (P::is_static ? t1_interleave_rows(R) : p.ilv);
    int bid;
    const int slice = slice_of(ex, p.sl, bid);
    const cf4* const t1 = p.t1 + slice * p.sl.t1_stride;
    const size_t pl_off = slice * p.sl.plane_stride;
    // 8/BINS work-groups share a 128-byte line of T1 (16 bytes per bin)
    // (one signal per work-group, T = 1024: the 16 work-groups of a line - 8 bins x 2 signals)
    const int lbid = (ng == 2) ? xcd_remap(bid, 8 / BINS) : xcd_remap(bid, 16);
    const int kbase = (ng == 2) ? lbid * BINS : lbid / 2;
    const int slot0 = (ng == 2) ? 0 : lbid % 2;
    if (kbase >= p.Cb) return;
    const int ngroups = (ng == 2) ? 2 * BINS : 1;
    const int nthreads = ngroups * T;
    uint32_t* lhist = (uint32_t*)(lds + ngroups * LF);

    ex.each(st, [&](int tid, FftState& s) {
        if (ng == 2) {
            const int b = tid % BINS, lane = tid / BINS;
            const int k2 = kbase + b;
            // slot q holds row f2_row(lane, q): ILV consecutive rows of a thread are one
            // contiguous ILV*16-byte piece of T1
            auto load_rows = [&](auto ilv_c) {
                constexpr int ILV = decltype(ilv_c)::value;
#pragma unroll
                for (int q = 0; q < EMAX / 2; ++q) {
                    const int m = lane + (q / ILV) * 2 * T;               // row group
                    const int n = m * ILV + q % ILV;
                    // (clamped address, masked value: see k_f2s)
                    const bool live = n < R && k2 < p.Cb;
                    const int kc = k2 < p.Cb ? k2 : p.Cb - 1;
                    const int mcl = n < R ? m : (R - 1 - q % ILV) / ILV;
                    cf4 v;
                    if constexpr (FOLD) {
                        // slab-major T1: virtual column k2 = (k1, bin) lives in slab k1, row pitch = slab
                        v = t1[(size_t)(kc / p.slab) * p.slab_elems + ((size_t)mcl * p.slab + (kc % p.slab)) * ILV + q % ILV];
                    } else {
                        v = t1[((size_t)mcl * p.pitch4 + kc) * ILV + q % ILV];
                    }
                    s.xr[2 * q] = live ? v.x : 0.f; s.xi[2 * q] = live ? v.y : 0.f; s.xr[2 * q + 1] = live ? v.z : 0.f; s.xi[2 * q + 1] = live ? v.w : 0.f;
                }
            };
            if (ilv == 2) load_rows(std::integral_constant<int, 2>{});
            else load_rows(std::integral_constant<int, 1>{});
        } else {
#pragma unroll
            for (int q = 0; q < EMAX; ++q) {
                const int n = tid + q * T;
                float re = 0.f, im = 0.f;
                if (n < R) {
                    const cf2* src = (const cf2*)(t1 + ((size_t)(n / ilv) * p.pitch4 + kbase) * ilv + n % ilv) + slot0;
                    re = src->x; im = src->y;
                }
                s.xr[q] = re; s.xi[q] = im;
            }
        }
        if (p.hist) for (int h = tid; h < HIST1_BINS; h += nthreads) lhist[h] = 0;
    });

#ifndef SM_F2_PACK
#define SM_F2_PACK 1
#endif
    wg_fft<P, SM_F2_PACK, 2>(ex, st, pl, lds,
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            const float* x = comp_of<comp>(s);
            if (ng == 2) {
                const int b = tid % BINS, lane = tid / BINS;
                float* la = lds + (2 * b) * LF;
                auto scatter_rows = [&](auto ilv_c) {
                    constexpr int ILV = decltype(ilv_c)::value;
#pragma unroll
                    for (int q = 0; q < EMAX / 2; ++q) {
                        const int n = (lane + (q / ILV) * 2 * T) * ILV + q % ILV;
                        if (n < R) { la[lpad(n)] = x[2 * q]; la[LF + lpad(n)] = x[2 * q + 1]; }
                    }
                };
                if (ilv == 2) scatter_rows(std::integral_constant<int, 2>{});
                else scatter_rows(std::integral_constant<int, 1>{});
            } else {
#pragma unroll
                for (int q = 0; q < EMAX; ++q) {
                    const int n = tid + q * T;
                    if (n < R) lds[lpad(n)] = x[q];
                }
            }
        },
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            const int g = tid / T, t = tid % T;
            const float* l = lds + g * LF;
            float* o = comp_of<comp>(s);
#pragma unroll
            for (int u = 0; u < EMAX / 4; ++u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k1 = 4 * (t + u * T) + c;
                    if (k1 < R) o[4 * u + c] = l[lpad(k1)];
                }
            }
        });

    ex.each(st, [&](int tid, FftState& s) {
        const int g = tid / T, t = tid % T;
        const int slot = (ng == 2) ? (g & 1) : slot0;
        const int k2 = kbase + ((ng == 2) ? (g >> 1) : 0);
        if (k2 >= p.Cb) return;
        const bool role_a = (slot ^ p.swap) == 0;
        const float sc = p.scale[slot];
        size_t poff = (size_t)k2 * R;
        int kreal = k2;
        if constexpr (FOLD) { kreal = fold_bin(k2, p.slab, p.Cb_real, R, p.Rfull, poff); if (kreal < 0) return; }
        const uint32_t w = (uint32_t)bin_weight(kreal, p.C);
        float* dre = (role_a ? p.reA : p.reB) + pl_off + poff;
        float* dim = p.imA + pl_off + poff;
#pragma unroll
        for (int u = 0; u < EMAX / 4; ++u) {
            const int k0 = 4 * (t + u * T);
            if (k0 + 3 < R && (R & 3) == 0) {
                cf4 vr = {s.xr[4 * u] * sc, s.xr[4 * u + 1] * sc, s.xr[4 * u + 2] * sc, s.xr[4 * u + 3] * sc};
                *(cf4*)(dre + k0) = vr;
                if (role_a) {
                    cf4 vi = {s.xi[4 * u] * sc, s.xi[4 * u + 1] * sc, s.xi[4 * u + 2] * sc, s.xi[4 * u + 3] * sc};
                    *(cf4*)(dim + k0) = vi;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (k0 + c < R) {
                        dre[k0 + c] = s.xr[4 * u + c] * sc;
                        if (role_a) dim[k0 + c] = s.xi[4 * u + c] * sc;
                    }
                }
            }
            if (p.hist) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (k0 + c < R) {
                        const uint32_t key = f2u(s.xr[4 * u + c] * sc) & 0x7fffffffu;
                        ex.lds_atomic_add(&lhist[key >> 20], w);
                    }
                }
            }
        }
    });
    if (p.hist) {
        ex.sync();
        ex.each(st, [&](int tid, FftState&) {
            for (int h = tid; h < HIST1_BINS; h += nthreads) {
                const uint32_t v = lhist[h];
                if (v) ex.global_atomic_add(&p.hist[h], (unsigned long long)v);
            }
        });
    }
    }
}

// bins per work-group of the column passes for a plan with T threads per transform
// (measured on MI355X: 1024-thread work-groups lose more than their wider row segments
//  gain - 8192^2: F2 382 -> 415 us, I1 164 -> 203 us - so the target is 512 threads)
#ifndef SM_COL_THREADS
#define SM_COL_THREADS 512
#endif
// (the 256-point row blocks of a split column length are so short that a work-group's fixed
//  costs - its histogram flush above all: the same ~150 hot addresses from every work-group - outweigh
//  its payload: 8 bins there, whole 128-byte lines of T1, half the work-groups)
constexpr int f2_bins_for(int T, int N = 1 << 30) {
    if (T == 64 && N <= 256) return 8;        // (measured: 512-point blocks lose 20 % with 8)
    return (SM_COL_THREADS / (2 * T)) > 1 ? (SM_COL_THREADS / (2 * T) > 8 ? 8 : SM_COL_THREADS / (2 * T)) : 1;
}
// the inverse column pass measured fastest with two bins per work-group at every length
#ifndef SM_I1_THREADS
constexpr int i1_bins_for(int T) { return 2; }     // the host launches KI1x1 instead when 2T is too large
#else
constexpr int i1_bins_for(int T) { return (SM_I1_THREADS / T) > 2 ? ((SM_I1_THREADS / T) > 16 ? 16 : (SM_I1_THREADS / T)) : 2; }
#endif
template <class P> constexpr int f2_bins() { if constexpr (P::is_static) return f2_bins_for(P::T, P::N); else return 1; }
template <class P> constexpr int i1_bins() { if constexpr (P::is_static) return i1_bins_for(P::T); else return 2; }

// ---------------------------------------------------------------------------------
// F2S: forward column pass of ONE signal (tournament rounds >= 2: the other input of the pair is
// an intermediate that stayed in the spectral domain).  The row pass ran in row-pair mode
// (F1Params::row_stride = 2C): T1[m][k] = (row 2m's bin k, row 2m+1's bin k), so one float4 holds
// two CONSECUTIVE elements of a column.  A work-group transforms G adjacent bin columns
// (G = the 2*BINS transforms of the two-signal kernel, same threads / LDS), reading G*16
// contiguous bytes per row pair.  role_a: the signal is the pair's larger-norm input (writes
// Re a, Im a), else it writes Re b only.
// ---------------------------------------------------------------------------------
struct F2SParams {
    FftPlanDev plan;       // N = R (even)
    const cf4* t1;         // [R/2][pitch4]
    int pitch4;
    int R, C, Cb;
    int role_a;
    float scale;
    float* re; float* im;  // destination planes [Cb][R] (im unused when !role_a)
    unsigned long long* hist;
    int slab, Cb_real, Rfull;   // FOLD variant: as F2Params
    size_t slab_elems;
    double* im_partials;        // role a only, or null: [grid] sum w (Im a)^2 of what this work-group stored
                                // (the Parseval norm of the pair's result when it stays spectral)
    SliceGrid sl;
    // two = 1: BOTH inputs of a pair in one launch (one role a, one role b; no slices): the second signal's T1, role,
    // scale and Re plane; the work-groups of the two signals alternate in blocks of XG (pick_signal) - one launch
    // boundary less per pair merge
    int two;
    const cf4* t1_2; int role_a_2; float scale_2; float* re_2;
};
template <class P> constexpr int f2s_groups() {
    if constexpr (P::is_static) return f2_nsig_for(P::T) == 2 ? 2 * f2_bins_for(P::T, P::N) : 1; else return 1;
}

#ifndef SM_F2S_ILV
#define SM_F2S_ILV 1        // 2 = row PAIRS interleaved in the single-signal T1 (64-byte pieces for the column pass): measured on
                            // MI355X it takes 4-12 % off f2s and puts 8-12 % on the row pass (half-line stores) - net loss
#endif
constexpr int F2S_ILV = SM_F2S_ILV;
// DIRECT (round 4): the column pass takes its loads straight into the first radix pass's layout and stores straight out
// of the last one's - two of the transform's four exchanges through LDS (and 8 of its 16 barriers) go away:
//  * first pass, butterfly j = t + m T takes rows j + i N/R0: lanes 2k and 2k + 1 need the SAME row pairs (T1 holds rows
//    2m, 2m + 1 in one float4) - the even lane loads the pairs of even i, the odd lane those of odd i, and they trade
//    halves through a quad-permute DPP move (Ex::lane_pair_trade), no LDS;
//  * last pass, thread t holds outputs t + T u: 64 lanes store 256 contiguous bytes per instruction (dword stores;
//    measured with tools/membench2.hip: as fast as the exchanged 16-byte form at two columns per work-group).
// Columns are dealt by thread group (g = tid / T) instead of by lane (tid % G): the loader must be the owner.
// MEASURED ON MI355X AND LEFT OFF (profiles/r04_ab_f2s_direct.txt): bit-identical results, but the owner-loads mapping
// gives up the 32-byte pieces two adjacent lanes read (f2s 529 -> 583 us at 28672 x 8192, 464 -> 567 at 8192 x 28672)
// and the dword stores lose more than the exchange they save on the 8192-point plan (464 -> 531; 529 -> 516 on the
// folded 7168-point one): the column passes are bound by their access pattern, not by their exchanges.
#ifndef SM_F2S_DIRECT
#define SM_F2S_DIRECT 0     // bit 0: loads into the first pass's layout (DPP trade), bit 1: stores out of the last pass's
#endif
template <class P, int G> constexpr int f2s_direct() {
    if constexpr (!P::is_static) return 0;
    else {
        constexpr int R0 = P::radix(0);
        constexpr bool ok = F2S_ILV == 1 && (G == 1 || G == 2) && P::npass >= 2 && P::T % 2 == 0 && R0 % 2 == 0 &&
                            (P::N / R0) % 2 == 0 && (EMAX / R0) * R0 * P::T >= P::N;
        return ok ? (SM_F2S_DIRECT & 3) : 0;
    }
}
template <class P, int G, bool FOLD = false, class Ex>
SM_HD void k_f2s(Ex& ex, const F2SParams& p) {
    if constexpr (FOLD && !fold_col_plan<P>()) { return; } else {
    typename Ex::template State<FftState> st;
    ex.init(st);
    float* lds = ex.lds() + LDS_SCRATCH_FLOATS;
    const FftPlanDev& pl = p.plan;
    const int T = plan_T<P>(pl), R = plan_N<P>(pl);
    const int LF = plan_lds<P>(pl);
    constexpr int XG = G >= 8 ? 1 : 8 / G;                 // work-groups that share a 128-byte line of T1
    int bid_in_slice;
    const int slice = slice_of(ex, p.sl, bid_in_slice);
    int sig2 = 0;
    const int lbid = p.two ? pick_signal(2, bid_in_slice, XG, sig2) : xcd_remap(bid_in_slice, XG);
    const cf4* const t1 = (sig2 ? p.t1_2 : p.t1) + slice * p.sl.t1_stride;
    const size_t pl_off = slice * p.sl.plane_stride;
    const int role_a_ = sig2 ? p.role_a_2 : p.role_a;
    const float scale_ = sig2 ? p.scale_2 : p.scale;
    float* const re_ = sig2 ? p.re_2 : p.re;
    const int kbase = lbid * G;
    if (kbase >= p.Cb) {                      // padding work-group: its partial must still read zero
        if (p.im_partials) ex.each(st, [&](int tid, FftState&) { if (tid == 0) p.im_partials[ex.bid()] = 0.0; });
        return;
    }
    const int nthreads = G * T;
    const int half = R / 2;
    uint32_t* lhist = (uint32_t*)(lds + G * LF);

    constexpr int DIRECT = f2s_direct<P, G>();
    if constexpr ((DIRECT & 1) != 0) {
        constexpr int R0 = P::radix(0), MB0 = EMAX / R0, nb0 = P::N / R0;
        ex.each(st, [&](int tid, FftState& s) {
            const int g = tid / T, t = tid % T;
            const int k2 = kbase + g;
            const int kc = k2 < p.Cb ? k2 : p.Cb - 1;
            const int par = t & 1;
            // (clamped address, masked value: a load behind a branch is waited for on the spot)
            static_for<0, MB0>([&](auto m_c) {
                constexpr int m = decltype(m_c)::value;
                const bool live = (t + m * T) < nb0 && k2 < p.Cb;
                static_for<0, R0 / 2>([&](auto q_c) {
                    constexpr int q = decltype(q_c)::value;
                    // the row pair of rows j + i nb0, j = t + m T, i = 2 q + par (nb0, T even)
                    const int pm = live ? (t >> 1) + m * (T / 2) + q * nb0 + par * (nb0 / 2) : 0;
                    cf4 v;
#if defined(SM_DIAG_NOMEM)
                    v = cf4{(float)q, (float)t, 1.f, (float)kc};
#else
                    if constexpr (FOLD) v = t1[(size_t)(kc / p.slab) * p.slab_elems + (size_t)pm * p.slab + (kc % p.slab)];
                    else v = t1[(size_t)pm * p.pitch4 + kc];
#endif
                    s.xr[m * R0 + 2 * q] = live ? v.x : 0.f; s.xi[m * R0 + 2 * q] = live ? v.y : 0.f;
                    s.xr[m * R0 + 2 * q + 1] = live ? v.z : 0.f; s.xi[m * R0 + 2 * q + 1] = live ? v.w : 0.f;
                });
            });
            if (p.hist) for (int h = tid; h < HIST1_BINS; h += nthreads) lhist[h] = 0;
        });
        // the even lane holds (own rows of i = 2q | the partner's rows of i = 2q), the odd lane the mirror image
        ex.template lane_pair_trade<MB0 * (R0 / 2) * 2>(st, [](FftState& s, auto k_c) {
            constexpr int k = decltype(k_c)::value;
            constexpr int idx = k / 2, m = idx / (R0 / 2), q = idx % (R0 / 2);
            float* arr = (k % 2) ? s.xi : s.xr;
            return FloatPairRef{arr[m * R0 + 2 * q], arr[m * R0 + 2 * q + 1]};
        });
    } else
    ex.each(st, [&](int tid, FftState& s) {
        const int b = tid % G, lane = tid / G;
        const int k2 = kbase + b;
#pragma unroll
        for (int q = 0; q < EMAX / 2; ++q) {
            const int m = F2S_ILV * (lane + (q / F2S_ILV) * T) + q % F2S_ILV;      // row pair (rows 2m, 2m+1)
            // (the address is clamped, the value masked afterwards: behind a branch every load is waited for on the
            // spot - the 7168-point plan, whose last two slots are empty, ran its 14 strided loads one after the other)
            const bool live = m < half && k2 < p.Cb;
            const int mc = m < half ? m : half - 1, kc = k2 < p.Cb ? k2 : p.Cb - 1;
            cf4 v;
#if defined(SM_DIAG_NOMEM)
            v = cf4{(float)q, (float)lane, 1.f, (float)kc};          // diagnostic build: no global loads
#else
            if constexpr (FOLD) {
                v = t1[(size_t)(kc / p.slab) * p.slab_elems + ((size_t)(mc / F2S_ILV) * p.slab + (kc % p.slab)) * F2S_ILV + mc % F2S_ILV];
            } else {
                v = t1[((size_t)(mc / F2S_ILV) * p.pitch4 + kc) * F2S_ILV + mc % F2S_ILV];
            }
#endif
            s.xr[2 * q] = live ? v.x : 0.f; s.xi[2 * q] = live ? v.y : 0.f; s.xr[2 * q + 1] = live ? v.z : 0.f; s.xi[2 * q + 1] = live ? v.w : 0.f;
        }
        if (p.hist) for (int h = tid; h < HIST1_BINS; h += nthreads) lhist[h] = 0;
    });

#ifndef SM_F2S_PACK
#define SM_F2S_PACK 1
#endif
    wg_fft<P, SM_F2S_PACK, 4, DIRECT>(ex, st, pl, lds,
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            const float* x = comp_of<comp>(s);
            const int b = tid % G, lane = tid / G;
            float* la = lds + b * LF;
#pragma unroll
            for (int q = 0; q < EMAX / 2; ++q) {
                const int m = F2S_ILV * (lane + (q / F2S_ILV) * T) + q % F2S_ILV;
                if (m < half) { la[lpad(2 * m)] = x[2 * q]; la[lpad(2 * m + 1)] = x[2 * q + 1]; }
            }
        },
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            const int g = tid / T, t = tid % T;
            const float* l = lds + g * LF;
            float* o = comp_of<comp>(s);
#pragma unroll
            for (int u = 0; u < EMAX / 4; ++u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k1 = 4 * (t + u * T) + c;
                    if (k1 < R) o[4 * u + c] = l[lpad(k1)];
                }
            }
        });

    ex.each(st, [&](int tid, FftState& s) {
        const int g = tid / T, t = tid % T;
        const int k2 = kbase + g;
        s.red[0] = 0.0;
        if (k2 >= p.Cb) return;
        const float sc = scale_;
        size_t poff = (size_t)k2 * R;
        int kreal = k2;
        if constexpr (FOLD) { kreal = fold_bin(k2, p.slab, p.Cb_real, R, p.Rfull, poff); if (kreal < 0) return; }
        const uint32_t w = (uint32_t)bin_weight(kreal, p.C);
        float* dre = re_ + pl_off + poff;
        float* dim = p.im + pl_off + poff;
        float imsq = 0.f;
        if constexpr ((DIRECT & 2) != 0) {
            // last-pass layout: slot m RL + i is output j + i NsL, j = t + m T: consecutive lanes, consecutive floats
            constexpr int RL = P::radix(P::npass - 1), MBL = EMAX / RL, NsL = P::N / RL;
            static_for<0, MBL>([&](auto m_c) {
                constexpr int m = decltype(m_c)::value;
                const int j = t + m * T;
                if (j < NsL) {
                    static_for<0, RL>([&](auto i_c) {
                        constexpr int i = decltype(i_c)::value, e = m * RL + i;
                        const float vr = s.xr[e] * sc;
#if defined(SM_DIAG_NOMEM)
                        if (vr == 12345.678f)
#endif
                        dre[j + i * NsL] = vr;
                        if (role_a_) {
                            const float vi = s.xi[e] * sc;
#if defined(SM_DIAG_NOMEM)
                            if (vi == 12345.678f)
#endif
                            dim[j + i * NsL] = vi;
                            imsq += vi * vi;
                        }
                        if (p.hist) ex.lds_atomic_add(&lhist[(f2u(vr) & 0x7fffffffu) >> 20], w);
                    });
                }
            });
        } else
#pragma unroll
        for (int u = 0; u < EMAX / 4; ++u) {
            const int k0 = 4 * (t + u * T);
            if (k0 + 3 < R && (R & 3) == 0) {
                cf4 vr = {s.xr[4 * u] * sc, s.xr[4 * u + 1] * sc, s.xr[4 * u + 2] * sc, s.xr[4 * u + 3] * sc};
#if defined(SM_DIAG_NOMEM)
                if (vr.x == 12345.678f)                                  // diagnostic build: no global stores
#endif
                *(cf4*)(dre + k0) = vr;
                if (role_a_) {
                    cf4 vi = {s.xi[4 * u] * sc, s.xi[4 * u + 1] * sc, s.xi[4 * u + 2] * sc, s.xi[4 * u + 3] * sc};
#if defined(SM_DIAG_NOMEM)
                    if (vi.x == 12345.678f)
#endif
                    *(cf4*)(dim + k0) = vi;
                    imsq += vi.x * vi.x + vi.y * vi.y + vi.z * vi.z + vi.w * vi.w;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (k0 + c < R) {
                        dre[k0 + c] = s.xr[4 * u + c] * sc;
                        if (role_a_) { const float vi = s.xi[4 * u + c] * sc; dim[k0 + c] = vi; imsq += vi * vi; }
                    }
                }
            }
            if (p.hist) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (k0 + c < R) {
                        const uint32_t key = f2u(s.xr[4 * u + c] * sc) & 0x7fffffffu;
                        ex.lds_atomic_add(&lhist[key >> 20], w);
                    }
                }
            }
        }
        s.red[0] = (double)imsq * (double)w;
    });
    if (p.hist) {
        ex.sync();
        ex.each(st, [&](int tid, FftState&) {
            for (int h = tid; h < HIST1_BINS; h += nthreads) {
                const uint32_t v = lhist[h];
                if (v) ex.global_atomic_add(&p.hist[h], (unsigned long long)v);
            }
        });
    }
    if (p.im_partials) {
        ex.sync();
        ex.template block_sum<1>(st, [&](const double* tot) { p.im_partials[ex.bid()] = tot[0]; });
    }
    }
}

// R == 1 (1-D tensors): the column transform is the identity, so the column passes are
// plain element-wise kernels (a 2049-work-group launch of the transform kernel for a
// 4096-element layernorm weight cost 230 us)
template <class Ex>
SM_HD void k_f2_r1(Ex& ex, const F2Params& p) {
    typename Ex::template State<EmptyStateF> st;
    ex.init(st);
    uint32_t* lhist = (uint32_t*)(ex.lds() + LDS_SCRATCH_FLOATS);
    const int nt = ex.nthreads();
    if (p.hist) {
        ex.each(st, [&](int tid, EmptyStateF&) { for (int h = tid; h < HIST1_BINS; h += nt) lhist[h] = 0; });
        ex.sync();
    }
    ex.each(st, [&](int tid, EmptyStateF&) {
        for (int k2 = ex.bid() * nt + tid; k2 < p.Cb; k2 += ex.nblocks() * nt) {
            const cf4 v = p.t1[k2];
            const float in[2][2] = {{v.x, v.y}, {v.z, v.w}};
            const uint32_t w = (uint32_t)bin_weight(k2, p.C);
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                const bool role_a = (slot ^ p.swap) == 0;
                const float re = in[slot][0] * p.scale[slot], im = in[slot][1] * p.scale[slot];
                if (role_a) { p.reA[k2] = re; p.imA[k2] = im; } else { p.reB[k2] = re; }
                if (p.hist) ex.lds_atomic_add(&lhist[(f2u(re) & 0x7fffffffu) >> 20], w);
            }
        }
    });
    if (p.hist) {
        ex.sync();
        ex.each(st, [&](int tid, EmptyStateF&) {
            for (int h = tid; h < HIST1_BINS; h += nt) {
                const uint32_t v = lhist[h];
                if (v) ex.global_atomic_add(&p.hist[h], (unsigned long long)v);
            }
        });
    }
}

// =====================================================================
// I1: inverse column pass
// =====================================================================
struct I1Params {
    FftPlanDev plan;       // N = R
    const float* reR;      // Re R plane [Cb][R] (cull applied on read)
    const float* imA;      // Im plane
    const float* cull_thr; // device scalar or null: |re| < *cull_thr -> 0
    float cull_val;        // used when cull_thr is null (0: no cull)
    int R, Cb;
    int s;                 // bin columns per work-group
    cf2* G;                // [R][pitchG]
    int pitchG;
    SliceGrid sl;          // (t1_stride counts float2 of G here)
};

// FOLD: the planes hold each bin column in the folded order [k1][k2] (k = k1 + 4 k2, see k_f1q):
// the loads stay contiguous (16 bytes from slab k1), the natural position is restored in the
// first LDS scatter.
template <class P> constexpr bool fold_full_plan() {
    if constexpr (P::is_static) return P::N == 8192 || P::N == 14336 || P::N == 16384 || P::N == 28672; else return false;
}
template <class P, int S, bool FOLD = false, class Ex>
SM_HD void k_i1(Ex& ex, const I1Params& p) {
    if constexpr (FOLD && !fold_full_plan<P>()) { return; } else {
    typename Ex::template State<FftState> st;
    ex.init(st);
    float* lds = ex.lds() + LDS_SCRATCH_FLOATS;
    const FftPlanDev& pl = p.plan;
    const int T = plan_T<P>(pl), R = plan_N<P>(pl);
    const int LF = plan_lds<P>(pl);
    int bid_in_slice;
    const int slice = slice_of(ex, p.sl, bid_in_slice);
    const size_t pl_off = slice * p.sl.plane_stride;
    cf2* const Gs = p.G + slice * p.sl.t1_stride;
    const int bid = xcd_remap(bid_in_slice, 16 / S);      // 16 bins of 8 B share a 128-B line
    if (bid * S >= p.Cb) return;
    const float thr = p.cull_thr ? *p.cull_thr : p.cull_val;

    ex.each(st, [&](int tid, FftState& s) {
        const int g = tid / T, t = tid % T;
        const int k2 = bid * S + g;
        const bool valid = k2 < p.Cb;
        const float* sre = p.reR + pl_off + (size_t)k2 * R;
        const float* sim = p.imA + pl_off + (size_t)k2 * R;
#pragma unroll
        for (int u = 0; u < EMAX / 4; ++u) {
            const int k0 = 4 * (t + u * T);            // FOLD: four consecutive elements of one slab
            float re[4] = {0.f, 0.f, 0.f, 0.f}, im[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                if (k0 + 3 < R && (R & 3) == 0) {
                    const cf4 a = *(const cf4*)(sre + k0), b = *(const cf4*)(sim + k0);
                    re[0] = a.x; re[1] = a.y; re[2] = a.z; re[3] = a.w;
                    im[0] = b.x; im[1] = b.y; im[2] = b.z; im[3] = b.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (k0 + c < R) { re[c] = sre[k0 + c]; im[c] = sim[k0 + c]; }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (fabsf(re[c]) < thr) re[c] = 0.f;
                // inverse via the swap trick: ifft(x) = swap(fft(swap(x)))
                s.xr[4 * u + c] = im[c];
                s.xi[4 * u + c] = re[c];
            }
        }
    });

#ifndef SM_I1_PACK
#define SM_I1_PACK 0
#endif
    wg_fft<P, SM_I1_PACK, 8>(ex, st, pl, lds,
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            const int g = tid / T, t = tid % T;
            float* l = lds + g * LF;
            const float* x = comp_of<comp>(s);
#pragma unroll
            for (int u = 0; u < EMAX / 4; ++u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k1 = 4 * (t + u * T) + c;
                    if (k1 < R) {
                        if constexpr (FOLD) {
                            const int R2 = R / 4;                      // stored position k1 = slab * R2 + k2
                            l[lpad((k1 / R2) + 4 * (k1 % R2))] = x[4 * u + c];
                        } else {
                            l[lpad(k1)] = x[4 * u + c];
                        }
                    }
                }
            }
        },
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            float* o = comp_of<comp>(s);
            // thread handles the row PAIRS m = tid + j*(S*T) (rows 2m, 2m+1) and needs every
            // group's value: slot (2j + par)*S + g
#pragma unroll
            for (int j = 0; j < EMAX / (2 * S); ++j) {
                const int m = tid + j * S * T;
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const int r = 2 * m + par;
                    if (r < R) {
#pragma unroll
                        for (int g = 0; g < S; ++g) o[(2 * j + par) * S + g] = lds[g * LF + lpad(r)];
                    }
                }
            }
        });

    // G is stored in row pairs, [R/2][pitchG][2] float2: the inverse row pass takes rows
    // (2m, 2m+1) as one complex transform, and this pass writes 16 contiguous bytes per
    // (pair, bin) - 32 with two bins - whatever the number of bins per work-group
    ex.each(st, [&](int tid, FftState& s) {
#pragma unroll
        for (int j = 0; j < EMAX / (2 * S); ++j) {
            const int m = tid + j * S * T;
            if (2 * m < R) {
                cf4* dst = (cf4*)Gs + (size_t)m * p.pitchG + (size_t)bid * S;
                const bool odd = 2 * m + 1 < R;
#pragma unroll
                for (int g = 0; g < S; ++g) {
                    if (bid * S + g < p.pitchG) {
                        // swap trick: true (re, im) = (xi, xr)
                        const int e0 = (2 * j) * S + g, e1 = (2 * j + 1) * S + g;
                        cf4 v = {s.xi[e0], s.xr[e0], odd ? s.xi[e1] : 0.f, odd ? s.xr[e1] : 0.f};
                        dst[g] = v;
                    }
                }
            }
        }
    });
    }
}

template <class Ex>
SM_HD void k_i1_r1(Ex& ex, const I1Params& p) {
    typename Ex::template State<EmptyStateF> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const float thr = p.cull_thr ? *p.cull_thr : p.cull_val;
    ex.each(st, [&](int tid, EmptyStateF&) {
        for (int k2 = ex.bid() * nt + tid; k2 < p.Cb; k2 += ex.nblocks() * nt) {
            float re = p.reR[k2];
            if (fabsf(re) < thr) re = 0.f;
            cf4 v = {re, p.imA[k2], 0.f, 0.f};           // row pair (0, -)
            ((cf4*)p.G)[k2] = v;
        }
    });
}

// =====================================================================
// I2: inverse row pass, two rows per transform
// =====================================================================
enum { OUT_BF16 = 0, OUT_F32 = 1 };
struct I2Params {
    FftPlanDev plan;       // N = C
    const cf2* G;
    int pitchG;
    int R, C, Cb;
    int nb;                // row pairs per work-group
    int vec;               // C % 8 == 0 (and aligned out/base)
    float inv_n;           // 1/(R*C): the inverse transform's normalisation
    float post;            // target_norm (1 for the arithmetic branch / raw merges)
    int ifft_policy;       // 1: NaN -> 0 (counted) and Inf flagged right after the inverse transform
    const void* base; int base_dtype;   // add-back tensor or null
    void* out; int out_mode;
    uint32_t* flags;       // [0] NaNs zeroed after ifft, [1] Inf after ifft, [2] NaNs zeroed after add-back, [3] Inf after add-back
    double* norm_partials; // [grid][2] or null: sum of squares of what this work-group stored (next round's ||merged||)
};

SM_HD void i2_finish(const I2Params& p, float v, size_t off, uint32_t& nan1, uint32_t& inf1, uint32_t& nan2, uint32_t& inf2, float& outv) {
    v *= p.inv_n;
    if (p.ifft_policy) {
        if (is_nan(v)) { v = 0.f; nan1++; }
        if (is_inf(v)) inf1 = 1;
    }
    v *= p.post;
    if (p.base) {
        v += load_elem(p.base, p.base_dtype, off);
        if (is_nan(v)) { v = 0.f; nan2++; }
        if (is_inf(v)) inf2 = 1;
    }
    outv = v;
}

template <class P, class Ex>
SM_HD void k_i2(Ex& ex, const I2Params& p) {
    typename Ex::template State<FftState> st;
    ex.init(st);
    float* lds = ex.lds() + LDS_SCRATCH_FLOATS;
    const FftPlanDev& pl = p.plan;
    const int T = plan_T<P>(pl), C = plan_N<P>(pl);
    const int LF = plan_lds<P>(pl);
    const int bid = ex.bid();
    const bool vec = P::is_static ? true : (p.vec != 0);      // see k_f1

    ex.each(st, [&](int tid, FftState& s) {
        const int g = tid / T, t = tid % T;
        const int r0 = 2 * (bid * p.nb + g), r1 = r0 + 1;
        const bool v0 = r0 < p.R, v1 = r1 < p.R;
        // loads are issued in two batches, none inside a divergent branch (out-of-range
        // ones are clamped to a valid address and masked), so each batch is in flight together
        const cf4* GP = (const cf4*)p.G + (size_t)(v0 ? r0 / 2 : 0) * p.pitchG;     // row pairs, see k_i1
        constexpr int NU = EMAX / 2 + 1, HU = (NU + 1) / 2;
        static_for<0, 2>([&](auto half_c) {
            constexpr int u0 = decltype(half_c)::value * HU;
            constexpr int u1 = (u0 + HU < NU) ? u0 + HU : NU;
            cf4 ap[HU];
#pragma unroll
            for (int u = u0; u < u1; ++u) {
                const int k = t + u * T;
                ap[u - u0] = GP[k < p.Cb ? k : 0];
            }
#pragma unroll
            for (int u = u0; u < u1; ++u) {
                const int k = t + u * T;
                if (k < p.Cb) {
                    cf2 g0 = {ap[u - u0].x, ap[u - u0].y}, g1 = {ap[u - u0].z, ap[u - u0].w};
                    if (!v0) { g0.x = 0.f; g0.y = 0.f; }
                    if (!v1) { g1.x = 0.f; g1.y = 0.f; }
                    if (k == 0 || 2 * k == C) { g0.y = 0.f; g1.y = 0.f; }   // c2r: DC / Nyquist are real
                    // Y[k] = G0[k] + i G1[k];  Y[C-k] = conj(G0[k]) + i conj(G1[k])
                    const float ykr = g0.x - g1.y, yki = g0.y + g1.x;
                    const float ymr = g0.x + g1.y, ymi = g1.x - g0.y;
                    // swap trick on input: feed (im, re)
                    s.xr[2 * u] = yki; s.xi[2 * u] = ykr;
                    s.xr[2 * u + 1] = ymi; s.xi[2 * u + 1] = ymr;
                }
            }
        });
    });

    wg_fft<P, f1_pack<P>(), 16>(ex, st, pl, lds,
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            const int g = tid / T, t = tid % T;
            float* l = lds + g * LF;
            const float* x = comp_of<comp>(s);
#pragma unroll
            for (int u = 0; u < EMAX / 2 + 1; ++u) {
                const int k = t + u * T;
                if (k < p.Cb) {
                    l[lpad(k)] = x[2 * u];
                    if (k != 0 && 2 * k != C) l[lpad(C - k)] = x[2 * u + 1];
                }
            }
        },
        [&](int tid, FftState& s, auto comp_c) {
            constexpr int comp = decltype(comp_c)::value;
            const int g = tid / T, t = tid % T;
            const float* l = lds + g * LF;
            float* o = comp_of<comp>(s);
            if (vec) {
#pragma unroll
                for (int q = 0; q < EMAX / 8; ++q) {
                    const int n0 = 8 * (t + q * T);
                    if (n0 < C) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) o[q * 8 + c] = l[lpad(n0 + c)];
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < EMAX; ++q) {
                    const int n = t + q * T;
                    if (n < C) o[q] = l[lpad(n)];
                }
            }
        });

    ex.each(st, [&](int tid, FftState& s) {
        const int g = tid / T, t = tid % T;
        const int r0 = 2 * (bid * p.nb + g);
        uint32_t nan1 = 0, inf1 = 0, nan2 = 0, inf2 = 0;
        double ss = 0.0;
        // after the swap trick the true (re, im) = (xi, xr): row r0 = re, row r0+1 = im
        static_for<0, 2>([&](auto h_c) {
            constexpr int h = decltype(h_c)::value;
            const int row = r0 + h;
            if (row >= p.R) return;
            const float* x = comp_of<1 - h>(s);
            if (vec) {
                // the add-back loads of the row (16-bit base: the merge itself) go out together
                constexpr int NQ = EMAX / 8;
                const bool base16 = p.base && p.base_dtype != DT_F32;
                u32x4 braw[NQ];
                if (base16) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int n0 = 8 * (t + q * T);
                        braw[q] = ((const u32x4*)p.base)[n0 < C ? ((size_t)row * C + n0) / 8 : 0];
                    }
                }
#pragma unroll
                for (int q = 0; q < EMAX / 8; ++q) {
                    const int n0 = 8 * (t + q * T);
                    if (n0 < C) {
                        const size_t off = (size_t)row * C + n0;
                        float o[8];
                        float bv[8];
                        if (base16) decode16x8(braw[q], p.base_dtype, bv);
                        else if (p.base) load_elem8(p.base, p.base_dtype, off, bv);
                        // fast path: no NaN/Inf anywhere in the group (an all-ones exponent is
                        // looked for with an and + max per value); otherwise the exact policy
                        uint32_t e1 = 0, e2 = 0;
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const float v1 = x[q * 8 + c] * p.inv_n;
                            e1 = umax(e1, f2u(v1) & 0x7f800000u);
                            float v2 = v1 * p.post;
                            if (p.base) v2 += bv[c];
                            e2 = umax(e2, f2u(v2) & 0x7f800000u);
                            o[c] = v2;
                        }
                        if ((p.ifft_policy && e1 == 0x7f800000u) || e2 == 0x7f800000u) {
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                float v = x[q * 8 + c] * p.inv_n;
                                if (p.ifft_policy) {
                                    if (is_nan(v)) { v = 0.f; nan1++; }
                                    if (is_inf(v)) inf1 = 1;
                                }
                                v *= p.post;
                                if (p.base) {
                                    v += bv[c];
                                    if (is_nan(v)) { v = 0.f; nan2++; }
                                    if (is_inf(v)) inf2 = 1;
                                }
                                o[c] = v;
                            }
                        }
                        if (p.norm_partials) {
                            float ps = 0.f;
#pragma unroll
                            for (int c = 0; c < 8; ++c) ps += o[c] * o[c];
                            ss += ps;
                        }
                        if (p.out_mode == OUT_BF16) {
                            u32x4 w;
                            w.x = pack_bf16x2(o[0], o[1]);
                            w.y = pack_bf16x2(o[2], o[3]);
                            w.z = pack_bf16x2(o[4], o[5]);
                            w.w = pack_bf16x2(o[6], o[7]);
                            ((u32x4*)p.out)[off / 8] = w;
                        } else {
                            cf4 w0 = {o[0], o[1], o[2], o[3]}, w1 = {o[4], o[5], o[6], o[7]};
                            ((cf4*)p.out)[off / 4] = w0;
                            ((cf4*)p.out)[off / 4 + 1] = w1;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < EMAX; ++q) {
                    const int n = t + q * T;
                    if (n < C) {
                        const size_t off = (size_t)row * C + n;
                        float v;
                        i2_finish(p, x[q], off, nan1, inf1, nan2, inf2, v);
                        ss += (double)v * v;
                        if (p.out_mode == OUT_BF16) ((uint16_t*)p.out)[off] = f_to_bf16(v);
                        else ((float*)p.out)[off] = v;
                    }
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

// =====================================================================
// streaming kernels over the half-spectrum planes [Cb][R]
// =====================================================================
struct EmptyState { double red[8]; unsigned long long keep[8]; };

SM_HD int sgn(float v) { return (v > 0.f) - (v < 0.f); }
// torch.sign(NaN) = NaN and NaN == NaN is False: NaNs never "agree"
// (on the bit patterns: |x| > inf is a NaN, |x| == 0 is a zero of either sign)
SM_HD bool same_sign(float a, float b) {
    const uint32_t ua = f2u(a), ub = f2u(b);
    const uint32_t ka = ua & 0x7fffffffu, kb = ub & 0x7fffffffu;
    const bool not_nan = ka <= 0x7f800000u && kb <= 0x7f800000u;
    const bool za = ka == 0u, zb = kb == 0u;
    const bool both_nonzero_same = !za && !zb && ((ua ^ ub) >> 31) == 0u;
    return not_nan && ((za && zb) || both_nonzero_same);
}


// selection state (device memory): exact k-th smallest key by radix levels
struct SelState {
    unsigned long long rank;    // in: 0-based rank wanted; updated to rank within the prefix
    uint32_t prefix;            // key bits decided so far
    uint32_t level;             // levels done
    float value;                // final: the k-th smallest |x|
    uint32_t pad;
};

struct HistParams {
    const float* X; const float* Y;     // Y may be null
    int R, C, Cb;
    int vec4;                           // R % 4 == 0: float4 loads, one weight per quad
    int level;                          // 1, 2 or 3
    const SelState* sel;
    unsigned long long* hist;
    int chunks;                         // quads per thread
    const uint32_t* only_if;            // when set: run only if *only_if != 0 (fallback after a list overflow)
};

// load up to 4 consecutive plane values starting at element i0
SM_HD int load_quad(const float* src, size_t i0, size_t total, int vec4, float* out) {
    if (vec4) {
        const cf4 v = *(const cf4*)(src + i0);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        return 4;
    }
    int n = 0;
    for (int e = 0; e < 4; ++e) {
        if (i0 + e < total) { out[e] = src[i0 + e]; n = e + 1; } else out[e] = 0.f;
    }
    return n;
}

template <class Ex>
SM_HD void k_hist(Ex& ex, const HistParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    uint32_t* lh = (uint32_t*)(ex.lds() + LDS_SCRATCH_FLOATS);
    const int nt = ex.nthreads();
    const int nbins = p.level == 1 ? HIST1_BINS : HIST_LO_BINS;
    const size_t total = (size_t)p.Cb * p.R;
    const size_t nquad = (total + 3) / 4;
    if (p.only_if && !*p.only_if) return;
    const uint32_t prefix = p.sel->prefix;
    const WeightRanges wr = weight_ranges(p.R, p.C, p.Cb);
    ex.each(st, [&](int tid, EmptyState&) { for (int b = tid; b < nbins; b += nt) lh[b] = 0; });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int c = 0; c < p.chunks; ++c) {
            const size_t qi = start + (size_t)c * nt + tid;
            if (qi >= nquad) break;
            const size_t i0 = 4 * qi;
            for (int which = 0; which < 2; ++which) {
                const float* src = which ? p.Y : p.X;
                if (!src) continue;
                float v[4];
                const int n = load_quad(src, i0, total, p.vec4, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e < n) {
                        const uint32_t w = weight_at(wr, i0 + e);
                        const uint32_t key = f2u(v[e]) & 0x7fffffffu;
                        if (p.level == 1) ex.lds_atomic_add(&lh[key >> 20], w);
                        else if (p.level == 2) { if ((key >> 20) == prefix) ex.lds_atomic_add(&lh[(key >> 10) & 1023u], w); }
                        else { if ((key >> 10) == prefix) ex.lds_atomic_add(&lh[key & 1023u], w); }
                    }
                }
            }
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        for (int b = tid; b < nbins; b += nt) {
            const uint32_t v = lh[b];
            if (v) ex.global_atomic_add(&p.hist[b], (unsigned long long)v);
        }
    });
}

// ---------------------------------------------------------------------------
// Level-2 selection pass with candidate compaction (and, for the cutoff
// threshold, the masked slerp sums fused in).
//
// After level 1 the k-th smallest key is known to lie in bin B1 = sel->prefix
// (key >> 20).  One streaming pass over the plane(s)
//   * histograms bits [19:10] of the keys in B1 (as k_hist level 2 does), and
//   * appends those keys (~1 % of the data) to a candidate list, so that level 3
//     runs on the list instead of re-reading the planes;
//   * with `fuse_reduce`: accumulates the slerp-class sums over every bin whose
//     class does not depend on the exact threshold (|r1|'s level-1 bin above B1),
//     and appends the undecided bins (|r1| in B1, signs agree) to a pair list that
//     k_reduce_cand settles once the threshold is known.
// Lists are staged in LDS and appended with one global atomic per work-group.
// Any overflow (staging or list capacity) raises *overflow; the caller then
// falls back to the plain full passes (k_hist level 3 / k_reduce), which exit
// immediately otherwise.
// ---------------------------------------------------------------------------
constexpr int STAGE_KEYS = 2048;     // per work-group staging (expected ~160 keys / ~40 pairs)
constexpr int STAGE_PAIRS = 1024;
struct CandLists {
    uint32_t* keys;        // key | (weight-1) << 31
    cf4* pairs;            // (r0, r1, weight, 0)
    uint32_t cap_keys, cap_pairs;
    uint32_t* counters;    // [0] n_keys, [1] n_pairs, [2] overflow
};
// blend constants (device memory), written by k_slerp_consts
struct BlendConsts {
    float thr;        // cutoff threshold (0 when cutoff_pct == 0)
    float dot;        // clamped cosine between the slerp-class vectors
    float cos_t, sin_t;
    float inv_rel;    // 1 / max(||v1 - dot v0||, 1e-12)
    float pad[3];
    double s00, s01, s11;
    unsigned long long n_slerp;
};

struct Select2Params {
    const float* X; const float* Y;     // Y may be null
    int R, C, Cb;
    int vec4;
    SelState* sel;                      // written by work-group 0: level-1 bin and rank inside it
    const unsigned long long* hist1;    // level-1 histogram (HIST1_BINS), complete
    unsigned long long rank;            // 0-based rank wanted
    unsigned long long* hist;           // level-2 histogram (HIST_LO_BINS), accumulated here
    CandLists cand;
    int fuse_reduce;                    // needs Y: X = Re a, Y = Re b
    int sumsq;                          // Y == null: partials[4*wg] += sum w x^2 over the elements whose level-1 bin lies
                                        // ABOVE the threshold's (they survive the cull whatever its low bits): the
                                        // Parseval norm of a spectral intermediate, fused into the cull selection
    double* partials;                   // [grid][4] when fuse_reduce or sumsq
    int chunks;
    int flush_always;          // test hook: flush the staged candidates after every round
    const uint32_t* skip;      // plain mode, optional: *skip != 0 -> nothing to do (the speculative pass below did it)
    // BLEND mode (k_select2<true>): X = Re a, blendB = Re b; the kernel computes Re R = blend(a, b), stores it,
    // takes its level-1 histogram - and, SPECULATING that the cull threshold's level-1 bin is *guess - 1 (what
    // it was the last time, 0 = no guess), does this selection pass's work on it in the same sweep.
    // k_spec_check confirms the guess from the complete histogram or voids the speculative output.
    const float* blendB; float* blendR;
    const BlendConsts* consts; float t_sum;
    unsigned long long* hist1_out;
    const uint32_t* guess;
};

SM_HD float blend_slerp_one(const BlendConsts& c, float t_sum, float a, float b) {      // = blend_one(), BLEND_SLERP
    if (same_sign(a, b)) {
        if (fabsf(b) < c.thr) return a + t_sum * b;
        return a * c.cos_t + ((b - a * c.dot) * c.inv_rel) * c.sin_t;
    }
    return (fabsf(a) > fabsf(b)) ? a : b;
}

template <bool BLEND = false, class Ex>
SM_HD void k_select2(Ex& ex, const Select2Params& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    uint32_t* lh = (uint32_t*)(ex.lds() + LDS_SCRATCH_FLOATS);
    uint32_t* lctl = lh + HIST_LO_BINS;                 // [0] nkeys [1] npairs [2] base keys [3] base pairs [4..6] resolve
    uint32_t* lkeys = lctl + 8;
    cf4* lpairs = (cf4*)(lkeys + STAGE_KEYS);
    const int nt = ex.nthreads();
    const size_t total = (size_t)p.Cb * p.R;
    const size_t nquad = (total + 3) / 4;
    uint32_t* lh1 = lkeys + STAGE_KEYS;                 // BLEND: level-1 histogram of Re R (no pair staging there)
    uint32_t prefix;
    BlendConsts bc;
    if constexpr (BLEND) {
        prefix = *p.guess - 1u;                          // 0xffffffff: no guess, nothing matches
        bc = *p.consts;
    } else {
        if (p.skip && *p.skip) return;
        // level 1: every work-group resolves it for itself (the staging area doubles as scratch)
        wg_resolve(ex, st, p.hist1, HIST1_BINS, p.rank, (unsigned long long*)lkeys, lctl + 4);
        prefix = lctl[4];
        if (ex.bid() == 0) {
            ex.each(st, [&](int tid, EmptyState&) {
                if (tid == 0) {
                    p.sel->prefix = prefix; p.sel->level = 1;
                    p.sel->rank = (unsigned long long)lctl[5] | ((unsigned long long)lctl[6] << 32);
                }
            });
        }
        ex.sync();
    }
    const WeightRanges wr = weight_ranges(p.R, p.C, p.Cb);
    ex.each(st, [&](int tid, EmptyState&) {
        for (int b = tid; b < HIST_LO_BINS; b += nt) lh[b] = 0;
        if (tid < 8) lctl[tid] = 0;
        if constexpr (BLEND) for (int b = tid; b < HIST1_BINS; b += nt) lh1[b] = 0;
    });
    ex.sync();
    // The stream is cut into rounds of ROUND steps; after each round the staged candidates go
    // to the global lists, so a work-group's staging area only has to hold one round's worth
    // (a 235 M-element tensor used to overflow it - and redo the whole layer in safe mode)
#ifndef SM_SEL_ROUND
#define SM_SEL_ROUND 16
#endif
#ifndef SM_SEL_U
#define SM_SEL_U 4
#endif
#ifndef SM_DIAG_SEL
#define SM_DIAG_SEL 0          // diagnostic builds: bit 0 drops the staging path, bit 1 the sums
#endif
    constexpr int ROUND = SM_SEL_ROUND;
    ex.each(st, [&](int, EmptyState& s) { s.red[0] = s.red[1] = s.red[2] = s.red[3] = 0.0; });
    for (int r0 = 0; r0 < p.chunks; r0 += ROUND) {
    const int r1 = (r0 + ROUND < p.chunks) ? r0 + ROUND : p.chunks;
    ex.each(st, [&](int tid, EmptyState& s) {
        double s00 = 0, s01 = 0, s11 = 0, cnt = 0;
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        const bool hasY = !BLEND && p.Y != nullptr;
        // up to 4 consecutive plane elements from i0 on
        // uniform_w: the 4 elements lie in one column (R % 4 == 0), one weight serves all
        auto quad = [&](size_t i0, const float* a_in, const float* b, int n, bool uniform_w) {
            float q00 = 0.f, q01 = 0.f, q11 = 0.f, qc = 0.f;
            const uint32_t w0 = weight_at(wr, i0);
            float rr[4];
            const float* a = a_in;
            if constexpr (BLEND) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rr[e] = blend_slerp_one(bc, p.t_sum, a_in[e], b[e]);
                if (n == 4 && uniform_w) { cf4 v = {rr[0], rr[1], rr[2], rr[3]}; *(cf4*)(p.blendR + i0) = v; }
                else for (int e = 0; e < n; ++e) p.blendR[i0 + e] = rr[e];
                a = rr;
            }
            // keys of the selected level-1 bin: lo <= key < lo + 2^20, i.e. (key - lo) < 2^20 unsigned; above it: key >= hi
            // (prefix 0xffffffff - no guess - gives lo = 0xfff00000, which no 31-bit key reaches from either side)
            const uint32_t lo = prefix << 20, hi = lo + (1u << 20);
            const bool has_hi = prefix < 2047u;
            uint32_t ka[4], kb[4], wv[4];
            bool hit = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wv[e] = (e < n) ? (uniform_w ? w0 : weight_at(wr, i0 + e)) : 0u;
                ka[e] = f2u(a[e]) & 0x7fffffffu;
                kb[e] = hasY ? (f2u(b[e]) & 0x7fffffffu) : 0u;
                hit = hit || ((e < n) && ((ka[e] - lo) < (1u << 20) || (hasY && (kb[e] - lo) < (1u << 20))));
            }
            if constexpr (BLEND) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < n) ex.lds_atomic_add(&lh1[ka[e] >> 20], wv[e]);
            }
            // The bin holds a few per cent of the elements: one test per quad, the staging work behind it
            if (hit && !(SM_DIAG_SEL & 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e < n) {
                        const uint32_t w = wv[e];
                        if ((ka[e] - lo) < (1u << 20)) {
                            ex.lds_atomic_add(&lh[(ka[e] >> 10) & 1023u], w);
                            const uint32_t pos = ex.lds_atomic_add_ret(&lctl[0], 1u);
                            if (pos < (uint32_t)STAGE_KEYS) lkeys[pos] = ka[e] | ((w - 1u) << 31);
                        }
                        if (hasY && (kb[e] - lo) < (1u << 20)) {
                            ex.lds_atomic_add(&lh[(kb[e] >> 10) & 1023u], w);
                            const uint32_t pos = ex.lds_atomic_add_ret(&lctl[0], 1u);
                            if (pos < (uint32_t)STAGE_KEYS) lkeys[pos] = kb[e] | ((w - 1u) << 31);
                            if (p.fuse_reduce && same_sign(a[e], b[e])) {
                                const uint32_t pp = ex.lds_atomic_add_ret(&lctl[1], 1u);
                                if (pp < (uint32_t)STAGE_PAIRS) { cf4 v = {a[e], b[e], (float)w, 0.f}; lpairs[pp] = v; }
                            }
                        }
                    }
                }
            }
            // the sums over what lies ABOVE the bin (it survives whatever the threshold's low bits): branch-free, a value
            // outside the class enters as an exact zero
            if (p.sumsq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = (e < n) && has_hi && ka[e] >= hi;
                    const float av = in ? a[e] : 0.f;
                    q00 += ((float)wv[e] * av) * av;
                }
            }
            if (hasY && p.fuse_reduce && !(SM_DIAG_SEL & 2)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = (e < n) && has_hi && kb[e] >= hi && same_sign(a[e], b[e]);   // |r1| >= threshold whatever its low bits
                    const float wf = (float)wv[e];
                    const float av = in ? a[e] : 0.f, bv = in ? b[e] : 0.f;
                    const float wa = wf * av;
                    q00 += wa * av; q01 += wa * bv; q11 += (wf * bv) * bv; qc += in ? wf : 0.f;
                }
            }
            s00 += q00; s01 += q01; s11 += q11; cnt += qc;
        };
        if (p.vec4) {
            // 16-byte loads, U steps in flight together: the address is clamped instead of
            // branched around so that the loads issue back to back
            constexpr int U = SM_SEL_U;
            const cf4* X4 = (const cf4*)p.X;
            const cf4* Y4 = (const cf4*)(BLEND ? p.blendB : p.Y);
            for (int c0 = r0; c0 < r1; c0 += U) {
                cf4 av[U], bv[U];
                size_t qv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    qv[u] = start + (size_t)(c0 + u) * nt + tid;
                    av[u] = X4[qv[u] < nquad ? qv[u] : nquad - 1];
                }
                if (hasY || BLEND) {
#pragma unroll
                    for (int u = 0; u < U; ++u) bv[u] = Y4[qv[u] < nquad ? qv[u] : nquad - 1];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (c0 + u < r1 && qv[u] < nquad) {
                        const float a[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
                        const float b[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
                        quad(4 * qv[u], a, b, 4, !wr.full);
                    }
                }
            }
        } else {
            for (int c = r0; c < r1; ++c) {
                const size_t qi = start + (size_t)c * nt + tid;
                if (qi >= nquad) break;
                float a[4], b[4] = {0.f, 0.f, 0.f, 0.f};
                const int n = load_quad(p.X, 4 * qi, total, 0, a);
                if (hasY) load_quad(p.Y, 4 * qi, total, 0, b);
                if constexpr (BLEND) load_quad(p.blendB, 4 * qi, total, 0, b);
                quad(4 * qi, a, b, n, false);
            }
        }
        s.red[0] += s00; s.red[1] += s01; s.red[2] += s11; s.red[3] += cnt;
    });
    ex.sync();
    // flush when the staging area is more than half full (the counts are the same for every
    // thread after the barrier, so the branch is uniform), and at the end
    const bool last_round = r1 >= p.chunks;
    if (!last_round && !p.flush_always && lctl[0] <= (uint32_t)STAGE_KEYS / 2 && lctl[1] <= (uint32_t)STAGE_PAIRS / 2) continue;
    ex.each(st, [&](int tid, EmptyState&) {
        if (tid == 0) {
            uint32_t nk = lctl[0], np = lctl[1];
            bool over = false;
            if (nk > (uint32_t)STAGE_KEYS) { nk = STAGE_KEYS; over = true; }
            if (np > (uint32_t)STAGE_PAIRS) { np = STAGE_PAIRS; over = true; }
            const uint32_t bk = nk ? ex.global_atomic_add_ret_u32(&p.cand.counters[0], nk) : 0u;
            const uint32_t bp = np ? ex.global_atomic_add_ret_u32(&p.cand.counters[1], np) : 0u;
            if (bk + nk > p.cand.cap_keys || bp + np > p.cand.cap_pairs) over = true;
            // (a speculative pass leaves the sticky word to k_spec_check: an overflow on a wrong guess is void)
            if (over) { ex.global_atomic_or_u32(&p.cand.counters[2], 1u); if (!BLEND) ex.global_atomic_or_u32(&p.cand.counters[3], 1u); }
            lctl[0] = nk; lctl[1] = np; lctl[2] = bk; lctl[3] = bp;
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        const uint32_t nk = lctl[0], np = lctl[1], bk = lctl[2], bp = lctl[3];
        for (uint32_t q = tid; q < nk; q += nt) if (bk + q < p.cand.cap_keys) p.cand.keys[bk + q] = lkeys[q];
        for (uint32_t q = tid; q < np; q += nt) if (bp + q < p.cand.cap_pairs) p.cand.pairs[bp + q] = lpairs[q];
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) { if (tid < 4) lctl[tid] = 0; });
    ex.sync();
    }   // rounds
    ex.each(st, [&](int tid, EmptyState&) {
        for (int b = tid; b < HIST_LO_BINS; b += nt) {
            const uint32_t v = lh[b];
            if (v) ex.global_atomic_add(&p.hist[b], (unsigned long long)v);
        }
        if constexpr (BLEND) {
            for (int b = tid; b < HIST1_BINS; b += nt) {
                const uint32_t v = lh1[b];
                if (v) ex.global_atomic_add(&p.hist1_out[b], (unsigned long long)v);
            }
        }
    });
    if (p.fuse_reduce || p.sumsq) {
        ex.sync();
        ex.template block_sum<4>(st, [&](const double* tot) {
            for (int q = 0; q < 4; ++q) p.partials[4 * (size_t)ex.bid() + q] = tot[q];
        });
    }
}

// Single work-group: was the speculated level-1 bin of k_select2<true> the right one?  Resolves level 1 from
// the complete histogram, remembers it as the next guess and either confirms (writes the selection state
// exactly as the plain pass's work-group 0 does, *flag = 1: the plain pass that follows returns at once)
// or voids the speculative output (level-2 histogram and list counters cleared, *flag = 0).
struct SpecCheckParams {
    const unsigned long long* hist1; unsigned long long rank;
    uint32_t* guess; uint32_t* flag;
    SelState* sel;
    unsigned long long* hist2;
    uint32_t* counters;
};
template <class Ex>
SM_HD void k_spec_check(Ex& ex, const SpecCheckParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    unsigned long long* scratch = (unsigned long long*)(ex.lds() + LDS_SCRATCH_FLOATS);
    uint32_t* res = (uint32_t*)(scratch + nt + 16);             // behind wg_resolve's partials and group totals
    wg_resolve(ex, st, p.hist1, HIST1_BINS, p.rank, scratch, res);
    ex.each(st, [&](int tid, EmptyState&) {
        if (tid == 0) {
            const uint32_t g = *p.guess;
            const uint32_t hit = (g == res[0] + 1u) ? 1u : 0u;
            res[3] = hit;
            *p.flag = hit;
            p.flag[1] += 1u; p.flag[2] += hit;               // running totals: speculations checked / confirmed
            *p.guess = res[0] + 1u;
            if (hit) {
                p.sel->prefix = res[0]; p.sel->level = 1;
                p.sel->rank = (unsigned long long)res[1] | ((unsigned long long)res[2] << 32);
                if (p.counters[2]) p.counters[3] = 1u;           // the overflow was real
            }
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        if (res[3]) return;
        for (int b = tid; b < HIST_LO_BINS; b += nt) p.hist2[b] = 0ull;
        if (tid < 3) p.counters[tid] = 0u;
    });
}

// the candidates' share of the Parseval sum once the cull threshold is known (sumsq above)
struct SumsqCandParams {
    CandLists cand;
    const float* thr;
    double* partials;          // [nblocks][4], appended after the selection pass's partials
};
template <class Ex>
SM_HD void k_sumsq_cand(Ex& ex, const SumsqCandParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const uint32_t n = p.cand.counters[2] ? 0u : p.cand.counters[0];
    const float thr = *p.thr;
    ex.each(st, [&](int tid, EmptyState& s) {
        double acc = 0;
        for (size_t q = (size_t)ex.bid() * nt + tid; q < n; q += (size_t)ex.nblocks() * nt) {
            const uint32_t v = p.cand.keys[q];
            const float x = u2f(v & 0x7fffffffu);
            if (!(x < thr)) acc += (double)(1u + (v >> 31)) * (double)x * (double)x;
        }
        s.red[0] = acc; s.red[1] = 0; s.red[2] = 0; s.red[3] = 0;
    });
    ex.template block_sum<4>(st, [&](const double* tot) {
        for (int q = 0; q < 4; ++q) p.partials[4 * (size_t)ex.bid() + q] = tot[q];
    });
}
// S_re = sum of partials[4i], S_im = sum of im_partials[i] -> out[0], out[1]; clears the list counters
struct SumSpecParams { const double* part4; int n4; const double* part1; int n1; double* out; uint32_t* zero_u32; int zero_u32_count; };
template <class Ex>
SM_HD void k_sum_spec(Ex& ex, const SumSpecParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState& s) {
        double a0 = 0, a1 = 0;
        for (int i = tid; i < p.n4; i += nt) a0 += p.part4[4 * (size_t)i];
        for (int i = tid; i < p.n1; i += nt) a1 += p.part1[i];
        if (p.zero_u32 && tid < p.zero_u32_count) p.zero_u32[tid] = 0u;
        s.red[0] = a0; s.red[1] = a1;
    });
    ex.template block_sum<2>(st, [&](const double* tot) { p.out[0] = tot[0]; p.out[1] = tot[1]; });
}

// level 3 on the candidate list (or nothing when the list overflowed)
struct Select3Params {
    CandLists cand;
    SelState* sel;                      // in: level-1 prefix/rank; work-group 0 writes the level-2 ones
    const unsigned long long* hist2;    // level-2 histogram, complete
    unsigned long long* hist;           // level-3 histogram, accumulated here
};
template <class Ex>
SM_HD void k_select3(Ex& ex, const Select3Params& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    uint32_t* lh = (uint32_t*)(ex.lds() + LDS_SCRATCH_FLOATS);
    uint32_t* lres = lh + HIST_LO_BINS;
    unsigned long long* part = (unsigned long long*)(lres + 8);
    const int nt = ex.nthreads();
    const uint32_t n = p.cand.counters[2] ? 0u : p.cand.counters[0];     // overflow: the caller redoes the layer
    const uint32_t prefix1 = p.sel->prefix;
    const unsigned long long rank1 = p.sel->rank;
    wg_resolve(ex, st, p.hist2, HIST_LO_BINS, rank1, part, lres);
    const uint32_t prefix = (prefix1 << 10) | lres[0];
    const unsigned long long rank2 = (unsigned long long)lres[1] | ((unsigned long long)lres[2] << 32);
    ex.each(st, [&](int tid, EmptyState&) { for (int b = tid; b < HIST_LO_BINS; b += nt) lh[b] = 0; });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        for (size_t q = (size_t)ex.bid() * nt + tid; q < n; q += (size_t)ex.nblocks() * nt) {
            const uint32_t v = p.cand.keys[q];
            const uint32_t key = v & 0x7fffffffu;
            if ((key >> 10) == prefix) ex.lds_atomic_add(&lh[key & 1023u], 1u + (v >> 31));
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        for (int b = tid; b < HIST_LO_BINS; b += nt) {
            const uint32_t v = lh[b];
            if (v) ex.global_atomic_add(&p.hist[b], (unsigned long long)v);
        }
    });
    // every work-group has read the level-1 state before any can get here? No: work-group 0
    // may run ahead, so the level-2 state goes to the SECOND slot (sel + 1)
    if (ex.bid() == 0) {
        ex.each(st, [&](int tid, EmptyState&) {
            if (tid == 0) { p.sel[1].prefix = prefix; p.sel[1].rank = rank2; p.sel[1].level = 2; }
        });
    }
}

// settle the undecided pairs once the cutoff threshold is known
struct ReduceCandParams {
    CandLists cand;
    const float* thr;
    double* partials;          // [nblocks][4], appended after the fused pass's partials
};
template <class Ex>
SM_HD void k_reduce_cand(Ex& ex, const ReduceCandParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const bool over = p.cand.counters[2] != 0;
    const uint32_t n = over ? 0u : p.cand.counters[1];
    const float thr = *p.thr;
    ex.each(st, [&](int tid, EmptyState& s) {
        double s00 = 0, s01 = 0, s11 = 0, cnt = 0;
        for (size_t q = (size_t)ex.bid() * nt + tid; q < n; q += (size_t)ex.nblocks() * nt) {
            const cf4 v = p.cand.pairs[q];
            if (!(fabsf(v.y) < thr)) {
                s00 += (double)v.z * v.x * v.x; s01 += (double)v.z * v.x * v.y; s11 += (double)v.z * v.y * v.y; cnt += v.z;
            }
        }
        s.red[0] = s00; s.red[1] = s01; s.red[2] = s11; s.red[3] = cnt;
    });
    ex.template block_sum<4>(st, [&](const double* tot) {
        for (int q = 0; q < 4; ++q) p.partials[4 * (size_t)ex.bid() + q] = tot[q];
    });
}

// Work-group-wide resolution of one radix-select level: which bin of `hist` holds the
// element of 0-based rank `rank`, and its rank inside that bin.  Every thread of a
// 256-thread work-group takes part; the result lands in res[0] (bin), res[1..2] (new
// rank, lo/hi).  Consumer kernels run this on the previous level's histogram instead of
// waiting for a separate single-work-group scan launch.
template <class Ex, class StT>
SM_HD void wg_resolve(Ex& ex, StT& st, const unsigned long long* hist, int nbins, unsigned long long rank,
                      unsigned long long* part, uint32_t* res) {
    using S = typename StT::value_type;
    const int nt = ex.nthreads();                       // 256
    const int per = (nbins + nt - 1) / nt;              // <= 8: the thread's bins stay in its state
    unsigned long long* grp = part + nt;                // 16 group totals (the scratch holds 2 * nt words)
    ex.each(st, [&](int tid, S& s) {
        unsigned long long sum = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int b = tid * per + q;
            const unsigned long long h = (q < per && b < nbins) ? hist[b] : 0ull;
            s.keep[q] = h;
            sum += h;
        }
        part[tid] = sum;
        if (tid == 0) { res[0] = 0; res[1] = 0; res[2] = 0; }
    });
    ex.sync();
    ex.each(st, [&](int tid, S&) {                      // 16 threads total 16 partials each
        if (tid < 16) {
            unsigned long long g = 0;
            for (int q = 0; q < 16; ++q) { const int i = tid * 16 + q; if (i < nt) g += part[i]; }
            grp[tid] = g;
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, S& s) {
        unsigned long long excl = 0, total = 0;
        const int g0 = tid / 16;
        for (int g = 0; g < 16; ++g) { const unsigned long long v = grp[g]; total += v; if (g < g0) excl += v; }
        for (int q = g0 * 16; q < tid; ++q) excl += part[q];
        if (total == 0) return;
        unsigned long long r = rank >= total ? total - 1 : rank;
        if (r >= excl && r < excl + part[tid]) {
            unsigned long long cum = excl;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned long long h = s.keep[q];
                if (q < per && r >= cum && r < cum + h) {
                    res[0] = (uint32_t)(tid * per + q);
                    const unsigned long long nr = r - cum;
                    res[1] = (uint32_t)(nr & 0xffffffffull); res[2] = (uint32_t)(nr >> 32);
                }
                cum += h;
            }
        }
    });
    ex.sync();
}

struct ScanParams {
    unsigned long long* hist;   // consumed and zeroed
    SelState* sel;
    int nbins;                  // HIST1_BINS or HIST_LO_BINS
    int shift;                  // bits this level contributes (11 or 10)
    int final_level;            // 1: write sel->value
    float* value_out;           // optional extra copy of the value
    int init;                   // 1: first level, start from rank_init / empty prefix
    unsigned long long rank_init;
    unsigned long long* zero_also; int zero_count;   // further histogram words to clear (fast path: all levels)
    uint32_t* zero_u32; int zero_u32_count;          // and 32-bit words (the candidate-list counters)
};

// single work-group of 256 threads
template <class Ex>
SM_HD void k_scan(Ex& ex, const ScanParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    unsigned long long* part = (unsigned long long*)(ex.lds() + LDS_SCRATCH_FLOATS);
    const int nt = ex.nthreads();                       // 256
    const int per = (p.nbins + nt - 1) / nt;            // <= 8
    unsigned long long* grp = part + nt;                // 16 group totals
    ex.each(st, [&](int tid, EmptyState& s) {
        unsigned long long sum = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int b = tid * per + q;
            const unsigned long long h = (q < per && b < p.nbins) ? p.hist[b] : 0ull;
            s.keep[q] = h;
            sum += h;
        }
        part[tid] = sum;
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        if (tid < 16) {
            unsigned long long g = 0;
            for (int q = 0; q < 16; ++q) { const int i = tid * 16 + q; if (i < nt) g += part[i]; }
            grp[tid] = g;
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState& s) {
        unsigned long long excl = 0, total = 0;
        const int g0 = tid / 16;
        for (int g = 0; g < 16; ++g) { const unsigned long long v = grp[g]; total += v; if (g < g0) excl += v; }
        for (int q = g0 * 16; q < tid; ++q) excl += part[q];
        unsigned long long rank = p.init ? p.rank_init : p.sel->rank;
        const uint32_t prefix0 = p.init ? 0u : p.sel->prefix;
        const uint32_t level0 = p.init ? 0u : p.sel->level;
        if (total == 0) { if (tid == 0) { p.sel->value = 0.f; if (p.value_out) *p.value_out = 0.f; } return; }
        if (rank >= total) rank = total - 1;
        if (rank >= excl && rank < excl + part[tid]) {
            unsigned long long cum = excl;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned long long h = s.keep[q];
                if (q < per && rank >= cum && rank < cum + h) {
                    const uint32_t np = (prefix0 << p.shift) | (uint32_t)(tid * per + q);
                    p.sel->prefix = np;
                    p.sel->rank = rank - cum;
                    p.sel->level = level0 + 1;
                    if (p.final_level) {
                        p.sel->value = u2f(np);
                        if (p.value_out) *p.value_out = u2f(np);
                    }
                }
                cum += h;
            }
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        for (int b = tid * per; b < (tid + 1) * per && b < p.nbins; ++b) p.hist[b] = 0;
        if (p.zero_also) for (int b = tid; b < p.zero_count; b += nt) p.zero_also[b] = 0;
        if (p.zero_u32) for (int b = tid; b < p.zero_u32_count; b += nt) p.zero_u32[b] = 0u;
    });
}


struct ReduceParams {
    const float* reA; const float* reB;
    int R, C, Cb;
    int vec4;
    const float* thr;           // device scalar (cutoff threshold) or null (-> 0)
    double* partials;           // [grid][4]: s00, s01, s11, count
    int chunks;
    const uint32_t* only_if;    // when set: if *only_if == 0 write zero partials and leave
};


template <class Ex>
SM_HD void k_reduce(Ex& ex, const ReduceParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t total = (size_t)p.Cb * p.R;
    const size_t nquad = (total + 3) / 4;
    const float thr = p.thr ? *p.thr : 0.f;
    const WeightRanges wr = weight_ranges(p.R, p.C, p.Cb);
    const bool skip = p.only_if && !*p.only_if;
    ex.each(st, [&](int tid, EmptyState& s) {
        double s00 = 0, s01 = 0, s11 = 0, cnt = 0;
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int c = 0; c < p.chunks && !skip; ++c) {
            const size_t qi = start + (size_t)c * nt + tid;
            if (qi >= nquad) break;
            const size_t i0 = 4 * qi;
            float a[4], b[4];
            const int n = load_quad(p.reA, i0, total, p.vec4, a);
            load_quad(p.reB, i0, total, p.vec4, b);
            float q00 = 0.f, q01 = 0.f, q11 = 0.f, qc = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e < n && same_sign(a[e], b[e]) && !(fabsf(b[e]) < thr)) {
                    const float w = (float)weight_at(wr, i0 + e);
                    q00 += w * a[e] * a[e]; q01 += w * a[e] * b[e]; q11 += w * b[e] * b[e]; qc += w;
                }
            }
            s00 += q00; s01 += q01; s11 += q11; cnt += qc;
        }
        s.red[0] = s00; s.red[1] = s01; s.red[2] = s11; s.red[3] = cnt;
    });
    ex.template block_sum<4>(st, [&](const double* tot) {
        for (int q = 0; q < 4; ++q) p.partials[4 * (size_t)ex.bid() + q] = tot[q];
    });
}

struct SlerpConstParams {
    const double* partials; int nparts;      // fused/definite part + candidate part (contiguous)
    const double* fallback; int nfallback;   // full-pass partials, used instead when *overflow != 0
    const uint32_t* overflow;
    const float* thr;           // device scalar or null (-> 0)
    float t;
    BlendConsts* out;
    uint32_t* zero_u32; int zero_u32_count;  // candidate-list counters to clear for the next selection (or null)
    const float* ref_norms;     // norm_mode = reference_cpu, ordered emulation (test hook): ||v0||, ||v1|| of the gathered
                                // class as torch.norm returns them on CPU (sm_aten_norm.hpp), or null
    const double* emf_part;     // norm_mode = reference_cpu: k_class_emf's partials [emf_nparts][2 * EMF_VALS] (or null)
    int emf_nparts;
    int emf_elo;                // binade of its first rounding level
};

// ---- norm_mode = reference_cpu: what torch.norm makes of the gathered slerp-class vectors -----------------------
// (reference functions.py:36,40: v0.norm(), v1.norm() on fp32 CPU tensors of ~0.4 n elements.)  ATen's kernel
// accumulates fma(x, x, acc) serially in 8 fp32 lanes (sm_aten_norm.hpp), which loses low bits once a lane's sum is
// large: -2e-4 at 16 M elements, -1.5e-3 at 67 M.  The reference's cosine is taken with THOSE norms.  Bit-identity
// is not on offer here (the reference gathers its own spectrum's values in its own order), but the bias is a
// statistical property of the values: while a lane's sum S sits in the binade with ulp u, an element moves it by
// rne(x^2 / u) u, so with g(u) = the class mean of that quantity the sum follows dS/di = g(ulp(S)) binade by binade.
// k_class_emf takes g for EMF_LEVELS consecutive binades (and the exact mean) from a sample of the planes (all of
// them below 4 M elements, 1 piece in 16 from 64 M on); k_slerp_consts integrates the trajectory for cnt / 8 elements per lane and scales the exact norm by the
// ratio.  Against torch.norm on the reference's own gathered vectors: 1e-6 ... 4e-6 where the exact norm is off by
// 2e-4 (4096^2; oracle/aten_norm_model_probe.py); the ordered emulation of sm_aten_norm.hpp (test hook
// "class_norms" = 2) gives the same to 2e-6.
constexpr int EMF_LEVELS = 16;
constexpr int EMF_VALS = EMF_LEVELS + 2;        // weighted count, sum of squares, EMF_LEVELS sums of rounded squares
#ifndef SM_EMF_MAX_SAMPLE
#define SM_EMF_MAX_SAMPLE 16
#endif
constexpr int EMF_MAX_SAMPLE = SM_EMF_MAX_SAMPLE;   // at most one piece of 8 rows (64 plane elements) in every 16 is read ...
constexpr size_t EMF_MIN_SAMPLED = (size_t)2 << 20;   // ... as long as this many elements are
struct ClassEmfParams {
    const float* reA; const float* reB;         // planes [Cb][R]
    const float* thr;                           // device scalar (cutoff threshold) or null (-> 0)
    int R, C, Cb;
    size_t n;                                   // plane elements
    int elo;                                    // level k rounds to multiples of 2^(elo + k - 23)
    int sample;                                 // one piece of 8 rows in every `sample`
    int iters;                                  // pieces per 8 threads
    double* partials;                           // [grid][2 * EMF_VALS]
};
struct EmfState { double red[2 * EMF_VALS]; };
SM_HD double emf_pow2(int e) {
    if (e < -1000) e = -1000;
    if (e > 1000) e = 1000;
    const unsigned long long b = (unsigned long long)(1023 + e) << 52;
    double d; memcpy(&d, &b, 8); return d;
}
template <class Ex>
SM_HD void k_class_emf(Ex& ex, const ClassEmfParams& p) {
    typename Ex::template State<EmfState> st;
    ex.init(st);
    const int nt = ex.nthreads();                       // 256: 32 pieces of 8 rows per sweep
    const size_t rows = (p.n + 7) / 8;
    const size_t npieces = (rows + 8 * (size_t)p.sample - 1) / (8 * (size_t)p.sample);
    const float thr = p.thr ? *p.thr : 0.f;
    const WeightRanges wr = weight_ranges(p.R, p.C, p.Cb);
    const double sc0 = emf_pow2(23 - p.elo);
    ex.each(st, [&](int tid, EmfState& q) {
        double acc[2 * EMF_VALS];
#pragma unroll
        for (int i = 0; i < 2 * EMF_VALS; ++i) acc[i] = 0.0;
        for (int it = 0; it < p.iters; ++it) {
            const size_t piece = ((size_t)ex.bid() * p.iters + it) * (nt / 8) + (size_t)(tid / 8);
            const size_t r = piece * (8 * (size_t)p.sample) + (size_t)(tid % 8);
            // (the address is clamped instead of branched around: a load behind a branch is waited for on the spot;
            // a ragged last row of fewer than 8 elements is left out of the sample)
            const bool live = piece < npieces && r < rows && r * 8 + 8 <= p.n;
            const size_t i0 = live ? r * 8 : 0;
            const cf4 a0 = ((const cf4*)p.reA)[i0 / 4], a1 = ((const cf4*)p.reA)[i0 / 4 + 1];
            const cf4 b0 = ((const cf4*)p.reB)[i0 / 4], b1 = ((const cf4*)p.reB)[i0 / 4 + 1];
            const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = live && same_sign(a[e], b[e]) && !(fabsf(b[e]) < thr);
                if (!in) continue;
                const double w = (double)weight_at(wr, i0 + e);
                const double ya = (double)a[e] * (double)a[e], yb = (double)b[e] * (double)b[e];
                acc[0] += w; acc[1] += w * ya;
                acc[EMF_VALS] += w; acc[EMF_VALS + 1] += w * yb;
                double va = ya * sc0, vb = yb * sc0;
#pragma unroll
                for (int k = 0; k < EMF_LEVELS; ++k) {
                    acc[2 + k] += w * floor(va + 0.5);
                    acc[EMF_VALS + 2 + k] += w * floor(vb + 0.5);
                    va *= 0.5; vb *= 0.5;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * EMF_VALS; ++i) q.red[i] = acc[i];
    });
    ex.template block_sum<2 * EMF_VALS>(st, [&](const double* tot) {
        double* o = p.partials + (size_t)ex.bid() * (2 * EMF_VALS);
        for (int i = 0; i < 2 * EMF_VALS; ++i) o[i] = tot[i];
    });
}
// torch.norm / exact norm of a gathered class vector of `cnt` elements from its sampled statistics v[EMF_VALS]
SM_HD double emf_norm_ratio(const double* v, int elo, double cnt) {
    const double cnt_s = v[0], s_s = v[1];
    const double N = floor(cnt / 8.0);
    if (!(cnt_s > 0) || !(s_s > 0) || !(N >= 1)) return 1.0;
    const double mean_y = s_s / cnt_s;
    unsigned long long mb; memcpy(&mb, &mean_y, 8);
    int e = (int)((mb >> 52) & 0x7ffu) - 1023 - 2;
    double S = 0.0, left = N;
    for (int guard = 0; guard < 400 && left > 0; ++guard) {
        const double hi = emf_pow2(e + 1);
        if (S >= hi) { ++e; continue; }
        const int k = e - elo;
        double g = (k >= 0 && k < EMF_LEVELS) ? v[2 + k] * emf_pow2(e - 23) / cnt_s : mean_y;
        if (!(g > 0)) break;                     // every square rounds to nothing: the sum stalls here
        const double need = (hi - S) / g;
        if (need >= left) { S += left * g; left = 0; }
        else { S = hi; left -= need; ++e; }
    }
    const double r = S / (N * mean_y);
    return r > 0 ? sqrt(r) : 1.0;
}

// one work-group: sum the partials (fixed order per thread, then the block
// reduction) and derive the constants (reference functions.py:36-43 on the
// gathered slerp-class vectors)
struct SlerpConstsState { double red[2 * EMF_VALS]; };
template <class Ex>
SM_HD void k_slerp_consts(Ex& ex, const SlerpConstParams& p) {
    typename Ex::template State<SlerpConstsState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const bool use_fb = p.overflow && *p.overflow;
    const double* src = use_fb ? p.fallback : p.partials;
    const int nsrc = use_fb ? p.nfallback : p.nparts;
    double* emf_tot = (double*)(ex.lds() + 2 * LDS_SCRATCH_FLOATS);       // [2 * EMF_VALS] (the 36-value block sum's scratch
                                                                          // runs past LDS_SCRATCH_FLOATS)
    if (p.emf_part) {
        ex.each(st, [&](int tid, SlerpConstsState& s) {
            double a[2 * EMF_VALS];
#pragma unroll
            for (int q = 0; q < 2 * EMF_VALS; ++q) a[q] = 0.0;
            for (int i = tid; i < p.emf_nparts; i += nt) {
#pragma unroll
                for (int q = 0; q < 2 * EMF_VALS; ++q) a[q] += p.emf_part[(size_t)i * (2 * EMF_VALS) + q];
            }
#pragma unroll
            for (int q = 0; q < 2 * EMF_VALS; ++q) s.red[q] = a[q];
        });
        ex.template block_sum<2 * EMF_VALS>(st, [&](const double* tot) {
            for (int q = 0; q < 2 * EMF_VALS; ++q) emf_tot[q] = tot[q];
        });
    }
    ex.each(st, [&](int tid, SlerpConstsState& s) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int i = tid; i < nsrc; i += nt) {
            a0 += src[4 * i]; a1 += src[4 * i + 1];
            a2 += src[4 * i + 2]; a3 += src[4 * i + 3];
        }
        if (p.zero_u32 && tid < p.zero_u32_count) p.zero_u32[tid] = 0u;
        s.red[0] = a0; s.red[1] = a1; s.red[2] = a2; s.red[3] = a3;
    });
    ex.template block_sum<4>(st, [&](const double* tot) {
        const double s00 = tot[0], s01 = tot[1], s11 = tot[2], cnt = tot[3];
        BlendConsts c;
        c.thr = p.thr ? *p.thr : 0.f;
        c.s00 = s00; c.s01 = s01; c.s11 = s11; c.n_slerp = (unsigned long long)cnt;
        double n0 = sqrt(s00), n1 = sqrt(s11), rel_bias = 1.0;
        if (s00 > 0 && s11 > 0) {
            // the reference's cosine divides by ITS norms; F.normalize's norm of (v1 - dot v0) carries about v1's bias
            if (p.ref_norms) {
                rel_bias = (double)p.ref_norms[1] / n1;
                n0 = (double)p.ref_norms[0]; n1 = (double)p.ref_norms[1];
            } else if (p.emf_part) {
                rel_bias = emf_norm_ratio(emf_tot + EMF_VALS, p.emf_elo, cnt);
                n0 *= emf_norm_ratio(emf_tot, p.emf_elo, cnt);
                n1 *= rel_bias;
            }
        }
        double dot = s01 / (n0 * n1);
        if (dot > 1.0) dot = 1.0;
        if (dot < -1.0) dot = -1.0;
        const float dotf = (float)dot;
        const float theta = acosf(dotf) * p.t;
        double rel2 = s11 - 2.0 * (double)dotf * s01 + (double)dotf * (double)dotf * s00;
        if (rel2 < 0) rel2 = 0;
        double reln = sqrt(rel2) * rel_bias;
        if (reln < 1e-12) reln = 1e-12;
        c.dot = dotf; c.cos_t = cosf(theta); c.sin_t = sinf(theta); c.inv_rel = (float)(1.0 / reln);
        c.pad[0] = c.pad[1] = c.pad[2] = 0.f;
        *p.out = c;
    });
}

// sum [n][2] double partials into out[2] (norm^2 of the two signals of F1 / combine)
// Results the host waits for are written by the last small kernel straight into a host-mapped
// "mailbox" (pinned, device-visible): a stream synchronisation replaces the copy kernels and
// staging of hipMemcpyAsync into pageable memory.
struct Mailbox {
    double norm2[16];           // k_sum_partials (2) / k_delta_norms' reduction (up to 16 models)
    uint32_t flags[12];         // NaN/Inf flags [0..7] + candidate counters [8..11] (k_publish)
    float thr[4];               // d_thr(0..3)
    BlendConsts consts;
    float snorm[16];            // k_serial_norm results (norm_mode = reference_cpu)
    float corr[64];             // k_corr_finish: the K x K matrix of correlate_pairs
};
struct PublishParams {
    const uint32_t* flags;      // device flags + counters (12 words)
    const float* thr;
    const BlendConsts* consts;
    Mailbox* mail;
    uint32_t* zero_flags;       // when set: the 12 flag words are cleared after the copy
};
template <class Ex>
SM_HD void k_publish(Ex& ex, const PublishParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    ex.each(st, [&](int tid, EmptyState&) {
        if (tid < 12) {
            p.mail->flags[tid] = p.flags[tid];
            if (p.zero_flags) p.zero_flags[tid] = 0u;
        }
        if (tid < 4) p.mail->thr[tid] = p.thr[tid];
        if (tid == 0) p.mail->consts = *p.consts;
    });
}

struct SumPartialsParams { const double* partials; int nparts; double* out; };
template <class Ex>
SM_HD void k_sum_partials(Ex& ex, const SumPartialsParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState& s) {
        double a0 = 0, a1 = 0;
        for (int i = tid; i < p.nparts; i += nt) { a0 += p.partials[2 * i]; a1 += p.partials[2 * i + 1]; }
        s.red[0] = a0; s.red[1] = a1;
    });
    ex.template block_sum<2>(st, [&](const double* tot) { p.out[0] = tot[0]; p.out[1] = tot[1]; });
}

enum { BLEND_SLERP = 0, BLEND_ARITH = 1 };
struct BlendParams {
    const float* reA; const float* reB;
    float* reR;
    int R, C, Cb;
    int vec4;
    int mode;
    int agreement;              // arithmetic branch: sign agreement on/off
    float t;                    // arithmetic: R = ra + t*rb
    float t_sum;
    const BlendConsts* consts;  // slerp mode
    unsigned long long* hist;   // level-1 histogram of |Re R| for the cull, or null
    int chunks;
};

SM_HD float blend_one(const BlendParams& p, const BlendConsts& c, float a, float b) {
    if (p.mode == BLEND_SLERP) {
        if (same_sign(a, b)) {
            if (fabsf(b) < c.thr) return a + p.t_sum * b;
            return a * c.cos_t + ((b - a * c.dot) * c.inv_rel) * c.sin_t;
        }
        return (fabsf(a) > fabsf(b)) ? a : b;
    }
    const bool agree = p.agreement ? same_sign(a, b) : true;
    return agree ? (a + p.t * b) : b;                 // quirk Q3: disagreeing bins take b
}

template <class Ex>
SM_HD void k_blend(Ex& ex, const BlendParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    uint32_t* lh = (uint32_t*)(ex.lds() + LDS_SCRATCH_FLOATS);
    const int nt = ex.nthreads();
    const size_t total = (size_t)p.Cb * p.R;
    const size_t nquad = (total + 3) / 4;
    BlendConsts c;
    memset(&c, 0, sizeof(c));
    if (p.mode == BLEND_SLERP) c = *p.consts;
    const WeightRanges wr = weight_ranges(p.R, p.C, p.Cb);
    if (p.hist) {
        ex.each(st, [&](int tid, EmptyState&) { for (int b = tid; b < HIST1_BINS; b += nt) lh[b] = 0; });
        ex.sync();
    }
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        auto quad = [&](size_t i0, const float* a, const float* b, int n, bool vec) {
            float r[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = blend_one(p, c, a[e], b[e]);
            if (vec) {
                cf4 v = {r[0], r[1], r[2], r[3]};
                *(cf4*)(p.reR + i0) = v;
            } else {
                for (int e = 0; e < n; ++e) p.reR[i0 + e] = r[e];
            }
            if (p.hist) {
                const uint32_t w0 = weight_at(wr, i0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e < n) {
                        const uint32_t key = f2u(r[e]) & 0x7fffffffu;
                        ex.lds_atomic_add(&lh[key >> 20], (vec && !wr.full) ? w0 : weight_at(wr, i0 + e));
                    }
                }
            }
        };
        if (p.vec4) {
            // 16-byte loads of U steps in flight together (clamped, not branched around)
            constexpr int U = 4;
            const cf4* A4 = (const cf4*)p.reA;
            const cf4* B4 = (const cf4*)p.reB;
            for (int q0 = 0; q0 < p.chunks; q0 += U) {
                cf4 av[U], bv[U];
                size_t qv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    qv[u] = start + (size_t)(q0 + u) * nt + tid;
                    av[u] = A4[qv[u] < nquad ? qv[u] : nquad - 1];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) bv[u] = B4[qv[u] < nquad ? qv[u] : nquad - 1];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (q0 + u < p.chunks && qv[u] < nquad) {
                        const float a[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
                        const float b[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
                        quad(4 * qv[u], a, b, 4, true);
                    }
                }
            }
        } else {
            for (int q = 0; q < p.chunks; ++q) {
                const size_t qi = start + (size_t)q * nt + tid;
                if (qi >= nquad) break;
                float a[4], b[4];
                const int n = load_quad(p.reA, 4 * qi, total, 0, a);
                load_quad(p.reB, 4 * qi, total, 0, b);
                quad(4 * qi, a, b, n, false);
            }
        }
    });
    if (p.hist) {
        ex.sync();
        ex.each(st, [&](int tid, EmptyState&) {
            for (int b = tid; b < HIST1_BINS; b += nt) {
                const uint32_t v = lh[b];
                if (v) ex.global_atomic_add(&p.hist[b], (unsigned long long)v);
            }
        });
    }
}

// =====================================================================
// spectral intermediates (tournament rounds >= 2, DESIGN.md "K >= 3")
// =====================================================================
// A pair merge that is not the last leaves its result in the spectral domain: (Re R, Im a) with
// the cull still to be applied.  k_spec_norm takes || post * ifft(R) ||^2 * n / post^2 =
// sum over the FULL spectrum of |R_culled|^2 (Parseval), i.e. what the reference's torch.norm of
// the materialised intermediate measures (fast_fourier.py:209-210).
struct SpecNormParams {
    const float* re; const float* im;
    int R, C, Cb;
    int vec4;
    const float* thr;           // device scalar (cull threshold) or null
    double* partials;           // [grid][2]: sum w*re_c^2, sum w*im^2
    int chunks;
};
template <class Ex>
SM_HD void k_spec_norm(Ex& ex, const SpecNormParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t total = (size_t)p.Cb * p.R;
    const size_t nquad = (total + 3) / 4;
    const float thr = p.thr ? *p.thr : 0.f;
    const WeightRanges wr = weight_ranges(p.R, p.C, p.Cb);
    ex.each(st, [&](int tid, EmptyState& s) {
        double sr = 0, si = 0;
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int c = 0; c < p.chunks; ++c) {
            const size_t qi = start + (size_t)c * nt + tid;
            if (qi >= nquad) break;
            const size_t i0 = 4 * qi;
            float a[4], b[4];
            const int n = load_quad(p.re, i0, total, p.vec4, a);
            load_quad(p.im, i0, total, p.vec4, b);
            float qr = 0.f, qi2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e < n) {
                    const float w = (float)weight_at(wr, i0 + e);
                    if (!(fabsf(a[e]) < thr)) qr += w * a[e] * a[e];
                    qi2 += w * b[e] * b[e];
                }
            }
            sr += qr; si += qi2;
        }
        s.red[0] = sr; s.red[1] = si;
    });
    ex.template block_sum<2>(st, [&](const double* tot) {
        p.partials[2 * (size_t)ex.bid()] = tot[0];
        p.partials[2 * (size_t)ex.bid() + 1] = tot[1];
    });
}

// Rounding-noise model.  In the reference an intermediate goes through ifft -> fft before the next
// round uses it, so the bins its cull zeroed come back as rounding noise (measured against the
// real reference, oracle/chaos_probe.py: Gaussian, sigma = 1.1e-7 .. 1.4e-7 of the rms bin of the
// unit-norm spectrum), and the next round takes its sign-agreement decisions and its 8 % quantile
// ON that noise.  Exact zeros there would change those decisions systematically (sign(0) = 0 never
// agrees; the quantile becomes 0), so the noise is modelled: a counter-based hash of
// (seed, element index) -> Box-Muller.  Only the statistics are reproducible by anyone - the
// reference does not reproduce its own noise across FFT libraries.
SM_HD uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
SM_HD float model_noise(uint32_t seed, size_t i, float sigma) {
    const uint32_t h1 = hash_u32((uint32_t)i * 2u + 1u + seed * 0x9e3779b9u) ^ hash_u32((uint32_t)(i >> 31) + seed);
    const uint32_t h2 = hash_u32(h1 + 0x68bc21ebu);
    // 8 % of the reference's noise values are exact zeros (fp32 cancellation; sign(0) = 0 is its own
    // sign class there): measured 0.094 / 0.081 / 0.076 at 256^2 / 1024^2 / 4096^2
    if ((h2 >> 24) < 21u) return 0.f;
    // sum of four uniforms (variance 1/3): Gaussian enough - only the sign and the rank of a noise
    // value among the other noise values ever matter - and no transcendental in a streaming kernel
    const float s4 = (float)(h1 & 0xffffu) + (float)(h1 >> 16) + (float)(h2 & 0xffffu) + (float)(hash_u32(h2) >> 16);
    return sigma * 1.7320508f * (s4 * (1.0f / 65536.0f) - 2.0f);
}

struct SpecRescaleParams {
    const float* re; const float* im;   // the intermediate's planes (im may be null: role b)
    float* dre; float* dim;             // destination planes (may alias the sources)
    int R, C, Cb;
    int vec4;
    float thr;                          // cull threshold (value), applied on read
    float scale;                        // 1 / sqrt(sum |R_c|^2 / n): unit spatial norm
    float sigma; uint32_t seed;         // noise model for the culled bins
    unsigned long long* hist;           // level-1 histogram of |dre| or null
    int chunks;
    double* im_partials;                // role a (im != null) or null: [grid] sum w (Im written)^2
};
template <class Ex>
SM_HD void k_spec_rescale(Ex& ex, const SpecRescaleParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    uint32_t* lh = (uint32_t*)(ex.lds() + LDS_SCRATCH_FLOATS);
    const int nt = ex.nthreads();
    const size_t total = (size_t)p.Cb * p.R;
    const size_t nquad = (total + 3) / 4;
    const WeightRanges wr = weight_ranges(p.R, p.C, p.Cb);
    if (p.hist) {
        ex.each(st, [&](int tid, EmptyState&) { for (int b = tid; b < HIST1_BINS; b += nt) lh[b] = 0; });
        ex.sync();
    }
    ex.each(st, [&](int tid, EmptyState& s) {
        double imacc = 0.0;
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int c = 0; c < p.chunks; ++c) {
            const size_t qi = start + (size_t)c * nt + tid;
            if (qi >= nquad) break;
            const size_t i0 = 4 * qi;
            float a[4], b[4] = {0.f, 0.f, 0.f, 0.f};
            const int n = load_quad(p.re, i0, total, p.vec4, a);
            if (p.im) load_quad(p.im, i0, total, p.vec4, b);
            float r[4], im[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[e] = (fabsf(a[e]) < p.thr) ? model_noise(p.seed, i0 + e, p.sigma) : a[e] * p.scale;
                im[e] = b[e] * p.scale;
            }
            if (p.vec4) {
                cf4 v = {r[0], r[1], r[2], r[3]};
                *(cf4*)(p.dre + i0) = v;
                if (p.im) { cf4 vi = {im[0], im[1], im[2], im[3]}; *(cf4*)(p.dim + i0) = vi; }
            } else {
                for (int e = 0; e < n; ++e) { p.dre[i0 + e] = r[e]; if (p.im) p.dim[i0 + e] = im[e]; }
            }
            if (p.hist) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < n) ex.lds_atomic_add(&lh[(f2u(r[e]) & 0x7fffffffu) >> 20], weight_at(wr, i0 + e));
            }
            if (p.im_partials) {
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < n) q += (float)weight_at(wr, i0 + e) * im[e] * im[e];
                imacc += q;
            }
        }
        s.red[0] = imacc;
    });
    if (p.hist) {
        ex.sync();
        ex.each(st, [&](int tid, EmptyState&) {
            for (int b = tid; b < HIST1_BINS; b += nt) {
                const uint32_t v = lh[b];
                if (v) ex.global_atomic_add(&p.hist[b], (unsigned long long)v);
            }
        });
    }
    if (p.im_partials) {
        ex.sync();
        ex.template block_sum<1>(st, [&](const double* tot) { p.im_partials[ex.bid()] = tot[0]; });
    }
}

// =====================================================================
// elementwise spatial kernels (deltas, norms, add branch, K=1 finish)
// =====================================================================
// ||finetune_i - base_i||^2 of every model in ONE pass (K >= 3 needs all norms before the
// first pairing): a base tensor shared by several models is loaded once per octet.
// 16-bit inputs, n % 8 == 0, 16-byte aligned pointers (the host falls back to k_combine otherwise).
constexpr int NORMS_MAX = 16;       // row length of the partials / mailbox
constexpr int NORMS_K = 8;          // models one launch handles
struct DeltaNormsParams {
    int k;                           // <= NORMS_K
    const void* ft[NORMS_K];
    const void* ubase[2];            // the (at most two) distinct base tensors
    int base_of[NORMS_K];            // model -> 0 / 1
    int dtype;
    size_t n;
    int chunks;                      // octets per thread
    double* partials;                // [grid][NORMS_MAX]
};
template <class Ex>
SM_HD void k_delta_norms(Ex& ex, const DeltaNormsParams& p) {
    typename Ex::template State<NormsState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t noct = p.n / 8;
    const u32x4* b0 = (const u32x4*)p.ubase[0];
    const u32x4* b1 = (const u32x4*)(p.ubase[1] ? p.ubase[1] : p.ubase[0]);
    ex.each(st, [&](int tid, NormsState& s) {
        double acc[NORMS_K];
#pragma unroll
        for (int i = 0; i < NORMS_K; ++i) acc[i] = 0.0;
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int q = 0; q < p.chunks; ++q) {
            const size_t oi = start + (size_t)q * nt + tid;
            if (oi >= noct) break;
            u32x4 rf[NORMS_K];
            const u32x4 r0 = b0[oi], r1 = b1[oi];
#pragma unroll
            for (int i = 0; i < NORMS_K; ++i) rf[i] = ((const u32x4*)p.ft[i < p.k ? i : 0])[oi];
            float v0[8], v1[8];
            decode16x8(r0, p.dtype, v0);
            decode16x8(r1, p.dtype, v1);
#pragma unroll
            for (int i = 0; i < NORMS_K; ++i) {
                float vf[8];
                decode16x8(rf[i], p.dtype, vf);
                const bool second = p.base_of[i] != 0;
                float ps = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = vf[e] - (second ? v1[e] : v0[e]);
                    ps += d * d;
                }
                if (i < p.k) acc[i] += ps;
            }
        }
#pragma unroll
        for (int i = 0; i < NORMS_MAX; ++i) s.red[i] = i < NORMS_K ? acc[i < NORMS_K ? i : 0] : 0.0;
    });
    ex.template block_sum<NORMS_MAX>(st, [&](const double* tot) {
        for (int i = 0; i < NORMS_MAX; ++i) p.partials[(size_t)ex.bid() * NORMS_MAX + i] = tot[i];
    });
}
// ---------------------------------------------------------------------------------
// norm_mode = reference_cpu: ||x||_2 exactly as torch.norm computes it on CPU for a contiguous
// fp32 tensor - acc = fma(x, x, acc) SERIALLY in 8 fp32 lanes (element i goes to lane i % 8),
// the lanes then added in order, sqrt.  That sum loses low bits once the running sum is large
// (-7e-4 at 16 M elements, -5e-3 at 67 M: oracle/norm_bias_probe.py), and the reference's
// pick-the-larger decisions (|ra| / ||a||  vs  |rb| / ||b||) follow the BIASED norms: an
// implementation with accurate norms differs from the reference's device=cpu output by 2e-2 on
// the merged delta at 8192^2 (profiles/parity_fullsize.json).  The chain is inherently sequential
// (n/8 dependent adds per lane: ~20 ms for 8192^2), so this mode is opt-in.
// One work-group per signal; wave 0's lanes 0..7 hold the accumulators, all waves stage squares
// of (x - base) through LDS, double-buffered.
// ---------------------------------------------------------------------------------
constexpr int SER_CHUNK = 4096;             // elements per staged chunk (512 rows of 8)
struct SerialNormParams {
    int k;
    SigDesc sig[16];
    size_t n;                               // n % 8 == 0
    float* out;                             // [k]: the fp32 norm torch would return
};
template <class Ex>
SM_HD void k_serial_norm(Ex& ex, const SerialNormParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    float* buf = ex.lds() + LDS_SCRATCH_FLOATS;          // two buffers of SER_CHUNK floats
    const int nt = ex.nthreads();
    const SigDesc sg = p.sig[ex.bid()];
    const size_t nchunks = (p.n + SER_CHUNK - 1) / SER_CHUNK;
    auto stage = [&](size_t c, int which) {
        ex.each(st, [&](int tid, EmptyState&) {
            float* dst = buf + which * SER_CHUNK;
            for (int o = tid; o < SER_CHUNK / 8; o += nt) {
                const size_t i0 = c * SER_CHUNK + (size_t)o * 8;
                float v[8];
                if (i0 < p.n) load_sig8(sg, i0, v);
                else { for (int e = 0; e < 8; ++e) v[e] = 0.f; }
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[o * 8 + e] = v[e];               // acc = fma(x, x, acc): see sm_aten_norm.hpp
            }
        });
    };
    ex.each(st, [&](int, EmptyState& s) { s.red[0] = 0.0; });
    if (nchunks) stage(0, 0);
    ex.sync();
    for (size_t c = 0; c < nchunks; ++c) {
        const int cur = (int)(c & 1);
        if (c + 1 < nchunks) stage(c + 1, cur ^ 1);
        ex.each(st, [&](int tid, EmptyState& s) {
            if (tid < 8) {
                float acc = (float)s.red[0];
                const float* src = buf + cur * SER_CHUNK + tid;
                const size_t left = p.n - c * SER_CHUNK;
                const int rows = (int)((left < (size_t)SER_CHUNK ? left : (size_t)SER_CHUNK) / 8);
                int r = 0;
                for (; r + 16 <= rows; r += 16) {
                    float x[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) x[q] = src[(r + q) * 8];
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc = fmaf(x[q], x[q], acc);
                }
                for (; r < rows; ++r) acc = fmaf(src[r * 8], src[r * 8], acc);
                s.red[0] = (double)acc;
            }
        });
        ex.sync();
    }
    // lanes added in order, then the square root (fp32)
    float* lanes = buf;
    ex.each(st, [&](int tid, EmptyState& s) { if (tid < 8) lanes[tid] = (float)s.red[0]; });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        if (tid == 0) {
            float tot = 0.f;
            for (int q = 0; q < 8; ++q) tot = tot + lanes[q];
            p.out[ex.bid()] = sqrtf(tot);
        }
    });
}

struct SumNParams { const double* partials; int nparts; double* out; };
template <class Ex>
SM_HD void k_sum_partials_n(Ex& ex, const SumNParams& p) {
    typename Ex::template State<NormsState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, NormsState& s) {
        double a[NORMS_MAX];
#pragma unroll
        for (int i = 0; i < NORMS_MAX; ++i) a[i] = 0.0;
        for (int r = tid; r < p.nparts; r += nt) {
#pragma unroll
            for (int i = 0; i < NORMS_MAX; ++i) a[i] += p.partials[(size_t)r * NORMS_MAX + i];
        }
#pragma unroll
        for (int i = 0; i < NORMS_MAX; ++i) s.red[i] = a[i];
    });
    ex.template block_sum<NORMS_MAX>(st, [&](const double* tot) {
        for (int i = 0; i < NORMS_MAX; ++i) p.out[i] = tot[i];
    });
}

struct CombineParams {
    SigDesc a, b;               // b.x may be null
    float ca, cb;               // out = ca*a + cb*b
    size_t n;
    int vec8;                   // n % 8 == 0 and every pointer 16-byte aligned
    float* out_f32;             // or null (norms only)
    // optional finish: add base, NaN/Inf policy, bf16 / f32 store
    const void* base; int base_dtype;
    void* out_final; int out_mode;
    uint32_t* flags;
    double* partials;           // [grid][2] sum a^2, sum b^2 (of the un-weighted signals) or null
    int chunks;                 // octets per thread
};

template <class Ex>
SM_HD void k_combine(Ex& ex, const CombineParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t noct = (p.n + 7) / 8;
    ex.each(st, [&](int tid, EmptyState& s) {
        double sa = 0, sb = 0;
        uint32_t nan2 = 0, inf2 = 0;
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        // 16-bit inputs, no finishing stage (the norms of the raw deltas, K >= 3): the loads of
        // U octets go out together, clamped instead of branched around, and are decoded afterwards
        const bool fast16 = p.vec8 && !p.out_final && p.a.x && p.a.dtype != DT_F32 && (!p.b.x || p.b.dtype != DT_F32);
        if (fast16) {
            constexpr int U = 4;
            const bool has_ab = p.a.base != nullptr, has_b = p.b.x != nullptr, has_bb = has_b && p.b.base != nullptr;
            const u32x4* pa = (const u32x4*)p.a.x;
            const u32x4* pab = has_ab ? (const u32x4*)p.a.base : pa;
            const u32x4* pb = has_b ? (const u32x4*)p.b.x : pa;
            const u32x4* pbb = has_bb ? (const u32x4*)p.b.base : pa;
            for (int q0 = 0; q0 < p.chunks; q0 += U) {
                u32x4 ra[U], rab[U], rb[U], rbb[U];
                size_t oc[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const size_t oi = start + (size_t)(q0 + u) * nt + tid;
                    ok[u] = (q0 + u) < p.chunks && oi < noct;
                    oc[u] = ok[u] ? oi : 0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) ra[u] = pa[oc[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) rab[u] = pab[oc[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) rb[u] = pb[oc[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) rbb[u] = pbb[oc[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!ok[u]) continue;
                    float va[8], ba[8], vb[8], bb[8], o[8];
                    decode16x8(ra[u], p.a.dtype, va); decode16x8(rab[u], p.a.dtype, ba);
                    decode16x8(rb[u], p.b.dtype, vb); decode16x8(rbb[u], p.b.dtype, bb);
                    float pa2 = 0.f, pb2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xa = (va[e] - (has_ab ? ba[e] : 0.f)) * p.a.prescale;
                        const float xb = has_b ? (vb[e] - (has_bb ? bb[e] : 0.f)) * p.b.prescale : 0.f;
                        pa2 += xa * xa; pb2 += xb * xb;
                        o[e] = p.ca * xa + (has_b ? p.cb * xb : 0.f);
                    }
                    sa += pa2; sb += pb2;
                    if (p.out_f32) {
                        cf4 w0 = {o[0], o[1], o[2], o[3]}, w1 = {o[4], o[5], o[6], o[7]};
                        ((cf4*)p.out_f32)[2 * oc[u]] = w0; ((cf4*)p.out_f32)[2 * oc[u] + 1] = w1;
                    }
                }
            }
        }
        for (int q = 0; q < (fast16 ? 0 : p.chunks); ++q) {
            const size_t oi = start + (size_t)q * nt + tid;
            if (oi >= noct) break;
            const size_t i0 = 8 * oi;
            float a[8], b[8], bs[8], o[8];
            int cnt = 8;
            if (p.vec8) {
                load_sig8(p.a, i0, a);
                load_sig8(p.b, i0, b);
                if (p.out_final && p.base) load_elem8(p.base, p.base_dtype, i0, bs);
            } else {
                cnt = (int)((p.n - i0) < 8 ? (p.n - i0) : 8);
                for (int e = 0; e < 8; ++e) {
                    a[e] = b[e] = bs[e] = 0.f;
                    if (e < cnt) {
                        a[e] = load_sig1(p.a, i0 + e); b[e] = load_sig1(p.b, i0 + e);
                        if (p.out_final && p.base) bs[e] = load_elem(p.base, p.base_dtype, i0 + e);
                    }
                }
            }
            float pa = 0.f, pb = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pa += a[e] * a[e]; pb += b[e] * b[e];
                float v = p.ca * a[e] + (p.b.x ? p.cb * b[e] : 0.f);
                o[e] = v;
            }
            sa += pa; sb += pb;
            if (p.out_f32) {
                if (p.vec8) {
                    cf4 w0 = {o[0], o[1], o[2], o[3]}, w1 = {o[4], o[5], o[6], o[7]};
                    ((cf4*)p.out_f32)[i0 / 4] = w0; ((cf4*)p.out_f32)[i0 / 4 + 1] = w1;
                } else {
                    for (int e = 0; e < cnt; ++e) p.out_f32[i0 + e] = o[e];
                }
            }
            if (p.out_final) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = o[e];
                    if (p.base) v += bs[e];
                    if (e < cnt) {
                        if (is_nan(v)) { v = 0.f; nan2++; }
                        if (is_inf(v)) inf2 = 1;
                    }
                    o[e] = v;
                }
                if (p.vec8 && p.out_mode == OUT_BF16) {
                    u32x4 w;
                    w.x = (uint32_t)f_to_bf16(o[0]) | ((uint32_t)f_to_bf16(o[1]) << 16);
                    w.y = (uint32_t)f_to_bf16(o[2]) | ((uint32_t)f_to_bf16(o[3]) << 16);
                    w.z = (uint32_t)f_to_bf16(o[4]) | ((uint32_t)f_to_bf16(o[5]) << 16);
                    w.w = (uint32_t)f_to_bf16(o[6]) | ((uint32_t)f_to_bf16(o[7]) << 16);
                    ((u32x4*)p.out_final)[oi] = w;
                } else {
                    for (int e = 0; e < cnt; ++e) {
                        if (p.out_mode == OUT_BF16) ((uint16_t*)p.out_final)[i0 + e] = f_to_bf16(o[e]);
                        else ((float*)p.out_final)[i0 + e] = o[e];
                    }
                }
            }
        }
        s.red[0] = sa; s.red[1] = sb;
        if (nan2) ex.global_atomic_add_u32(&p.flags[2], nan2);
        if (inf2) ex.global_atomic_or_u32(&p.flags[3], 1u);
    });
    if (p.partials) {
        ex.template block_sum<2>(st, [&](const double* tot) {
            p.partials[2 * (size_t)ex.bid()] = tot[0];
            p.partials[2 * (size_t)ex.bid() + 1] = tot[1];
        });
    }
}

// =====================================================================
// N3: AdditionMerge / TaskAdditionMerge (reference shard/merge/addition.py:70-76,
// taskaddition.py:69-79) - sums of finetune deltas relative to output_base_model's tensor,
// in the TENSORS' dtype as torch computes it on CPU: a 16-bit elementwise op is an fp32 op
// rounded to the dtype, a 16-bit sum over the stacked dim accumulates in fp32 and rounds once.
//   mode 0 (AdditionMerge):      out = 0; for each ft: out = rnd(out + rnd(ft - base))
//   mode 1 (TaskAdditionMerge):  d_i = rnd(ft_i - base); s = sign(sum_i sign(d_i));
//                                out = rnd(sum_i d_i * [sign(d_i) == s])
// Neither adds the base back (the reference does not).
// =====================================================================
SM_HD uint16_t f_to_bf16_any(float v) {            // RNE, NaN stays NaN
    const uint32_t u = f2u(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    return f_to_bf16(v);
}
SM_HD uint16_t f_to_f16_any(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const _Float16 h = (_Float16)v;                   // v_cvt_f16_f32 (RNE)
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
#else
    const uint32_t u = f2u(v), sign = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (a >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);              // >= 65536: Inf (65520 rounds up below)
    if (a < 0x33000000u) return (uint16_t)sign;                             // < 2^-25: zero
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift = (e < -14) ? (13 + (-14 - e)) : 13;                         // subnormal halves lose more bits
    uint32_t half = m >> shift, rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    uint32_t out = (e < -14) ? half : (((uint32_t)(e + 15) << 10) + (half - 0x400u));
    return (uint16_t)(sign | out);
#endif
}
SM_HD float round_to_dtype(float v, int dtype) {
    if (dtype == DT_F32) return v;
    if (dtype == DT_BF16) return bf16_to_f(f_to_bf16_any(v));
    return f16_to_f(f_to_f16_any(v));
}
SM_HD float sign_of(float v) { return is_nan(v) ? v : (float)sgn(v); }      // torch.sign(NaN) = NaN

// =====================================================================
// A1 / A8 at function level (not on the merge's hot path - inside it both are fused into the kernels above):
// slerp(v0, v1, t) (reference shard/tensor/functions.py:24-43) and tensor / norm (functions.py:75-88).
//   dot   = clamp(sum(v0 v1) / (||v0|| ||v1||), -1, 1)        (quirk Q5: the cosine of the UN-normalised vectors)
//   theta = acos(dot) t;  rel = v1 - v0 dot;  rel /= max(||rel||_2 along the LAST dim, 1e-12)   (F.normalize(dim=-1))
//   out   = v0 cos(theta) + rel sin(theta)
// Sums are taken in fp64 and rounded once (torch sums in fp32: the two agree to a few 1e-8), the element-wise
// part is torch's sequence of fp32 operations.
// =====================================================================
struct FnSumsParams { const float* v0; const float* v1; size_t n; int chunks; double* partials; };   // [grid][4]
template <class Ex>
SM_HD void k_fn_sums(Ex& ex, const FnSumsParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState& s) {
        double s01 = 0, s00 = 0, s11 = 0;
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int c = 0; c < p.chunks; ++c) {
            const size_t i = start + (size_t)c * nt + tid;
            if (i >= p.n) break;
            const double a = (double)p.v0[i], b = (double)p.v1[i];
            s01 += a * b; s00 += a * a; s11 += b * b;
        }
        s.red[0] = s01; s.red[1] = s00; s.red[2] = s11; s.red[3] = 0.0;
    });
    ex.template block_sum<4>(st, [&](const double* tot) {
        for (int q = 0; q < 4; ++q) p.partials[4 * (size_t)ex.bid() + q] = tot[q];
    });
}
SM_HD float fn_acosf(float x) { return acosf(x); }
struct FnSlerpFinParams { const double* partials; int nparts; float t; float* consts; };    // consts: dot, cos(theta), sin(theta), -
template <class Ex>
SM_HD void k_fn_slerp_fin(Ex& ex, const FnSlerpFinParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState& s) {
        double a = 0, b = 0, c = 0;
        for (int i = tid; i < p.nparts; i += nt) { a += p.partials[4 * i]; b += p.partials[4 * i + 1]; c += p.partials[4 * i + 2]; }
        s.red[0] = a; s.red[1] = b; s.red[2] = c; s.red[3] = 0.0;
    });
    ex.template block_sum<4>(st, [&](const double* tot) {
        // (0 / 0 = NaN for a zero vector, as in the reference: clamp keeps NaN)
        float dot = (float)tot[0] / ((float)sqrt(tot[1]) * (float)sqrt(tot[2]));
        if (dot < -1.f) dot = -1.f;
        if (dot > 1.f) dot = 1.f;
        const float theta = fn_acosf(dot) * p.t;
        p.consts[0] = dot; p.consts[1] = cosf(theta); p.consts[2] = sinf(theta); p.consts[3] = 0.f;
    });
}
// one work-group per (row, column segment).  PHASE 0: sum of rel^2 of the segment -> part[row * cchunks + seg];
// PHASE 1: the outputs
struct FnSlerpRowsParams {
    const float* v0; const float* v1; float* out;
    size_t rows, cols;
    int cchunks, seg;               // segments per row, elements per segment
    const float* consts;
    double* part;                   // [rows * cchunks]
    const float* den;               // [rows]: max(||rel row||, 1e-12)
};
template <int PHASE, class Ex>
SM_HD void k_fn_slerp_rows(Ex& ex, const FnSlerpRowsParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t row = (size_t)ex.bid() / p.cchunks;
    const int sg = (int)((size_t)ex.bid() % p.cchunks);
    const float dot = p.consts[0], ct = p.consts[1], sn = p.consts[2];
    const size_t c0 = (size_t)sg * p.seg, c1 = (c0 + p.seg < p.cols) ? c0 + p.seg : p.cols;
    const float dn = PHASE == 1 ? p.den[row] : 1.f;
    ex.each(st, [&](int tid, EmptyState& s) {
        double acc = 0.0;
        for (size_t c = c0 + tid; c < c1; c += nt) {
            const size_t i = row * p.cols + c;
            const float a = p.v0[i];
            const float rel = aten_fadd_(p.v1[i], -aten_fmul_(a, dot));
            if (PHASE == 0) acc += (double)rel * (double)rel;
            else p.out[i] = aten_fadd_(aten_fmul_(a, ct), aten_fmul_(rel / dn, sn));
        }
        s.red[0] = acc;
    });
    if (PHASE == 0) ex.template block_sum<1>(st, [&](const double* tot) { p.part[row * p.cchunks + sg] = tot[0]; });
}
struct FnSlerpDenParams { const double* part; size_t rows; int cchunks; float* den; };
template <class Ex>
SM_HD void k_fn_slerp_den(Ex& ex, const FnSlerpDenParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t row = (size_t)ex.bid() * ex.nthreads() + tid;
        if (row >= p.rows) return;
        double a = 0.0;
        for (int c = 0; c < p.cchunks; ++c) a += p.part[row * p.cchunks + c];
        const float nrm = (float)sqrt(a);
        p.den[row] = nrm > 1e-12f ? nrm : 1e-12f;
    });
}
// sum of squares of a tensor of any dtype (the exact norm of normalize_tensor) -> partials [grid][2] (k_sum_partials' layout)
struct SumsqAnyParams { const void* x; int dtype; size_t n; int chunks; double* partials; };
template <class Ex>
SM_HD void k_sumsq_any(Ex& ex, const SumsqAnyParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState& s) {
        double a = 0.0;
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int c = 0; c < p.chunks; ++c) {
            const size_t i = start + (size_t)c * nt + tid;
            if (i >= p.n) break;
            const double v = (double)load_elem(p.x, p.dtype, i);
            a += v * v;
        }
        s.red[0] = a; s.red[1] = 0.0;
    });
    ex.template block_sum<2>(st, [&](const double* tot) { p.partials[2 * (size_t)ex.bid()] = tot[0]; p.partials[2 * (size_t)ex.bid() + 1] = tot[1]; });
}
// out = x / s in x's dtype: a 16-bit tensor divided by a Python float is an fp32 division rounded to the dtype
struct DivScalarParams { const void* x; void* out; int dtype; size_t n; float s; int chunks; };
template <class Ex>
SM_HD void k_div_scalar(Ex& ex, const DivScalarParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int c = 0; c < p.chunks; ++c) {
            const size_t i = start + (size_t)c * nt + tid;
            if (i >= p.n) break;
            const float v = load_elem(p.x, p.dtype, i) / p.s;
            if (p.dtype == DT_F32) ((float*)p.out)[i] = v;
            else if (p.dtype == DT_BF16) ((uint16_t*)p.out)[i] = f_to_bf16_any(v);
            else ((uint16_t*)p.out)[i] = f_to_f16_any(v);
        }
    });
}

constexpr int ADD_MAX_MODELS = 16;
struct AdditionParams {
    int k;
    const void* ft[ADD_MAX_MODELS];
    const void* base;
    int dtype;
    size_t n;
    int mode;                   // 0: AdditionMerge, 1: TaskAdditionMerge
    void* out;                  // dtype, [n]
    int vec8;                   // n % 8 == 0 and 16-byte aligned pointers
    int chunks;                 // octets per thread
};
template <class Ex>
SM_HD void k_addition(Ex& ex, const AdditionParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t noct = (p.n + 7) / 8;
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int q = 0; q < p.chunks; ++q) {
            const size_t oi = start + (size_t)q * nt + tid;
            if (oi >= noct) break;
            const size_t i0 = 8 * oi;
            const int cnt = p.vec8 ? 8 : (int)((p.n - i0) < 8 ? (p.n - i0) : 8);
            float b[8], acc[8], ssum[8];
            auto load8 = [&](const void* src, float* dst) {
                if (p.vec8) { load_elem8(src, p.dtype, i0, dst); return; }
                for (int e = 0; e < 8; ++e) dst[e] = e < cnt ? load_elem(src, p.dtype, i0 + e) : 0.f;
            };
            load8(p.base, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc[e] = 0.f; ssum[e] = 0.f; }
            if (p.mode == 0) {
                for (int i = 0; i < p.k; ++i) {
                    float f[8];
                    load8(p.ft[i], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = round_to_dtype(acc[e] + round_to_dtype(f[e] - b[e], p.dtype), p.dtype);
                }
            } else {
                for (int i = 0; i < p.k; ++i) {                  // pass 1: the majority sign
                    float f[8];
                    load8(p.ft[i], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ssum[e] += sign_of(round_to_dtype(f[e] - b[e], p.dtype));
                }
                for (int i = 0; i < p.k; ++i) {                  // pass 2: masked sum (the inputs come from L2 now)
                    float f[8];
                    load8(p.ft[i], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = round_to_dtype(f[e] - b[e], p.dtype);
                        const float keep = (sign_of(d) == sign_of(round_to_dtype(ssum[e], p.dtype))) ? 1.f : 0.f;
                        acc[e] += round_to_dtype(d * keep, p.dtype);   // NaN * 0 = NaN, as in the reference
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = round_to_dtype(acc[e], p.dtype);
            }
            if (p.dtype == DT_F32) {
                if (p.vec8) {
                    cf4 w0 = {acc[0], acc[1], acc[2], acc[3]}, w1 = {acc[4], acc[5], acc[6], acc[7]};
                    ((cf4*)p.out)[i0 / 4] = w0; ((cf4*)p.out)[i0 / 4 + 1] = w1;
                } else {
                    for (int e = 0; e < cnt; ++e) ((float*)p.out)[i0 + e] = acc[e];
                }
            } else {
                uint16_t h[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = p.dtype == DT_BF16 ? f_to_bf16_any(acc[e]) : f_to_f16_any(acc[e]);
                if (p.vec8) {
                    u32x4 w;
                    w.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); w.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                    w.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16); w.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
                    ((u32x4*)p.out)[oi] = w;
                } else {
                    for (int e = 0; e < cnt; ++e) ((uint16_t*)p.out)[i0 + e] = h[e];
                }
            }
        }
    });
}

// =====================================================================
// correlate_pairs (reference shard/tensor/functions.py:304-314; used by the legacy fourier.py
// operator's pairing): matrix[i][j] = mean over the trailing positions of
// cosine_similarity(t_i, t_j, dim=0).nan_to_num(0).  The tensors are viewed as [rows = shape[0]]
// x [cols = the rest]; one thread owns a column and walks a slice of the rows (coalesced across
// the lanes); partial sums of the K(K+1)/2 products go to a workspace, a second kernel finishes.
// =====================================================================
constexpr int CORR_MAX_K = 8;
constexpr int CORR_MAX_P = CORR_MAX_K * (CORR_MAX_K + 1) / 2;
struct CorrPartialParams {
    int k;
    const void* t[CORR_MAX_K];
    int dtype;
    size_t rows, cols;
    int splits;                 // row slices (grid y)
    double* part;               // [splits][P][cols]
};
SM_HD int corr_pair_index(int i, int j, int k) { return i * k - i * (i - 1) / 2 + (j - i); }     // i <= j
template <class Ex>
SM_HD void k_corr_partial(Ex& ex, const CorrPartialParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t cblocks = (p.cols + nt - 1) / nt;
    const size_t cb = (size_t)ex.bid() % cblocks, sp = (size_t)ex.bid() / cblocks;
    const int P = p.k * (p.k + 1) / 2;
    const size_t r0 = p.rows * sp / p.splits, r1 = p.rows * (sp + 1) / p.splits;
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t c = cb * nt + tid;
        if (c >= p.cols) return;
        double acc[CORR_MAX_P];
#pragma unroll
        for (int q = 0; q < CORR_MAX_P; ++q) acc[q] = 0.0;
        for (size_t r = r0; r < r1; ++r) {
            float v[CORR_MAX_K];
#pragma unroll
            for (int i = 0; i < CORR_MAX_K; ++i) v[i] = i < p.k ? load_elem(p.t[i], p.dtype, r * p.cols + c) : 0.f;
            int q = 0;
#pragma unroll
            for (int i = 0; i < CORR_MAX_K; ++i) {
#pragma unroll
                for (int j = i; j < CORR_MAX_K; ++j) {
                    if (i < p.k && j < p.k) acc[q] += (double)v[i] * (double)v[j];
                    if (j < p.k) ++q;            // q runs over the pairs of the first k tensors only
                }
            }
        }
        int q = 0;
        for (int i = 0; i < p.k; ++i)
            for (int j = i; j < p.k; ++j, ++q) p.part[((size_t)sp * P + q) * p.cols + c] = acc[q];
    });
}
struct CorrFinishParams {
    int k;
    size_t cols;
    int splits;
    const double* part;
    float eps;                  // torch.nn.functional.cosine_similarity: 1e-8
    float* out;                 // [k][k], host-mapped
};
template <class Ex>
SM_HD void k_corr_finish(Ex& ex, const CorrFinishParams& p) {
    // one work-group per pair (i < j)
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const int P = p.k * (p.k + 1) / 2;
    int pi = 0, pj = 0, cnt = 0;
    for (int i = 0; i < p.k; ++i)
        for (int j = i + 1; j < p.k; ++j, ++cnt)
            if (cnt == ex.bid()) { pi = i; pj = j; }
    const int qij = corr_pair_index(pi, pj, p.k), qii = corr_pair_index(pi, pi, p.k), qjj = corr_pair_index(pj, pj, p.k);
    ex.each(st, [&](int tid, EmptyState& s) {
        double sum = 0.0;
        for (size_t c = tid; c < p.cols; c += nt) {
            double dij = 0, dii = 0, djj = 0;
            for (int sp = 0; sp < p.splits; ++sp) {
                dij += p.part[((size_t)sp * P + qij) * p.cols + c];
                dii += p.part[((size_t)sp * P + qii) * p.cols + c];
                djj += p.part[((size_t)sp * P + qjj) * p.cols + c];
            }
            const double ni = sqrt(dii), nj = sqrt(djj);
            float cosv = (float)(dij / ((ni > p.eps ? ni : (double)p.eps) * (nj > p.eps ? nj : (double)p.eps)));
            if (is_nan(cosv)) cosv = 0.f;                                    // nan_to_num(0)
            else if (is_inf(cosv)) cosv = cosv > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
            sum += cosv;
        }
        s.red[0] = sum;
    });
    ex.template block_sum<1>(st, [&](const double* tot) {
        const float m = (float)(tot[0] / (double)p.cols);
        p.out[pi * p.k + pj] = m;
        p.out[pj * p.k + pi] = m;
    });
}

// expand half-spectrum planes to the full complex spectrum (test / API helper:
// the reference's fft_transform returns all R x C bins)
struct ExpandParams {
    const float* re; const float* im;   // planes [Cb][R]
    int R, C, Cb;
    cf2* full;                          // [R][C]
    int chunks;
};
template <class Ex>
SM_HD void k_expand(Ex& ex, const ExpandParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t total = (size_t)p.R * p.C;
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int q = 0; q < p.chunks; ++q) {
            const size_t i = start + (size_t)q * nt + tid;
            if (i >= total) break;
            const int r = (int)(i / p.C), k = (int)(i % p.C);
            cf2 v;
            if (k < p.Cb) {
                v.x = p.re[(size_t)k * p.R + r]; v.y = p.im[(size_t)k * p.R + r];
            } else {
                const int kk = p.C - k, rr = (p.R - r) % p.R;
                v.x = p.re[(size_t)kk * p.R + rr]; v.y = -p.im[(size_t)kk * p.R + rr];
            }
            p.full[i] = v;
        }
    });
}

// gather a full complex spectrum [R][C] (interleaved) into half-spectrum planes
// [Cb][R].  sym = 1 first projects onto the Hermitian part, which is what taking
// `.real` of the inverse transform does (reference functions.py:71-73).
struct PackParams {
    const cf2* full; int R, C, Cb;
    float* re; float* im;
    int sym;
    int chunks;
};
template <class Ex>
SM_HD void k_pack(Ex& ex, const PackParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const size_t total = (size_t)p.Cb * p.R;
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int q = 0; q < p.chunks; ++q) {
            const size_t i = start + (size_t)q * nt + tid;
            if (i >= total) break;
            const int k = (int)(i / p.R), r = (int)(i % p.R);
            cf2 v = p.full[(size_t)r * p.C + k];
            if (p.sym) {
                const int rr = (p.R - r) % p.R, kk = (p.C - k) % p.C;
                const cf2 m = p.full[(size_t)rr * p.C + kk];
                v.x = 0.5f * (v.x + m.x); v.y = 0.5f * (v.y - m.y);
            }
            p.re[i] = v.x; p.im[i] = v.y;
        }
    });
}

// interleaved complex <-> two planes in the same (row-major) order; cull in place
struct SplitParams { const cf2* full; float* re; float* im; size_t n; int chunks; };
template <class Ex>
SM_HD void k_split(Ex& ex, const SplitParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int q = 0; q < p.chunks; ++q) {
            const size_t i = start + (size_t)q * nt + tid;
            if (i >= p.n) break;
            const cf2 v = p.full[i];
            p.re[i] = v.x; p.im[i] = v.y;
        }
    });
}
struct JoinParams { const float* re; const float* im; cf2* full; size_t n; int chunks; };
template <class Ex>
SM_HD void k_join(Ex& ex, const JoinParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int q = 0; q < p.chunks; ++q) {
            const size_t i = start + (size_t)q * nt + tid;
            if (i >= p.n) break;
            cf2 v = {p.re[i], p.im[i]};
            p.full[i] = v;
        }
    });
}
struct CullParams { float* x; size_t n; const float* thr; int chunks; };
template <class Ex>
SM_HD void k_cull(Ex& ex, const CullParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const float thr = *p.thr;
    ex.each(st, [&](int tid, EmptyState&) {
        const size_t start = (size_t)ex.bid() * p.chunks * nt;
        for (int q = 0; q < p.chunks; ++q) {
            const size_t i = start + (size_t)q * nt + tid;
            if (i >= p.n) break;
            if (fabsf(p.x[i]) < thr) p.x[i] = 0.f;
        }
    });
}


// =====================================================================
// A whole SLERP pair merge of two 1-D tensors in ONE work-group (R = 1, C <= PAIR1D_MAX_C): row
// transform (two-for-one), Hermitian split, both order statistics (three-level radix select in
// LDS), slerp sums and constants, blend, cull, inverse transform, scale / add-back / cast.  The
// multi-kernel pipeline spends ~25 launches (0.25 ms of launch latency) on such a tensor - every
// norm weight of a model.  Same arithmetic, same exact thresholds; sums in another order.
// =====================================================================
constexpr int PAIR1D_MAX_C = 8192;
struct Pair1dParams {
    FftPlanDev plan;            // N = C
    SigDesc a, b;               // role a = the larger-norm input
    int C;
    float sa, sb;               // 1 / ||a||, 1 / ||b||
    float t, t_sum;
    unsigned long long rank_cut; int have_cut;
    unsigned long long rank_cull; int have_cull;
    I2Params fin;               // inv_n, post, ifft_policy, base, out, flags, norm_partials (plan / G unused)
    float* thr;                 // [2]: cutoff and cull thresholds (device, as the selection passes leave them)
    BlendConsts* consts;
};

// k-th smallest (0-based, with multiplicities w) of the keys a thread holds: keys[j] for j < nk, two key
// sets (x: always, y: optional).  11 + 10 + 10 bits, histograms in LDS; every thread returns the key.
template <class Ex, class StT, class KeyFn>
SM_HD uint32_t wg_select_lds(Ex& ex, StT& st, uint32_t* hist, uint32_t* part, uint32_t* ctl, unsigned long long rank,
                             int per_thread, KeyFn keys) {
    using S = typename StT::value_type;
    const int nt = ex.nthreads();
    uint32_t prefix = 0;
    int prefix_bits = 0;
    unsigned long long r = rank;
    for (int level = 0; level < 3; ++level) {
        const int bits = level == 0 ? 11 : 10;
        const int nbins = 1 << bits;
        const int shift = 31 - prefix_bits - bits;
        ex.each(st, [&](int tid, S&) { for (int b = tid; b < nbins; b += nt) hist[b] = 0; });
        ex.sync();
        ex.each(st, [&](int tid, S& s) {
            for (int j = 0; j < per_thread; ++j) {
                uint32_t key[2], w;
                const int n = keys(tid, s, j, key, w);
                for (int e = 0; e < n; ++e)
                    if (prefix_bits == 0 || (key[e] >> (31 - prefix_bits)) == prefix)
                        ex.lds_atomic_add(&hist[(key[e] >> shift) & (uint32_t)(nbins - 1)], w);
            }
        });
        ex.sync();
        const int per = (nbins + nt - 1) / nt;
        const int ngrp = (nt + 15) / 16;
        uint32_t* grp = part + nt;                     // group totals of 16 threads' partials (counts fit 32 bits)
        ex.each(st, [&](int tid, S&) {
            uint32_t sum = 0;
            for (int q = 0; q < per; ++q) { const int b = tid * per + q; if (b < nbins) sum += hist[b]; }
            part[tid] = sum;
        });
        ex.sync();
        ex.each(st, [&](int tid, S&) {
            if (tid >= ngrp) return;
            uint32_t g = 0;
            for (int q = 0; q < 16; ++q) { const int i = tid * 16 + q; if (i < nt) g += part[i]; }
            grp[tid] = g;
        });
        ex.sync();
        // every thread finds where its own bins start (a few independent LDS reads); the one whose range
        // holds the rank walks its bins
        ex.each(st, [&](int tid, S&) {
            uint32_t excl = 0, total = 0;
            const int g0 = tid / 16;
            for (int g = 0; g < ngrp; ++g) { const uint32_t v = grp[g]; total += v; if (g < g0) excl += v; }
            for (int q = g0 * 16; q < tid; ++q) excl += part[q];
            if (total == 0) { if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; } return; }
            const uint32_t rr = r >= (unsigned long long)total ? total - 1u : (uint32_t)r;
            const uint32_t mine = part[tid];
            if (rr >= excl && rr - excl < mine) {
                uint32_t cum = excl;
                int b = tid * per;
                while (b < nbins - 1 && cum + hist[b] <= rr) { cum += hist[b]; ++b; }
                ctl[0] = (uint32_t)b; ctl[1] = rr - cum; ctl[2] = 0u;
            }
        });
        ex.sync();
        prefix = (prefix << bits) | ctl[0];
        prefix_bits += bits;
        r = (unsigned long long)ctl[1] | ((unsigned long long)ctl[2] << 32);
        ex.sync();
    }
    return prefix;
}

template <class P> constexpr bool pair1d_plan() {      // 1-D tensors go through this kernel up to PAIR1D_MAX_C elements only
    if constexpr (P::is_static) return P::N <= PAIR1D_MAX_C; else return true;
}
template <class P, class Ex>
SM_HD void k_pair1d(Ex& ex, const Pair1dParams& p) {
    if constexpr (!pair1d_plan<P>()) { return; } else {      // (the 16384- / 28672-point bodies were dead code with 594 spilled registers)
    typename Ex::template State<FftState> st;
    ex.init(st);
    const FftPlanDev& pl = p.plan;
    const int T = plan_T<P>(pl), C = plan_N<P>(pl);
    const int LF = plan_lds<P>(pl);
    const int Cb = C / 2 + 1;
    float* fftbuf = ex.lds() + LDS_SCRATCH_FLOATS;
    float* zr = fftbuf + LF;
    float* zi = zr + C;
    uint32_t* hist = (uint32_t*)(zi + C);
    uint32_t* part = hist + HIST1_BINS;             // nt words + nt / 16 group totals
    uint32_t* ctl = part + 1024 + 64;               // 8 words + the constants broadcast
    float* cbuf = (float*)(ctl + 8);
    const int QB = (Cb + T - 1) / T;                // bins per thread (<= EMAX / 2 + 1)

    // ---- forward: z = a + i b, one C-point transform, split by Hermitian symmetry ----
    // (straight-line on purpose: wrapping the two transforms / the two selections into loops to share
    //  their code made the 8192-point kernel spill 378 registers and 50 % slower)
    ex.each(st, [&](int tid, FftState& s) {
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int n = tid + q * T;
            s.xr[q] = n < C ? load_sig1(p.a, (size_t)n) : 0.f;
            s.xi[q] = n < C ? load_sig1(p.b, (size_t)n) : 0.f;
        }
    });
    auto nat_scatter = [&](int tid, FftState& s, auto comp_c) {
        constexpr int comp = decltype(comp_c)::value;
        const float* x = comp_of<comp>(s);
#pragma unroll
        for (int q = 0; q < EMAX; ++q) { const int n = tid + q * T; if (n < C) fftbuf[lpad(n)] = x[q]; }
    };
    auto fin_gather = [&](int tid, FftState& s, auto comp_c) {
        constexpr int comp = decltype(comp_c)::value;
        float* o = comp_of<comp>(s);
#pragma unroll
        for (int q = 0; q < EMAX; ++q) { const int k = tid + q * T; if (k < C) o[q] = fftbuf[lpad(k)]; }
    };
    wg_fft<P>(ex, st, pl, fftbuf, nat_scatter, fin_gather);
    ex.each(st, [&](int tid, FftState& s) {
#pragma unroll
        for (int q = 0; q < EMAX; ++q) { const int k = tid + q * T; if (k < C) { zr[k] = s.xr[q]; zi[k] = s.xi[q]; } }
    });
    ex.sync();
    // bins k = tid + q T (q < QB): Re a -> xr[q], Im a -> xi[q], Re b -> xr[EMAX/2 + 1 + q]   (QB <= EMAX/2 + 1)
    constexpr int RB0 = EMAX / 2 + 1;
    ex.each(st, [&](int tid, FftState& s) {
        for (int q = 0; q < QB; ++q) {
            const int k = tid + q * T;
            float ra = 0.f, ia = 0.f, rb = 0.f;
            if (k < Cb) {
                const int m = (C - k) % C;
                const float r1 = zr[k], i1 = zi[k], r2 = zr[m], i2 = zi[m];
                ra = 0.5f * (r1 + r2) * p.sa; ia = 0.5f * (i1 - i2) * p.sa;
                rb = 0.5f * (i1 + i2) * p.sb;
            }
            s.xr[q] = ra; s.xi[q] = ia; s.xr[RB0 + q] = rb;
        }
    });
    ex.sync();

    // ---- cutoff threshold over |Re a| and |Re b| (functions.py:115-119) ----
    float thr0 = 0.f;
    if (p.have_cut) {
        const uint32_t key = wg_select_lds(ex, st, hist, part, ctl, p.rank_cut, QB,
            [&](int tid, FftState& s, int q, uint32_t* key2, uint32_t& w) {
                const int k = tid + q * T;
                if (k >= Cb) return 0;
                w = (uint32_t)bin_weight(k, C);
                key2[0] = f2u(s.xr[q]) & 0x7fffffffu; key2[1] = f2u(s.xr[RB0 + q]) & 0x7fffffffu;
                return 2;
            });
        thr0 = u2f(key);
    }
    // ---- slerp-class sums and constants (functions.py:36-43) ----
    ex.each(st, [&](int tid, FftState& s) {
        double s00 = 0, s01 = 0, s11 = 0, cnt = 0;
        for (int q = 0; q < QB; ++q) {
            const int k = tid + q * T;
            if (k >= Cb) continue;
            const float a = s.xr[q], b = s.xr[RB0 + q];
            if (same_sign(a, b) && !(fabsf(b) < thr0)) {
                const double w = (double)bin_weight(k, C);
                s00 += w * (double)(a * a); s01 += w * (double)(a * b); s11 += w * (double)(b * b); cnt += w;
            }
        }
        s.red[0] = s00; s.red[1] = s01; s.red[2] = s11; s.red[3] = cnt;
    });
    ex.template block_sum<4>(st, [&](const double* tot) {
        const double s00 = tot[0], s01 = tot[1], s11 = tot[2], cnt = tot[3];
        BlendConsts c;
        c.thr = thr0;
        c.s00 = s00; c.s01 = s01; c.s11 = s11; c.n_slerp = (unsigned long long)cnt;
        double dot = s01 / (sqrt(s00) * sqrt(s11));
        if (dot > 1.0) dot = 1.0;
        if (dot < -1.0) dot = -1.0;
        const float dotf = (float)dot;
        const float theta = acosf(dotf) * p.t;
        double rel2 = s11 - 2.0 * (double)dotf * s01 + (double)dotf * (double)dotf * s00;
        if (rel2 < 0) rel2 = 0;
        double reln = sqrt(rel2);
        if (reln < 1e-12) reln = 1e-12;
        c.dot = dotf; c.cos_t = cosf(theta); c.sin_t = sinf(theta); c.inv_rel = (float)(1.0 / reln);
        c.pad[0] = c.pad[1] = c.pad[2] = 0.f;
        *p.consts = c;
        p.thr[0] = thr0;
        cbuf[0] = c.dot; cbuf[1] = c.cos_t; cbuf[2] = c.sin_t; cbuf[3] = c.inv_rel;
    });
    BlendConsts bc;
    memset(&bc, 0, sizeof bc);
    bc.thr = thr0; bc.dot = cbuf[0]; bc.cos_t = cbuf[1]; bc.sin_t = cbuf[2]; bc.inv_rel = cbuf[3];
    // ---- blend (functions.py:121-145), cull threshold over |Re R| (:146-147) ----
    ex.each(st, [&](int tid, FftState& s) {
        for (int q = 0; q < QB; ++q) {
            const int k = tid + q * T;
            if (k < Cb) s.xr[q] = blend_slerp_one(bc, p.t_sum, s.xr[q], s.xr[RB0 + q]);
        }
    });
    ex.sync();
    float thr1 = 0.f;
    if (p.have_cull) {
        const uint32_t key = wg_select_lds(ex, st, hist, part, ctl, p.rank_cull, QB,
            [&](int tid, FftState& s, int q, uint32_t* key2, uint32_t& w) {
                const int k = tid + q * T;
                if (k >= Cb) return 0;
                w = (uint32_t)bin_weight(k, C);
                key2[0] = f2u(s.xr[q]) & 0x7fffffffu;
                return 1;
            });
        thr1 = u2f(key);
    }
    // ---- inverse: Hermitian extension of (Re R culled, Im a), swap trick, one C-point transform ----
    ex.each(st, [&](int tid, FftState& s) {
        if (tid == 0) p.thr[1] = thr1;
        for (int q = 0; q < QB; ++q) {
            const int k = tid + q * T;
            if (k < Cb) { float r = s.xr[q]; if (fabsf(r) < thr1) r = 0.f; zr[k] = r; zi[k] = s.xi[q]; }
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, FftState& s) {
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int n = tid + q * T;
            float re = 0.f, im = 0.f;
            if (n < C) {
                if (n < Cb) { re = zr[n]; im = zi[n]; }
                else { re = zr[C - n]; im = -zi[C - n]; }
            }
            s.xr[q] = im; s.xi[q] = re;             // ifft(x) = swap(fft(swap(x)))
        }
    });
    ex.sync();
    wg_fft<P>(ex, st, pl, fftbuf, nat_scatter, fin_gather);
    ex.each(st, [&](int tid, FftState& s) {
        uint32_t nan1 = 0, inf1 = 0, nan2 = 0, inf2 = 0;
        double ss = 0.0;
#pragma unroll
        for (int q = 0; q < EMAX; ++q) {
            const int n = tid + q * T;
            if (n < C) {
                float v;
                i2_finish(p.fin, s.xi[q], (size_t)n, nan1, inf1, nan2, inf2, v);
                ss += (double)v * v;
                if (p.fin.out_mode == OUT_BF16) ((uint16_t*)p.fin.out)[n] = f_to_bf16(v);
                else ((float*)p.fin.out)[n] = v;
            }
        }
        if (nan1) ex.global_atomic_add_u32(&p.fin.flags[0], nan1);
        if (inf1) ex.global_atomic_or_u32(&p.fin.flags[1], 1u);
        if (nan2) ex.global_atomic_add_u32(&p.fin.flags[2], nan2);
        if (inf2) ex.global_atomic_or_u32(&p.fin.flags[3], 1u);
        s.red[0] = ss; s.red[1] = 0.0;
    });
    if (p.fin.norm_partials) {
        ex.sync();
        ex.template block_sum<2>(st, [&](const double* tot) { p.fin.norm_partials[0] = tot[0]; p.fin.norm_partials[1] = 0.0; });
    }
    }
}

// =====================================================================
// Long / rough column lengths: R = p * M with M a length the work-group engine plans
// (even) and p <= DFTP_MAX_P anything (43 for the 11008 of Llama-2-7B, 37 for the 18944 of
// Qwen2-7B, 2 for 65536 ...).  Decimation in frequency with n = n1 + M n2, k = p k1 + k2:
//
//     X[p k1 + k2] = sum_n1 W_M^(n1 k1) * { W_R^(n1 k2) * sum_n2 W_p^(n2 k2) x[n1 + M n2] }
//
// so the p row BLOCKS [n2 M, (n2+1) M) are first combined by a p-point DFT (this kernel, in
// place on the row pass's output T1, with the W_R twiddle), and block k2 then goes through
// the ordinary M-point column pass - the pipeline treats the tensor as p slices of M rows
// (the `batch` machinery) whose planes hold the bins k = p k1 + k2 of slice k2.  The inverse
// runs the other way round: M-point inverse column passes per slice, then the conjugate
// twiddle and the inverse p-point DFT across the slices of G, then the inverse row pass.
// The direct p-point sums cost O(p) per element: HBM-bound up to p ~ 16, ALU-bound beyond.
// =====================================================================
constexpr int DFTP_MAX_P = 256;
constexpr int DFTP_COLS = 16;          // bin columns per work-group: 16 x 16 B = 256 contiguous bytes
constexpr int DFTP_KB = 4;             // outputs a thread accumulates together
struct DftpParams {
    cf4* buf;              // T1 (forward) or G (inverse): p slices
    const cf2* tw;         // tw[j] = exp(-2 pi i j / (p M)), j < p M
    int p, M;
    int units;             // row units per slice: M (a cf4 = signals a, b of one row) or M / 2 (rowpair)
    int rowpair;           // 1: a cf4 = rows 2u, 2u + 1 of ONE signal (their twiddles differ)
    int ilv;               // rows interleaved in groups of ilv (T1 of the two-signal row pass)
    int pitch;             // cf4 per row unit
    int ncols;             // valid bin columns
    size_t slice_stride;   // cf4 between slices
    int inverse;
    int cols;              // k_dftp_pairs: bin columns per work-group (16 ... 128; the fewer outputs per column - small p -
                           // the more columns share a work-group)
};
SM_HD cf2 cmul(cf2 a, cf2 b) { cf2 r = {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; return r; }
template <class Ex>
SM_HD void k_dftp(Ex& ex, const DftpParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    cf4* tile = (cf4*)(ex.lds() + LDS_SCRATCH_FLOATS);             // [p][DFTP_COLS]
    cf2* wp = (cf2*)(tile + (size_t)p.p * DFTP_COLS);               // W_p^j (conjugated for the inverse)
    const int ntiles = (p.ncols + DFTP_COLS - 1) / DFTP_COLS;
    const int u = ex.bid() / ntiles, c0 = (ex.bid() % ntiles) * DFTP_COLS;
    if (u >= p.units) return;
    const int n1a = p.rowpair ? 2 * u : u, n1b = p.rowpair ? 2 * u + 1 : u;
    const size_t row_off = ((size_t)(u / p.ilv) * p.pitch) * p.ilv + u % p.ilv;
    const float sgn_im = p.inverse ? -1.f : 1.f;
    ex.each(st, [&](int tid, EmptyState&) {
        for (int j = tid; j < p.p; j += nt) { cf2 w = p.tw[(size_t)j * p.M]; w.y *= sgn_im; wp[j] = w; }
        for (int idx = tid; idx < p.p * DFTP_COLS; idx += nt) {
            const int s = idx / DFTP_COLS, c = c0 + idx % DFTP_COLS;
            cf4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < p.ncols) {
                v = p.buf[(size_t)s * p.slice_stride + row_off + (size_t)c * p.ilv];
                if (p.inverse) {                   // conjugate twiddle first: slice index s is k2
                    cf2 wa = p.tw[(size_t)n1a * s], wb = p.tw[(size_t)n1b * s];
                    wa.y = -wa.y; wb.y = -wb.y;
                    const cf2 a = cmul({v.x, v.y}, wa), b = cmul({v.z, v.w}, wb);
                    v = {a.x, a.y, b.x, b.y};
                }
            }
            tile[idx] = v;
        }
    });
    ex.sync();
    // a thread owns one column and the outputs k = kl + j * (nt / 16): DFTP_KB of them share every tile
    // value it reads, and the two complex halves of a cf4 ride the packed-f32 ALU ops side by side
    ex.each(st, [&](int tid, EmptyState&) {
        const int cl = tid % DFTP_COLS, kl = tid / DFTP_COLS, kstep = nt / DFTP_COLS;
        const int c = c0 + cl;
        if (c >= p.ncols) return;
        for (int kb0 = kl; kb0 < p.p; kb0 += kstep * DFTP_KB) {
            vf2 accr[DFTP_KB], acci[DFTP_KB];
            int kk[DFTP_KB], wi[DFTP_KB];
#pragma unroll
            for (int j = 0; j < DFTP_KB; ++j) {
                const int k = kb0 + j * kstep;
                kk[j] = k < p.p ? k : 0;                 // (a lane past the end computes output 0 again and drops it)
                wi[j] = 0; accr[j] = mk2(0.f, 0.f); acci[j] = mk2(0.f, 0.f);
            }
            for (int sidx = 0; sidx < p.p; ++sidx) {
                const cf4 v = tile[sidx * DFTP_COLS + cl];
                const vf2 vr = mk2(v.x, v.z), vi = mk2(v.y, v.w);
#pragma unroll
                for (int j = 0; j < DFTP_KB; ++j) {
                    const cf2 w = wp[wi[j]];
                    accr[j] += vr * w.x - vi * w.y;
                    acci[j] += vr * w.y + vi * w.x;
                    wi[j] += kk[j]; if (wi[j] >= p.p) wi[j] -= p.p;
                }
            }
#pragma unroll
            for (int j = 0; j < DFTP_KB; ++j) {
                const int k = kb0 + j * kstep;
                if (k >= p.p) continue;
                cf2 a = {accr[j].x, acci[j].x}, b = {accr[j].y, acci[j].y};
                if (!p.inverse) {                  // slice index k is k2
                    a = cmul(a, p.tw[(size_t)n1a * k]); b = cmul(b, p.tw[(size_t)n1b * k]);
                }
                cf4 o = {a.x, a.y, b.x, b.y};
                p.buf[(size_t)k * p.slice_stride + row_off + (size_t)c * p.ilv] = o;
            }
        }
    });
}

// The same transform for p <= DFTP_PAIR_MAX_P with half the multiplies and no index arithmetic in the inner
// loop (the generic kernel spends more instructions on (s k) mod p than on the sums).  Outputs k and
// p - k use conjugate twiddles (c, -+s): with A = sum vr c, B = sum vi s, C = sum vr s, D = sum vi c
//     X[k] = (A - B, C + D),   X[p - k] = (A + B, D - C)
// so a "pair unit" h = 0 .. p/2 costs four (packed) FMAs per input for two outputs.  The matrix
// Wh[s][h] = W_p^(s h) sits in LDS, filled once per work-group, which then walks DFTP_UNITS row units.
constexpr int DFTP_PAIR_MAX_P = 126;
constexpr int DFTP_UNITS = 4;
SM_HD int dftp_pairs_cols(int p) {                 // columns per 256-thread work-group
    const int H = p / 2 + 1;
    if (H > 8) return DFTP_COLS;
    int h2 = 1;
    while (h2 < H) h2 *= 2;
    return 256 / h2 > 128 ? 128 : 256 / h2;
}
SM_HD size_t dftp_pairs_lds_bytes(int p) {
    const int H = p / 2 + 1;
    return (size_t)p * dftp_pairs_cols(p) * sizeof(cf4) + (size_t)p * H * sizeof(cf2);
}
template <int NP, class Ex, class St>
SM_HD void dftp_pairs_compute(Ex& ex, St& st, const DftpParams& p, const cf4* tile, const cf2* wh, int H, int c0,
                              size_t row_off, int n1a, int n1b) {
    const int nt = ex.nthreads();
    ex.each(st, [&](int tid, EmptyState&) {
        const int cl = tid % p.cols, hl = tid / p.cols, hstep = nt / p.cols;
        const int c = c0 + cl;
        if (c >= p.ncols) return;
        vf2 A[NP], B[NP], C[NP], D[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) { A[j] = mk2(0.f, 0.f); B[j] = A[j]; C[j] = A[j]; D[j] = A[j]; }
        const cf4* tp = tile + cl;
        const cf2* wq = wh + hl;                   // unit h = hl + j * hstep (reads past H land in the padding row)
        for (int sidx = 0; sidx < p.p; ++sidx) {
            const cf4 v = *tp;
            const vf2 vr = mk2(v.x, v.z), vi = mk2(v.y, v.w);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const cf2 w = wq[j * hstep];
                A[j] += vr * w.x; B[j] += vi * w.y; C[j] += vr * w.y; D[j] += vi * w.x;
            }
            tp += p.cols; wq += H;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int h = hl + j * hstep;
            if (h >= H) continue;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int k = side == 0 ? h : p.p - h;
                if (side == 1 && (h == 0 || 2 * h == p.p)) continue;          // its own partner
                const vf2 re = side == 0 ? A[j] - B[j] : A[j] + B[j];
                const vf2 im = side == 0 ? C[j] + D[j] : D[j] - C[j];
                cf2 a = {re.x, im.x}, b = {re.y, im.y};
                if (!p.inverse) { a = cmul(a, p.tw[(size_t)n1a * k]); b = cmul(b, p.tw[(size_t)n1b * k]); }
                cf4 o = {a.x, a.y, b.x, b.y};
                p.buf[(size_t)k * p.slice_stride + row_off + (size_t)c * p.ilv] = o;
            }
        }
    });
}
template <class Ex>
SM_HD void k_dftp_pairs(Ex& ex, const DftpParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    const int H = p.p / 2 + 1;
    const int cols = p.cols;
    cf4* tile = (cf4*)(ex.lds() + LDS_SCRATCH_FLOATS);             // [p][cols]
    cf2* wh = (cf2*)(tile + (size_t)p.p * cols);                    // [p][H] (+ slack: the launch adds a row)
    const int ntiles = (p.ncols + cols - 1) / cols;
    const int ug = ex.bid() / ntiles, c0 = (ex.bid() % ntiles) * cols;
    const float sgn_im = p.inverse ? -1.f : 1.f;
    ex.each(st, [&](int tid, EmptyState&) {
        for (int idx = tid; idx < p.p * H; idx += nt) {
            const int sidx = idx / H, h = idx % H;
            cf2 w = p.tw[(size_t)((sidx * h) % p.p) * p.M];
            w.y *= sgn_im;
            wh[idx] = w;
        }
        for (int idx = p.p * H + tid; idx < (p.p + 1) * H + 64; idx += nt) { cf2 z = {0.f, 0.f}; wh[idx] = z; }
    });
    for (int uu = 0; uu < DFTP_UNITS; ++uu) {
        const int u = ug * DFTP_UNITS + uu;
        if (u >= p.units) break;
        const int n1a = p.rowpair ? 2 * u : u, n1b = p.rowpair ? 2 * u + 1 : u;
        const size_t row_off = ((size_t)(u / p.ilv) * p.pitch) * p.ilv + u % p.ilv;
        ex.sync();                                   // the previous unit's tile has been consumed (and wh is filled)
        ex.each(st, [&](int tid, EmptyState&) {
            for (int idx = tid; idx < p.p * cols; idx += nt) {
                const int sidx = idx / cols, c = c0 + idx % cols;
                cf4 v = {0.f, 0.f, 0.f, 0.f};
                if (c < p.ncols) {
                    v = p.buf[(size_t)sidx * p.slice_stride + row_off + (size_t)c * p.ilv];
                    if (p.inverse) {
                        cf2 wa = p.tw[(size_t)n1a * sidx], wb = p.tw[(size_t)n1b * sidx];
                        wa.y = -wa.y; wb.y = -wb.y;
                        const cf2 a = cmul({v.x, v.y}, wa), b = cmul({v.z, v.w}, wb);
                        v = {a.x, a.y, b.x, b.y};
                    }
                }
                tile[idx] = v;
            }
        });
        ex.sync();
        const int np = (H + nt / cols - 1) / (nt / cols);
        if (np <= 1) dftp_pairs_compute<1>(ex, st, p, tile, wh, H, c0, row_off, n1a, n1b);
        else if (np == 2) dftp_pairs_compute<2>(ex, st, p, tile, wh, H, c0, row_off, n1a, n1b);
        else if (np == 3) dftp_pairs_compute<3>(ex, st, p, tile, wh, H, c0, row_off, n1a, n1b);
        else dftp_pairs_compute<4>(ex, st, p, tile, wh, H, c0, row_off, n1a, n1b);
    }
}

// [R][C] -> [C][R], elements of 2 or 4 bytes moved as bits (64 x 64 tiles through LDS): a tensor
// whose ROW length is the rough one is merged transposed (fft2 commutes with the transpose and
// every statistic of the merge is a sum or an order statistic over all bins)
constexpr int TR_TILE = 64;         // 64 two-byte elements = one 128-byte line per tile row, both ways
struct TransposeParams { const void* src; void* dst; int R, C; int esize; };
template <class Ex>
SM_HD void k_transpose(Ex& ex, const TransposeParams& p) {
    typename Ex::template State<EmptyState> st;
    ex.init(st);
    const int nt = ex.nthreads();
    uint32_t* tile = (uint32_t*)(ex.lds() + LDS_SCRATCH_FLOATS);   // [TR_TILE][TR_TILE + 1]
    const int tc = (p.C + TR_TILE - 1) / TR_TILE;
    const int r0 = (ex.bid() / tc) * TR_TILE, c0 = (ex.bid() % tc) * TR_TILE;
    ex.each(st, [&](int tid, EmptyState&) {
        for (int idx = tid; idx < TR_TILE * TR_TILE; idx += nt) {
            const int r = r0 + idx / TR_TILE, c = c0 + idx % TR_TILE;
            uint32_t v = 0;
            if (r < p.R && c < p.C) {
                const size_t i = (size_t)r * p.C + c;
                v = p.esize == 4 ? ((const uint32_t*)p.src)[i] : (uint32_t)((const uint16_t*)p.src)[i];
            }
            tile[(idx / TR_TILE) * (TR_TILE + 1) + idx % TR_TILE] = v;
        }
    });
    ex.sync();
    ex.each(st, [&](int tid, EmptyState&) {
        for (int idx = tid; idx < TR_TILE * TR_TILE; idx += nt) {
            const int c = c0 + idx / TR_TILE, r = r0 + idx % TR_TILE;
            if (r < p.R && c < p.C) {
                const uint32_t v = tile[(idx % TR_TILE) * (TR_TILE + 1) + idx / TR_TILE];
                const size_t o = (size_t)c * p.R + r;
                if (p.esize == 4) ((uint32_t*)p.dst)[o] = v; else ((uint16_t*)p.dst)[o] = (uint16_t)v;
            }
        }
    });
}

}  // namespace smhip
