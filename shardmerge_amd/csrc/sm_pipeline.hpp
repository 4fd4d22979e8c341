// sm_pipeline.hpp - host-side orchestration of the spectral-merge pipeline,
// templated on a Backend (HipBackend in smhip_hip.hip; HostBackend in the CPU
// emulator under tests/emul).  Holds the FFT planner, the workspace, the pair
// merge sequence and the per-layer tournament of FourierMerge._merge_layer
// (reference shard/merge/fast_fourier.py:132-276).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/shardmerge_hip.h"
#include "sm_kernels.hpp"
#include "sm_aten_norm.hpp"
#include "sm_bluestein.hpp"

namespace smhip {

// ---- kernel tags -------------------------------------------------------------
#define SM_KERNEL_TAG(Tag, ParamsT, NAME, CALL)                                      \
    struct Tag {                                                                     \
        using Params = ParamsT;                                                      \
        static constexpr int waves = 4;                                              \
        static constexpr int max_threads = 1024;                                     \
        static const char* name() { return NAME; }                                   \
        template <class Ex> static SM_HD void run(Ex& ex, const Params& p) { CALL; } \
    };
// the same with explicit launch bounds (work-group size limit, waves per SIMD the registers are budgeted for)
#define SM_KERNEL_TAG_LB(Tag, ParamsT, NAME, CALL, MAXT, WAVES)                      \
    struct Tag {                                                                     \
        using Params = ParamsT;                                                      \
        static constexpr int waves = WAVES;                                          \
        static constexpr int max_threads = MAXT;                                     \
        static const char* name() { return NAME; }                                   \
        template <class Ex> static SM_HD void run(Ex& ex, const Params& p) { CALL; } \
    };
// the four transform kernels exist once per static plan plus once for DynPlan
template <class P, int GROUPS> constexpr int fft_max_threads() {
    if constexpr (P::is_static) return (GROUPS * P::T < 256) ? 256 : (GROUPS * P::T > 1024 ? 1024 : GROUPS * P::T);
    else return 1024;
}
#ifndef SM_W3_MAX_THREADS
#define SM_W3_MAX_THREADS 256
#endif
#define SM_FFT_KERNEL_TAG(Tag, ParamsT, NAME, CALL, GROUPS, WAVES)                   \
    template <class P> struct Tag {                                                  \
        using Params = ParamsT;                                                      \
        static constexpr int waves = (WAVES < 4 && P::is_static && fft_max_threads<P, GROUPS>() <= SM_W3_MAX_THREADS) ? WAVES : 4; \
        static constexpr int max_threads = waves < 4 ? fft_max_threads<P, GROUPS>() : 1024; \
        static const char* name() { return NAME; }                                   \
        template <class Ex> static SM_HD void run(Ex& ex, const Params& p) { CALL; } \
    };
SM_FFT_KERNEL_TAG(KF1, F1Params, "f1_rows_fwd", k_f1<P>(ex, p), 1, 3)
SM_FFT_KERNEL_TAG(KF2, F2Params, "f2_cols_fwd", (k_f2<P, f2_bins<P>()>(ex, p)), 2 * f2_bins<P>(), 4)
SM_FFT_KERNEL_TAG(KI1x1, I1Params, "i1_cols_inv", (k_i1<P, 1>(ex, p)), 1, 4)
SM_FFT_KERNEL_TAG(KI1x2, I1Params, "i1_cols_inv", (k_i1<P, i1_bins<P>()>(ex, p)), i1_bins<P>(), 4)
SM_FFT_KERNEL_TAG(KI2, I2Params, "i2_rows_inv", k_i2<P>(ex, p), 1, 4)
SM_FFT_KERNEL_TAG(KF2S, F2SParams, "f2s_cols_fwd1", (k_f2s<P, f2s_groups<P>()>(ex, p)), f2s_groups<P>(), 4)
// radix-4 column step folded into the row pass (k_f1q) and its column-side companions
SM_FFT_KERNEL_TAG(KF1Q, F1Params, "f1_rows_fwd", k_f1q<P>(ex, p), 4, 4)
SM_FFT_KERNEL_TAG(KF2Q, F2Params, "f2_cols_fwd", (k_f2<P, f2_bins<P>(), true>(ex, p)), 2 * f2_bins<P>(), 4)
SM_FFT_KERNEL_TAG(KF2SQ, F2SParams, "f2s_cols_fwd1", (k_f2s<P, f2s_groups<P>(), true>(ex, p)), f2s_groups<P>(), 4)
SM_FFT_KERNEL_TAG(KI1x1Q, I1Params, "i1_cols_inv", (k_i1<P, 1, true>(ex, p)), 1, 4)
SM_FFT_KERNEL_TAG(KI1x2Q, I1Params, "i1_cols_inv", (k_i1<P, i1_bins<P>(), true>(ex, p)), i1_bins<P>(), 4)
// row passes of a row length without a plan (sm_bluestein.hpp); P is the plan of the convolution length
// (one transform per work-group and two of them back to back: compiled for two waves per SIMD - 256 VGPRs - where
//  the work-group allows it; at 128 the 16384-point convolution kept 700 bytes per lane in scratch)
#define SM_BLUE_KERNEL_TAG(Tag, ParamsT, NAME, CALL)                                 \
    template <class P> struct Tag {                                                  \
        using Params = ParamsT;                                                      \
        static constexpr int max_threads = fft_max_threads<P, 1>();                  \
        static constexpr int waves = max_threads > 512 ? 4 : 2;                      \
        static const char* name() { return NAME; }                                   \
        template <class Ex> static SM_HD void run(Ex& ex, const Params& p) { CALL; } \
    };
SM_BLUE_KERNEL_TAG(KF1B, F1BParams, "f1_rows_fwd", k_f1b<P>(ex, p))
SM_BLUE_KERNEL_TAG(KI2B, I2BParams, "i2_rows_inv", k_i2b<P>(ex, p))
SM_FFT_KERNEL_TAG(KPair1d, Pair1dParams, "pair_1d", k_pair1d<P>(ex, p), 1, 1)      // one work-group per launch: all the registers it wants

// lengths that get straight-line kernels (powers of two, the 7 * 2^k of Llama-3 / Mixtral MLPs,
// the 3/5/7 * 2^k hidden and MLP sizes of other common models, and 256 / 512: the row blocks of the
// split column lengths 11008 = 43 * 256 (Llama-2-7B), 18944 = 37 * 512 (Qwen2-7B), 4544 = 71 * 64 and 4672 = 73 * 64
// (Falcon-7B: as run-time plans its 64-point blocks took 9 + 12 ms per 4544^2 pair merge): a run-time planned length
// runs the same code with its register arrays in scratch memory, 5120^2: 5.4 ms against 0.7);
// must agree with plan_shape() below
// (checked at dispatch: a mismatch silently falls back to the DynPlan kernel)
// largest work-group the column passes may use (A and B / two bins together)
#ifndef SM_COLS_MAX_THREADS
#define SM_COLS_MAX_THREADS 1024
#endif
#ifndef SM_F2_MAX_THREADS
#define SM_F2_MAX_THREADS 512
#endif
// experiment knobs for the 8192-point plan (override with -D on the hipcc line)
#ifndef SM_R14336
#define SM_R14336 16, 16, 8, 7
#endif
#ifndef SM_R7168
#define SM_R7168 32, 32, 7          // measured on MI355X as the folded column plan: 16,16,4,7 is 8 % slower
#endif
#ifndef SM_T8192
#define SM_T8192 256
#define SM_W8192 4
#define SM_R8192 32, 16, 16
#endif
// third parameter: complex (float2) LDS exchanges; plan_uses_cx() must agree
#define SM_STATIC_PLANS(X)                 \
    X(SPlan<1024, 64, false, 4, 32, 32>)       \
    X(SPlan<2048, 64, false, 4, 16, 16, 8>)    \
    X(SPlan<4096, 128, false, 4, 16, 16, 16>)  \
    X(SPlan<8192, SM_T8192, false, SM_W8192, SM_R8192>)  \
    X(SPlan<16384, 512, false, 4, 32, 32, 16>)\
    X(SPlan<14336, 512, false, 4, SM_R14336>) \
    X(SPlan<28672, 1024, false, 4, 16, 16, 16, 7>) \
    X(SPlan<3072, 128, false, 4, 16, 16, 4, 3>)    \
    X(SPlan<3584, 128, false, 4, 16, 32, 7>)       \
    X(SPlan<5120, 256, false, 4, 16, 16, 4, 5>)    \
    X(SPlan<6144, 256, false, 4, 16, 16, 8, 3>)    \
    X(SPlan<7168, 256, false, 4, SM_R7168>)    \
    X(SPlan<12288, 512, false, 4, 16, 16, 16, 3>)  \
    X(SPlan<13824, 512, false, 4, 32, 16, 3, 3, 3>) \
    X(SPlan<27648, 1024, false, 4, 32, 32, 3, 3, 3>) \
    X(SPlan<2304, 128, false, 4, 16, 16, 3, 3>) \
    X(SPlan<256, 64, false, 4, 8, 8, 4>)            \
    X(SPlan<512, 64, false, 4, 32, 16>)             \
    X(SPlan<64, 64, false, 4, 8, 8>)                \
    X(SPlan<128, 64, false, 4, 16, 8>)

// measured on MI355X (8192^2): the complex exchange halves occupancy and brings spills back -
// 1.6x slower than split exchanges, so no plan uses it for now
inline bool plan_uses_cx(int N) { (void)N; return false; }

template <class PL> inline void plan_from_static(int N, int& T, std::vector<int>& radices, bool& found) {
    if (!found && PL::N == N) {
        T = PL::T; radices.clear();
        for (int i = 0; i < PL::npass; ++i) radices.push_back(PL::radix(i));
        found = true;
    }
}
inline bool plan_shape_static(int N, int& T, std::vector<int>& radices) {
    bool found = false;
#define SM_PLAN_FROM(...) plan_from_static<__VA_ARGS__>(N, T, radices, found);
    SM_STATIC_PLANS(SM_PLAN_FROM)
#undef SM_PLAN_FROM
    return found;
}

template <class PL>
inline bool plan_matches(const FftPlanDev& pl) {
    if (pl.N != PL::N || pl.T != PL::T || pl.npass != PL::npass || pl.lds_floats != PL::lds_floats || (pl.cx != 0) != PL::cx) return false;
    for (int i = 0; i < PL::npass; ++i)
        if (pl.radix[i] != PL::radix(i)) return false;
    return true;
}
SM_KERNEL_TAG(KF2R1, F2Params, "f2_cols_fwd", k_f2_r1(ex, p))
SM_KERNEL_TAG(KI1R1, I1Params, "i1_cols_inv", k_i1_r1(ex, p))
SM_KERNEL_TAG(KPublish, PublishParams, "publish", k_publish(ex, p))
SM_KERNEL_TAG(KHist, HistParams, "select_hist", k_hist(ex, p))
SM_KERNEL_TAG(KScan, ScanParams, "select_scan", k_scan(ex, p))
SM_KERNEL_TAG(KSelect2, Select2Params, "select_lvl2", k_select2<false>(ex, p))
SM_KERNEL_TAG(KSelect2Cull, Select2Params, "select_lvl2_cull", k_select2<false>(ex, p))     // the cull's pass: returns at once after a confirmed speculation
SM_KERNEL_TAG(KBlendSel, Select2Params, "blend", k_select2<true>(ex, p))
SM_KERNEL_TAG(KSpecCheck, SpecCheckParams, "select_spec_check", k_spec_check(ex, p))
SM_KERNEL_TAG(KSelect3, Select3Params, "select_lvl3_cand", k_select3(ex, p))
SM_KERNEL_TAG(KReduceCand, ReduceCandParams, "slerp_reduce_cand", k_reduce_cand(ex, p))
SM_KERNEL_TAG(KReduce, ReduceParams, "slerp_reduce", k_reduce(ex, p))
SM_KERNEL_TAG(KSlerpConsts, SlerpConstParams, "slerp_consts", k_slerp_consts(ex, p))
SM_KERNEL_TAG_LB(KClassEmf, ClassEmfParams, "class_norm_stats", k_class_emf(ex, p), 256, 2)
SM_KERNEL_TAG(KSumPartials, SumPartialsParams, "sum_partials", k_sum_partials(ex, p))
SM_KERNEL_TAG(KDeltaNorms, DeltaNormsParams, "delta_norms", k_delta_norms(ex, p))
SM_KERNEL_TAG(KSumPartialsN, SumNParams, "sum_partials", k_sum_partials_n(ex, p))
SM_KERNEL_TAG(KBlend, BlendParams, "blend", k_blend(ex, p))
SM_KERNEL_TAG(KCombine, CombineParams, "combine", k_combine(ex, p))
SM_KERNEL_TAG(KExpand, ExpandParams, "expand_full", k_expand(ex, p))
SM_KERNEL_TAG(KPack, PackParams, "pack_planes", k_pack(ex, p))
SM_KERNEL_TAG(KSplit, SplitParams, "split_complex", k_split(ex, p))
SM_KERNEL_TAG(KJoin, JoinParams, "join_complex", k_join(ex, p))
SM_KERNEL_TAG(KCull, CullParams, "cull_inplace", k_cull(ex, p))
SM_KERNEL_TAG(KAddition, AdditionParams, "addition_merge", k_addition(ex, p))
SM_KERNEL_TAG(KFnSums, FnSumsParams, "fn_slerp_sums", k_fn_sums(ex, p))
SM_KERNEL_TAG(KFnSlerpFin, FnSlerpFinParams, "fn_slerp_consts", k_fn_slerp_fin(ex, p))
SM_KERNEL_TAG(KFnSlerpRows0, FnSlerpRowsParams, "fn_slerp_relnorm", k_fn_slerp_rows<0>(ex, p))
SM_KERNEL_TAG(KFnSlerpRows1, FnSlerpRowsParams, "fn_slerp_out", k_fn_slerp_rows<1>(ex, p))
SM_KERNEL_TAG(KFnSlerpDen, FnSlerpDenParams, "fn_slerp_den", k_fn_slerp_den(ex, p))
SM_KERNEL_TAG(KSumsqAny, SumsqAnyParams, "fn_sumsq", k_sumsq_any(ex, p))
SM_KERNEL_TAG(KDivScalar, DivScalarParams, "fn_div_scalar", k_div_scalar(ex, p))
SM_KERNEL_TAG(KCorrPartial, CorrPartialParams, "correlate_pairs", k_corr_partial(ex, p))
SM_KERNEL_TAG(KCorrFinish, CorrFinishParams, "correlate_finish", k_corr_finish(ex, p))
SM_KERNEL_TAG(KSerialNorm, SerialNormParams, "serial_norm", k_serial_norm(ex, p))
SM_KERNEL_TAG(KSpecNorm, SpecNormParams, "spec_norm", k_spec_norm(ex, p))
SM_KERNEL_TAG(KSumsqCand, SumsqCandParams, "spec_norm_cand", k_sumsq_cand(ex, p))
SM_KERNEL_TAG(KSumSpec, SumSpecParams, "spec_norm_sum", k_sum_spec(ex, p))
SM_KERNEL_TAG(KSpecRescale, SpecRescaleParams, "spec_rescale", k_spec_rescale(ex, p))
SM_KERNEL_TAG(KDftp, DftpParams, "dft_across_slices", k_dftp(ex, p))
SM_KERNEL_TAG(KDftpPairs, DftpParams, "dft_across_slices", k_dftp_pairs(ex, p))
SM_KERNEL_TAG(KTranspose, TransposeParams, "transpose", k_transpose(ex, p))
// (two instantiations each: signals x - base, and the slerp class of two spectrum planes)
SM_KERNEL_TAG_LB(KAtenPre, AtenPreParams, "aten_norm_pre", k_aten_pre<0>(ex, p), 256, 4)
SM_KERNEL_TAG_LB(KAtenPreC, AtenPreParams, "aten_norm_pre", k_aten_pre<1>(ex, p), 256, 4)
SM_KERNEL_TAG_LB(KAtenScan, AtenScanParams, "aten_norm_scan", k_aten_scan(ex, p), 256, 4)
SM_KERNEL_TAG_LB(KAtenRec, AtenRecParams, "aten_norm_rec", k_aten_rec(ex, p), 256, 4)
#ifndef SM_ATEN_PART_WAVES
#define SM_ATEN_PART_WAVES 3
#endif
SM_KERNEL_TAG_LB(KAtenPart16, AtenPartParams, "aten_norm_part", k_aten_part<ATEN_PART_RAW16>(ex, p), 256, SM_ATEN_PART_WAVES)   // 16 running summaries per thread
SM_KERNEL_TAG_LB(KAtenPart32, AtenPartParams, "aten_norm_part", k_aten_part<ATEN_PART_RAW32>(ex, p), 256, SM_ATEN_PART_WAVES)
SM_KERNEL_TAG_LB(KAtenPart, AtenPartParams, "aten_norm_part", k_aten_part<ATEN_PART_SIGNAL>(ex, p), 256, 2)
SM_KERNEL_TAG_LB(KAtenPartC, AtenPartParams, "aten_norm_part", k_aten_part<ATEN_PART_CLASS>(ex, p), 256, 2)
SM_KERNEL_TAG_LB(KAtenWalk, AtenWalkParams, "aten_norm_walk", k_aten_walk<0>(ex, p), 256, 4)
SM_KERNEL_TAG_LB(KAtenWalkC, AtenWalkParams, "aten_norm_walk", k_aten_walk<1>(ex, p), 256, 4)
SM_KERNEL_TAG_LB(KAtenFinish, AtenFinishParams, "aten_norm_finish", k_aten_finish(ex, p), 256, 4)
// Every kernel is instantiated in smhip_side.hip (one group per translation unit, -DSM_SIDE_GROUP=<g>) or, the
// static-plan transforms, in smhip_inst.hip; smhip_hip.hip holds host code only.  The build parallelises and a
// change to the host orchestration does not recompile a single kernel.
#define SM_SIDE_KERNELS_0(X) X(KAtenPre) X(KAtenPreC) X(KAtenScan) X(KAtenRec) X(KAtenPart16) X(KAtenPart32) X(KAtenPart) X(KAtenPartC) X(KAtenWalk) X(KAtenWalkC) X(KAtenFinish)
#define SM_SIDE_KERNELS_1(X) X(KF2R1) X(KI1R1) X(KPublish) X(KHist) X(KScan) X(KSelect2) X(KSelect2Cull) X(KBlendSel) \
    X(KSpecCheck) X(KSelect3) X(KReduceCand) X(KReduce) X(KSlerpConsts) X(KSumPartials) X(KClassEmf)
#define SM_SIDE_KERNELS_2(X) X(KDeltaNorms) X(KSumPartialsN) X(KBlend) X(KCombine) X(KExpand) X(KPack) X(KSplit) X(KJoin) \
    X(KCull) X(KAddition) X(KFnSums) X(KFnSlerpFin) X(KFnSlerpRows0) X(KFnSlerpRows1) X(KFnSlerpDen) X(KSumsqAny) X(KDivScalar) X(KCorrPartial) X(KCorrFinish) X(KSerialNorm) X(KSpecNorm) X(KSumsqCand) X(KSumSpec)       \
    X(KSpecRescale) X(KDftp) X(KDftpPairs) X(KTranspose)
#define SM_SIDE_GROUPS 7         // groups 3 - 6: the run-time planned (DynPlan) transform kernels

// ---- FFT planner ---------------------------------------------------------------
struct HostPlan {
    FftPlanDev dev;
    bool ok = false;
};

inline bool plan_radices(int N, int T, std::vector<int>& out) {
    out.clear();
    if (N == 1) { out.push_back(1); return true; }
    int rest = N, a = 0;
    while (rest % 2 == 0) { rest /= 2; ++a; }
    std::vector<int> odd;
    for (int p : {3, 5, 7, 11, 13})
        while (rest % p == 0) { rest /= p; odd.push_back(p); }
    if (rest != 1) return false;
    if (a > 0) {
        const int np = (a + 4) / 5;
        const int base = a / np, extra = a % np;
        for (int i = 0; i < np; ++i) out.push_back(1 << (base + (i < extra ? 1 : 0)));
    }
    for (int p : odd) out.push_back(p);
    if ((int)out.size() > MAX_PASSES) return false;
    for (int r : out) {
        const int nb = N / r;
        const int per = (nb + T - 1) / T;
        if (per * r > EMAX) return false;
    }
    return true;
}


inline bool plan_shape(int N, int& T, std::vector<int>& radices) {
    if (N < 1 || N > EMAX * 1024) return false;
    if (plan_shape_static(N, T, radices)) return true;
    int t0 = ((N + EMAX - 1) / EMAX + 63) / 64 * 64;
    if (t0 < 64) t0 = 64;
    for (T = t0; T <= 1024; T += 64)
        if (plan_radices(N, T, radices)) return true;
    return false;
}

// A column length the engine cannot plan (a prime factor > 13, or > 32768): R = p * M with M
// planned and even, p <= DFTP_MAX_P as small as possible (k_dftp).  p = 1: plain plan.
inline bool rough_split(int N, int& p, int& M) {
    int T;
    std::vector<int> rad;
    if (plan_shape(N, T, rad)) { p = 1; M = N; return true; }
    for (p = 2; p <= DFTP_MAX_P; ++p) {
        if (N % p) continue;
        M = N / p;
        if (M % 2 == 0 && plan_shape(M, T, rad)) return true;
    }
    return false;
}
// A ROW length the engine cannot plan: Bluestein's convolution on a power-of-two plan of L >= 2C - 1
// points (sm_bluestein.hpp).  0: too long.
inline int bluestein_len(int C) {
    if (C < 2) return 0;
    for (int L = 256; L <= EMAX * 1024; L *= 2)
        if (L >= 2 * C - 1) return L;
    return 0;
}
// [rows x cols] as the layer merge takes it: the row length wants a plan; the column length may
// be split (rough_split); a tensor that fits only the other way round is merged transposed; one whose
// lengths are both rough has its rows transformed by k_f1b / k_i2b
inline int shape_support(int rows, int cols, bool* transposed = nullptr) {
    int T, p, M;
    std::vector<int> rad;
    if (transposed) *transposed = false;
    if (rows < 1 || cols < 1) return 0;
    if (plan_shape(cols, T, rad) && rough_split(rows, p, M)) return 1;
    if (plan_shape(rows, T, rad) && rough_split(cols, p, M)) { if (transposed) *transposed = true; return 1; }
    if (bluestein_len(cols) && rough_split(rows, p, M)) return 1;
    if (bluestein_len(rows) && rough_split(cols, p, M)) { if (transposed) *transposed = true; return 1; }
    return 0;
}

struct Buffer {
    void* p = nullptr;
    size_t cap = 0;
};

struct ProfEntry { std::string name; uint64_t launches = 0; double ms = 0; };

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// what one pair merge needs besides its two inputs
struct PairOut {
    void* out = nullptr; int out_mode = OUT_F32;
    const void* base = nullptr; int base_dtype = DT_BF16;
    float post = 1.f;
    int ifft_policy = 1;      // 1: NaN -> 0 and Inf -> error on the inverse transform's output (functions.py:211-217)
};

template <class B>
class Pipeline {
  public:
    B be;
    std::string err;
    void* stream = nullptr;
    uint32_t debug_cand_cap = 0;      // test hook: clamp the candidate-list capacities
    uint32_t debug_sel_chunks = 0;    // test hook: steps per thread of the level-2 selection pass (0 = automatic)
    bool debug_flush_always = false;  // test hook: flush staged candidates after every round
    bool spec_cull = true;            // the blend pass speculates on the cull threshold's level-1 bin (k_select2<true>)
    size_t spec_min_bins = 1 << 20;   // ... from this many spectrum bins on (below, its two extra launches cost more than the pass)
    bool dftp_pairs_enabled = true;   // test hook: 0 = the generic k_dftp for every p
    int debug_force_split = 0;        // test hook: split a column length into this many row blocks even if it has a plan
    int sel_wgs_per_cu = 5;           // level-2 selection pass: work-groups per CU its grid is sized for
    bool safe_select = false;         // full-pass selection (no candidate lists): the retry mode after an overflow
    uint32_t noise_seed_base = 0;     // test hook "noise_seed": another realisation of the noise model (seed-sensitivity test)
    bool spectral_inter = true;       // K >= 3: intermediates of the tournament stay in the spectral domain
    bool f2s_pair = true;             // a pair of raw deltas: both single-signal column passes in one launch (test hook: 0 = two launches)
    size_t f2s_pair_max_bins = (size_t)64 << 20;   // ... for spectra up to this many bins
    bool f1_multi = true;             // rows_first: the row passes of all raw deltas in one launch (test hook: 0 = one launch each)
    bool fuse_spec_norm = true;       // their Parseval norm comes out of the cull selection and the role-a column pass
    float noise_sigma = 1.2e-7f;      // rounding-noise model for their culled bins (k_spec_rescale)

    explicit Pipeline(int device) : be(device) {}
    ~Pipeline() {
        for (auto& kv : plans_) if (kv.second.dev.tw) be.free((void*)kv.second.dev.tw);
        for (Buffer* b : {&cand_, &t1_, &small_, &tmpA_, &tmpB_, &tmpC_, &fullS_, &saveR_, &saveI_, &aten_, &emf_}) if (b->p) be.free(b->p);
        for (Buffer& b : pool_) if (b.p) be.free(b.p);
        if (mail_) be.free_host(mail_);
        for (Buffer& b : inter_) if (b.p) be.free(b.p);
        for (Buffer& b : rowspec_) if (b.p) be.free(b.p);
        for (Buffer& b : tr_) if (b.p) be.free(b.p);
        for (auto& kv : rough_tw_) if (kv.second) be.free(kv.second);
        for (auto& kv : blue_) { if (kv.second.chirp) be.free(kv.second.chirp); if (kv.second.filt) be.free(kv.second.filt); }
    }

    int fail(int code, const std::string& msg) { err = msg; return code; }

    // ---- plans ---------------------------------------------------------------
    int get_plan(int N, FftPlanDev& out) {
        auto it = plans_.find(N);
        if (it == plans_.end()) {
            HostPlan hp;
            int T;
            std::vector<int> rad;
            if (plan_shape(N, T, rad)) {
                hp.ok = true;
                hp.dev.N = N; hp.dev.T = T; hp.dev.npass = (int)rad.size();
                for (int i = 0; i < MAX_PASSES; ++i) hp.dev.radix[i] = i < (int)rad.size() ? rad[i] : 1;
                hp.dev.cx = plan_uses_cx(N) ? 1 : 0;
                hp.dev.lds_floats = hp.dev.cx ? (int)round_up((size_t)2 * (N + (N >> 4) + 1), 32)
                                              : (int)round_up((size_t)lpad(N) + 1, 32);
                std::vector<cf2> tw(N);
                for (int j = 0; j < N; ++j) {
                    const double ang = -2.0 * M_PI * (double)j / (double)N;
                    tw[j].x = (float)cos(ang); tw[j].y = (float)sin(ang);
                }
                void* d = be.alloc(sizeof(cf2) * N);
                if (!d) return fail(SMHIP_ERR_NOMEM, "twiddle alloc failed");
                be.h2d(d, tw.data(), sizeof(cf2) * N, stream);
                hp.dev.tw = (const cf2*)d;
            } else {
                hp.dev.tw = nullptr;
            }
            it = plans_.emplace(N, hp).first;
        }
        if (!it->second.ok) {
            char buf[256];
            snprintf(buf, sizeof buf, "unsupported transform length %d (a row length needs factors in {2,3,5,7,11,13} and <= %d; the layer merge "
                                       "also takes p * M column lengths, p <= %d)", N, EMAX * 1024, DFTP_MAX_P);
            return fail(SMHIP_ERR_SHAPE, buf);
        }
        out = it->second.dev;
        return SMHIP_OK;
    }

    // ---- workspace -------------------------------------------------------------
    int ensure(Buffer& b, size_t bytes) {
        if (b.cap >= bytes) return SMHIP_OK;
        if (b.p) { be.sync(stream); be.free(b.p); b.p = nullptr; b.cap = 0; }
        b.p = be.alloc(bytes);
        if (!b.p) return fail(SMHIP_ERR_NOMEM, "workspace allocation failed");
        b.cap = bytes;
        return SMHIP_OK;
    }
    size_t workspace_bytes() const {
        size_t t = cand_.cap + t1_.cap + small_.cap + tmpA_.cap + tmpB_.cap + tmpC_.cap + fullS_.cap + saveR_.cap + saveI_.cap + aten_.cap + emf_.cap;
        for (const Buffer& b : inter_) t += b.cap;
        for (const Buffer& b : rowspec_) t += b.cap;
        for (const Buffer& b : pool_) t += b.cap;
        for (const Buffer& b : tr_) t += b.cap;
        return t;
    }

    struct Geo {
        int R, C, Cb, pitch4, pitchG, ilv;
        size_t plane_floats;      // one plane, padded
        bool full;                // planes hold the full spectrum (weights 1)
        int Cw;                   // C passed to bin_weight (-1 in full mode)
        int fold;                 // 4: radix-4 column step folded into the row pass (k_f1q), else 1
        int batch;                // > 1: rank > 2 tensor = `batch` slices [R x C]; transforms per slice, statistics
                                  // over everything (the reference: fftn(dim=(-2,-1)), functions.py:58); Cb then
                                  // counts the bin columns of ALL slices (the planes hold them one after the other)
        size_t t1_slice;          // float4 of T1 per slice
        int rough;                // > 1: the `batch` slices are the row blocks of ONE [rough * R x C] tensor whose column
                                  // length the engine cannot plan; k_dftp combines them (see there)
    };
    // shapes whose column pass runs folded: a long column (its own plan needs >= 512 threads per
    // transform) over rows short enough for four of them to share a work-group
#ifndef SM_FOLD_MIN_ROWS
#define SM_FOLD_MIN_ROWS 14336
#endif
    int fold_min_rows = SM_FOLD_MIN_ROWS;
    bool fold_shape(int R, int C) const {
        if (R < fold_min_rows || !(R == 8192 || R == 14336 || R == 16384 || R == 28672)) return false;
        int T; std::vector<int> rad;
        if (!plan_shape_static(C, T, rad)) return false;
        return 4 * T <= 1024 && (C / T) % 2 == 0 && C % 8 == 0;
    }
    bool fold_enabled = true;
    Geo geo(int R, int C, bool full = false, bool allow_fold = false, int batch = 1) const {
        Geo g;
        g.R = R; g.C = C; g.full = full;
        g.batch = batch < 1 ? 1 : batch;
        g.fold = (allow_fold && fold_enabled && !full && g.batch == 1 && fold_shape(R, C) && !row_bluestein(C)) ? 4 : 1;
        g.Cb = full ? C : (C / 2 + 1) * g.batch;
        g.pitch4 = (int)round_up((size_t)(C / 2 + 1), 8);
        g.t1_slice = round_up((size_t)R, 8) * (size_t)g.pitch4;
        g.pitchG = g.pitch4;           // G: row pairs x bins, float4 (two rows' float2) per entry
        g.ilv = t1_interleave(R);
        g.plane_floats = round_up((size_t)g.Cb * R, 64);
        g.Cw = full ? -1 : C;
        g.rough = 1;
        return g;
    }
    // rows interleaved in the forward intermediate T1: the column pass then reads ilv*16
    // contiguous bytes per thread instead of 16 (its strided 16-byte reads ran at a third
    // of the HBM rate), the row pass pays with ilv work-groups sharing each line it writes
    // (measured on MI355X: pairs pay off from 8192 rows on, groups of 4 never)
#ifndef SM_T1_ILV
#define SM_T1_ILV 2
#endif
#ifndef SM_T1_ILV_MIN_ROWS
#define SM_T1_ILV_MIN_ROWS 8192
#endif
    static int t1_interleave(int R) { return t1_interleave_rows(R); }
    static constexpr int MAXGRID_PART = 1 << 20;
    int reserve(int R, int C, bool full = false, int batch = 1) {
        const Geo g = geo(R, C, full, false, batch);
        int rc;
        if ((rc = ensure(t1_, g.t1_slice * g.batch * sizeof(cf4)))) return rc;
        for (int i = 0; i < 4; ++i)
            if ((rc = ensure(pool_[pidx_[i]], g.plane_floats * sizeof(float)))) return rc;
        {   // candidate lists of the selection passes: ~1 % of the data is expected
            const size_t ck = std::max<size_t>(1 << 16, g.plane_floats / 4), cp = std::max<size_t>(1 << 15, g.plane_floats / 16);
            if ((rc = ensure(cand_, ck * 4 + cp * 16))) return rc;
            cap_keys_ = (uint32_t)std::min<size_t>(ck, 0xffffffffu); cap_pairs_ = (uint32_t)std::min<size_t>(cp, 0xffffffffu);
        }
        if (!small_.p) {
            if ((rc = ensure(small_, SMALL_BYTES))) return rc;
            be.memset(small_.p, 0, SMALL_BYTES, stream);
            if (!mail_) mail_ = (Mailbox*)be.alloc_host(sizeof(Mailbox));
            if (!mail_) return fail(SMHIP_ERR_NOMEM, "host mailbox");
        }
        return SMHIP_OK;
    }
    // small device block layout
    static constexpr size_t OFF_HIST = 0;                                  // u64[2048]
    static constexpr size_t OFF_SEL = OFF_HIST + 8 * (HIST1_BINS + 2 * HIST_LO_BINS);   // SelState[2]; hist1 | hist2 | hist3 before it
    static constexpr size_t OFF_CONSTS = OFF_SEL + 2 * sizeof(SelState);   // BlendConsts
    static constexpr size_t OFF_THR = OFF_CONSTS + 256;                    // float thr[4]
    static constexpr size_t OFF_FLAGS = OFF_THR + 64;                      // u32[8]
    static constexpr size_t OFF_CANDCTR = OFF_FLAGS + 32;                  // u32[4]: n_keys, n_pairs, overflow
    static constexpr size_t OFF_NORM2 = OFF_FLAGS + 64;                    // double[2]
    static constexpr size_t OFF_PART = OFF_NORM2 + 64;                     // double partials
    static constexpr size_t PART_DOUBLES = 4 * 65536 + 2 * 40000;
    static constexpr size_t OFF_PART_IM = OFF_PART + PART_DOUBLES * 8;     // sum w (Im a)^2 partials of the role-a producer
    static constexpr size_t PART_IM_DOUBLES = 131072;
    static constexpr size_t OFF_SPEC = OFF_PART_IM + PART_IM_DOUBLES * 8;   // u32 guess[8] (bin + 1, per cull slot), u32 flag
    static constexpr size_t SMALL_BYTES = OFF_SPEC + 64;
    unsigned long long* d_hist() { return (unsigned long long*)((char*)small_.p + OFF_HIST); }
    unsigned long long* d_hist2() { return d_hist() + HIST1_BINS; }
    unsigned long long* d_hist3() { return d_hist2() + HIST_LO_BINS; }
    SelState* d_sel(int i) { return (SelState*)((char*)small_.p + OFF_SEL) + i; }
    BlendConsts* d_consts() { return (BlendConsts*)((char*)small_.p + OFF_CONSTS); }
    float* d_thr(int i) { return (float*)((char*)small_.p + OFF_THR) + i; }
    uint32_t* d_flags() { return (uint32_t*)((char*)small_.p + OFF_FLAGS); }
    uint32_t* d_candctr() { return (uint32_t*)((char*)small_.p + OFF_CANDCTR); }
    CandLists cand_lists() {
        CandLists c;
        c.keys = (uint32_t*)cand_.p; c.pairs = (cf4*)((char*)cand_.p + (size_t)cap_keys_ * 4);
        c.cap_keys = cap_keys_; c.cap_pairs = cap_pairs_; c.counters = d_candctr();
        if (debug_cand_cap) { c.cap_keys = std::min(c.cap_keys, debug_cand_cap); c.cap_pairs = std::min(c.cap_pairs, debug_cand_cap); }
        return c;
    }
    double* d_norm2() { return (double*)((char*)small_.p + OFF_NORM2); }
    double* d_part() { return (double*)((char*)small_.p + OFF_PART); }
    double* d_part_im() { return (double*)((char*)small_.p + OFF_PART_IM); }
    uint32_t* d_spec_guess(int slot) { return (uint32_t*)((char*)small_.p + OFF_SPEC) + (slot < 0 ? 0 : slot > 7 ? 7 : slot); }
    uint32_t* d_spec_flag() { return (uint32_t*)((char*)small_.p + OFF_SPEC) + 8; }
    // The four working planes come from a pool of equally sized buffers: a pair merge whose
    // result stays in the spectral domain (K >= 3) keeps its Re R / Im a planes as they are - the
    // planes are detached from the working set and fresh ones take their place.
    float* plane(const Geo&, int i) { return (float*)pool_[pidx_[i]].p; }
    enum { P_REA = 0, P_IMA = 1, P_REB = 2, P_RER = 3 };
    int pool_acquire(size_t bytes) {
        int id = -1;
        for (size_t q = 0; q < pool_.size(); ++q) if (!pool_busy_[q]) { id = (int)q; break; }
        if (id < 0) { pool_.emplace_back(); pool_busy_.push_back(0); id = (int)pool_.size() - 1; }
        if (ensure(pool_[id], bytes)) return -1;
        pool_busy_[id] = 1;
        return id;
    }
    void pool_release(int id) { if (id >= 0) pool_busy_[id] = 0; }

    static bool is_static_plan(const FftPlanDev& pl) {
        bool found = false;
#define SM_IS_PLAN(...) if (plan_matches<__VA_ARGS__>(pl)) found = true;
        SM_STATIC_PLANS(SM_IS_PLAN)
#undef SM_IS_PLAN
        return found;
    }
    static int f2_bins_host(const FftPlanDev& pl) { return is_static_plan(pl) ? f2_bins_for(pl.T, pl.N) : 1; }
    static int i1_bins_host(const FftPlanDev& pl) {
        // two bins per work-group up to 512 threads; the 512-thread plans run one bin per
        // work-group (two independent groups per CU, as in the forward column pass)
        if (2 * pl.T > SM_F2_MAX_THREADS) return 1;
        return is_static_plan(pl) ? i1_bins_for(pl.T) : 2;
    }

    // launch the static-plan instantiation of a transform kernel when one matches
    template <template <class> class KT, class Params>
    void launch_fft(const FftPlanDev& pl, int grid, int block, size_t lds, const Params& p, bool allow_static = true) {
        bool done = !allow_static;
#define SM_TRY_PLAN(...)                                                     \
        if (!done && plan_matches<__VA_ARGS__>(pl)) {                        \
            be.template launch<KT<__VA_ARGS__>>(grid, block, lds, p, stream);\
            done = true;                                                     \
        }
        SM_STATIC_PLANS(SM_TRY_PLAN)
#undef SM_TRY_PLAN
        if (!done || !allow_static) be.template launch<KT<DynPlan>>(grid, block, lds, p, stream);
    }

    // ---- stages ------------------------------------------------------------------
    // chunks per thread such that a streaming kernel runs about `waves` work-groups per
    // CU in total: the per-work-group fixed costs (LDS histogram zero/flush, staging,
    // block reductions, atomics) are paid once per resident work-group, not per 8K elements
    static int pick_chunks(size_t units, int block, int min_chunks, int wgs_per_cu = 5) {
        const size_t target = (size_t)256 * wgs_per_cu;
        size_t c = (units + (size_t)block * target - 1) / ((size_t)block * target);
        if (c < (size_t)min_chunks) c = min_chunks;
        if (c > 4096) c = 4096;
        return (int)c;
    }
    static int stream_grid(size_t units, int block, int chunks) {
        const size_t per = (size_t)block * chunks;
        size_t g = (units + per - 1) / per;
        return (int)std::max<size_t>(g, 1);
    }
    static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
    static int vec4(const Geo& g) { return g.full ? (((size_t)g.R * g.C) % 4 == 0) : (g.R % 4 == 0); }

    // ---- a row length without a plan: Bluestein's convolution (sm_bluestein.hpp) ------------------
    bool debug_force_bluestein = false;      // test hook: k_f1b / k_i2b for planned row lengths too
    bool row_bluestein(int C) const {
        int T; std::vector<int> rad;
        if (debug_force_bluestein) return bluestein_len(C) > 0;
        return !plan_shape(C, T, rad) && bluestein_len(C) > 0;
    }
    struct BlueHost { void* chirp = nullptr; void* filt = nullptr; int L = 0; };
    // chirp[n] = exp(-i pi n^2 / C) and the spectrum of the convolution's filter, both from double precision
    int bluestein_tables(int C, BluesteinTab& out, FftPlanDev& plan) {
        const int L = bluestein_len(C);
        if (!L) return fail(SMHIP_ERR_SHAPE, "row length too long for the chirp-z transform");
        int rc = get_plan(L, plan);
        if (rc) return rc;
        auto it = blue_.find(C);
        if (it == blue_.end()) {
            std::vector<double> wr(C), wi(C);
            for (int n = 0; n < C; ++n) {
                const unsigned long long m = ((unsigned long long)n * (unsigned long long)n) % (2ull * (unsigned long long)C);
                const double ang = -M_PI * (double)m / (double)C;
                wr[n] = cos(ang); wi[n] = sin(ang);
            }
            std::vector<double> hr(L, 0.0), hi(L, 0.0);
            for (int m = 0; m < C; ++m) {
                hr[m] = wr[m]; hi[m] = -wi[m];
                if (m) { hr[L - m] = wr[m]; hi[L - m] = -wi[m]; }
            }
            // radix-2 transform of the filter on the host (L is a power of two)
            for (int i = 1, j = 0; i < L; ++i) {
                int bit = L >> 1;
                for (; j & bit; bit >>= 1) j ^= bit;
                j ^= bit;
                if (i < j) { std::swap(hr[i], hr[j]); std::swap(hi[i], hi[j]); }
            }
            for (int len = 2; len <= L; len <<= 1) {
                const int half = len / 2;
                for (int k = 0; k < half; ++k) {
                    const double ang = -2.0 * M_PI * (double)k / (double)len;
                    const double cr = cos(ang), ci = sin(ang);
                    for (int i = k; i < L; i += len) {
                        const int j = i + half;
                        const double tr = hr[j] * cr - hi[j] * ci, ti = hr[j] * ci + hi[j] * cr;
                        hr[j] = hr[i] - tr; hi[j] = hi[i] - ti;
                        hr[i] += tr; hi[i] += ti;
                    }
                }
            }
            std::vector<cf2> chirp(C), filt(L);
            for (int n = 0; n < C; ++n) { chirp[n].x = (float)wr[n]; chirp[n].y = (float)wi[n]; }
            for (int k = 0; k < L; ++k) { filt[k].x = (float)(hr[k] / L); filt[k].y = (float)(hi[k] / L); }
            BlueHost bh;
            bh.L = L;
            bh.chirp = be.alloc(sizeof(cf2) * C);
            bh.filt = be.alloc(sizeof(cf2) * L);
            if (!bh.chirp || !bh.filt) return fail(SMHIP_ERR_NOMEM, "chirp table alloc failed");
            be.h2d(bh.chirp, chirp.data(), sizeof(cf2) * C, stream);
            be.h2d(bh.filt, filt.data(), sizeof(cf2) * L, stream);
            be.sync(stream);                       // the host vectors die here
            it = blue_.emplace(C, bh).first;
        }
        out.chirp = (const cf2*)it->second.chirp; out.filt = (const cf2*)it->second.filt;
        return SMHIP_OK;
    }
    // the row plan of a tensor: its own, or the convolution's
    int get_row_plan(int C, FftPlanDev& plan, bool& blue, BluesteinTab& bt) {
        blue = row_bluestein(C);
        bt.chirp = nullptr; bt.filt = nullptr;
        return blue ? bluestein_tables(C, bt, plan) : get_plan(C, plan);
    }

    // exp(-2 pi i j / N) for a length without a plan (k_dftp's twiddles)
    int rough_twiddles(int N, const cf2** out) {
        auto it = rough_tw_.find(N);
        if (it == rough_tw_.end()) {
            std::vector<cf2> tw(N);
            for (int j = 0; j < N; ++j) {
                const double ang = -2.0 * M_PI * (double)j / (double)N;
                tw[j].x = (float)cos(ang); tw[j].y = (float)sin(ang);
            }
            void* d = be.alloc(sizeof(cf2) * N);
            if (!d) return fail(SMHIP_ERR_NOMEM, "twiddle alloc failed");
            be.h2d(d, tw.data(), sizeof(cf2) * N, stream);
            be.sync(stream);                       // the host vector dies here
            it = rough_tw_.emplace(N, d).first;
        }
        *out = (const cf2*)it->second;
        return SMHIP_OK;
    }
    // the p-point DFT across the slices of a rough-length tensor, in place on T1 / G
    int run_dftp(const Geo& g, void* buf, bool rowpair, int ilv, int pitch, size_t slice_stride, bool inverse) {
        DftpParams q;
        int rc = rough_twiddles(g.R * g.rough, &q.tw);
        if (rc) return rc;
        q.buf = (cf4*)buf; q.p = g.rough; q.M = g.R; q.units = rowpair ? g.R / 2 : g.R; q.rowpair = rowpair ? 1 : 0;
        q.ilv = ilv; q.pitch = pitch; q.ncols = g.C / 2 + 1; q.slice_stride = slice_stride; q.inverse = inverse ? 1 : 0;
        q.cols = DFTP_COLS;
        if (q.p <= DFTP_PAIR_MAX_P && dftp_pairs_enabled) {
            q.cols = dftp_pairs_cols(q.p);
            const int ntp = (q.ncols + q.cols - 1) / q.cols;
            // (+ one zero row of Wh and 4 * 16 entries: the unrolled units past H read there)
            const size_t ldsp = LDS_SCRATCH_FLOATS * 4 + dftp_pairs_lds_bytes(q.p) + ((size_t)(q.p / 2 + 1) + 64) * sizeof(cf2);
            be.template launch<KDftpPairs>((q.units + DFTP_UNITS - 1) / DFTP_UNITS * ntp, 256, ldsp, q, stream);
            return SMHIP_OK;
        }
        const int ntiles = (q.ncols + DFTP_COLS - 1) / DFTP_COLS;
        const size_t lds = LDS_SCRATCH_FLOATS * 4 + (size_t)q.p * DFTP_COLS * sizeof(cf4) + (size_t)q.p * sizeof(cf2);
        be.template launch<KDftp>(q.units * ntiles, 256, lds, q, stream);
        return SMHIP_OK;
    }

    // F1: T1 <- row spectra of (a, b); per-group sums of squares land in d_part()
    int run_f1(const Geo& g, const SigDesc& a, const SigDesc& b, int& grid_out) {
        F1Params p;
        bool blue; BluesteinTab bt;
        int rc = get_row_plan(g.C, p.plan, blue, bt);
        if (rc) return rc;
        p.a = a; p.b = b; p.R = g.R; p.C = g.C; p.Cb = g.C / 2 + 1; p.pitch4 = g.pitch4; p.ilv = g.ilv;
        p.sigs.n = 0;
        p.fuse.prefix = nullptr; p.fuse.grp = nullptr; p.fuse.nchunks = 0; p.fuse.sig0 = 0; p.fuse_rowpair = 0;
        p.nb = blue ? 1 : std::max(1, 256 / p.plan.T);
        p.row_stride = (size_t)g.C;
        p.vec = !blue && (g.C % 8 == 0) && aligned16(a.x) && aligned16(a.base) && aligned16(b.x) && aligned16(b.base);
        if (fused_.armed && p.vec && g.batch == 1 && g.rough == 1) { p.fuse = fused_.af; p.fuse_rowpair = 0; fused_.used = true; }
        p.t1 = (cf4*)t1_.p;
        p.partials = d_part();
        p.R2 = 0; p.rowpair = 0; p.Rcol = g.R; p.twR = nullptr; p.slab_elems = fold_slab_elems(g);
        if (g.fold == 4) {
            if (!p.vec) return fail(SMHIP_ERR_ARG, "internal: folded row pass on unaligned input");
            FftPlanDev colp;
            if ((rc = get_plan(g.R, colp))) return rc;
            p.R2 = g.R / 4; p.twR = colp.tw; p.ilv = FOLD_ILV; p.nb = 4;
            const int gridq = (int)round_up((size_t)p.R2, 8 * p.ilv);
            if ((size_t)gridq * 2 > PART_DOUBLES) return fail(SMHIP_ERR_SHAPE, "too many rows");
            launch_fft<KF1Q>(p.plan, gridq, 4 * p.plan.T, (LDS_SCRATCH_FLOATS + (size_t)4 * p.plan.lds_floats) * 4, p);
            grid_out = gridq;
            return SMHIP_OK;
        }
        const int xg = p.ilv > p.nb ? p.ilv / p.nb : 1;
        // the row blocks of a split column length are one tensor: their rows go in one launch when the
        // blocks lie back to back in T1
        const bool joint = joint_rows(g);
        const int nlaunch = joint ? 1 : g.batch;
        if (joint) p.R = g.R * g.rough;
        const int grid = (int)round_up((size_t)(p.R + p.nb - 1) / p.nb, 8 * xg);
        if ((size_t)grid * 2 * nlaunch > PART_DOUBLES) return fail(SMHIP_ERR_SHAPE, "too many rows");
        const size_t lds = (LDS_SCRATCH_FLOATS + (size_t)p.nb * p.plan.lds_floats) * 4;
        for (int bi = 0; bi < nlaunch; ++bi) {          // one launch per slice of a rank > 2 tensor
            p.a = sig_at(a, (size_t)bi * g.R * g.C); p.b = sig_at(b, (size_t)bi * g.R * g.C);
            p.t1 = (cf4*)t1_.p + (size_t)bi * g.t1_slice;
            p.partials = d_part() + (size_t)bi * 2 * grid;
            if (blue) { F1BParams q; q.f = p; q.bt = bt; launch_fft<KF1B>(p.plan, grid, p.plan.T, lds, q); }
            else launch_fft<KF1>(p.plan, grid, p.nb * p.plan.T, lds, p, p.vec != 0);
        }
        grid_out = grid * nlaunch;
        if (g.rough > 1) return run_dftp(g, t1_.p, false, p.ilv, g.pitch4, g.t1_slice, false);
        return SMHIP_OK;
    }
    // ---- one signal (rounds >= 2: the pair's other input stayed spectral) ---------------------
    static size_t dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }
    static SigDesc sig_at(const SigDesc& s, size_t elems) {        // the signal `elems` elements further on (next slice of a batch)
        SigDesc r = s;
        if (r.x) r.x = (const char*)r.x + elems * dt_size(s.dtype);
        if (r.base) r.base = (const char*)r.base + elems * dt_size(s.dtype);
        return r;
    }
    // rows (2m, 2m+1) of ONE signal as the two operands of the two-for-one row transform:
    // T1[m][k] = (spectrum of row 2m, spectrum of row 2m+1).  R must be even.
    // float4 per k1 slab of the folded T1 (R/4 rows of one real pitch)
    static size_t fold_slab_elems(const Geo& g, bool rowpair = false) {
        return round_up((size_t)g.R / (rowpair ? 8 : 4), 8) * (size_t)g.pitch4;
    }
    static bool joint_rows(const Geo& g) { return g.rough > 1 && g.R % 16 == 0; }
    // multi (optional): the row passes of multi->n signals in ONE launch (F1Sigs: x, base, t1 filled in by the caller;
    // partials are laid out here, *grid_out work-groups per signal); returns SMHIP_ERR_ARG untouched when the geometry
    // needs a launch per slice or the chirp-z kernel - the caller then goes signal by signal
    int run_f1_rowpairs(const Geo& g, const SigDesc& sig, void* t1buf = nullptr, double* partials = nullptr, int* grid_out = nullptr,
                        F1Sigs* multi = nullptr) {
        F1Params p;
        bool blue; BluesteinTab bt;
        int rc = get_row_plan(g.C, p.plan, blue, bt);
        if (rc) return rc;
        SigDesc a = sig, b = sig;
        b.x = (const char*)sig.x + (size_t)g.C * dt_size(sig.dtype);
        if (sig.base) b.base = (const char*)sig.base + (size_t)g.C * dt_size(sig.dtype);
        p.a = a; p.b = b; p.R = g.R / 2; p.C = g.C; p.Cb = g.C / 2 + 1; p.pitch4 = g.pitch4; p.ilv = F2S_ILV;
        p.sigs.n = 0;
        p.fuse.prefix = nullptr; p.fuse.grp = nullptr; p.fuse.nchunks = 0; p.fuse.sig0 = 0; p.fuse_rowpair = 0;
        const int nm = multi ? multi->n : 1;
        if (multi && (blue || nm < 2 || nm > F1_MAX_SIGS)) return SMHIP_ERR_ARG;
        p.nb = blue ? 1 : std::max(1, 256 / p.plan.T);
        p.row_stride = (size_t)2 * g.C;
        p.vec = !blue && (g.C % 8 == 0) && aligned16(a.x) && aligned16(a.base) && aligned16(b.x) && aligned16(b.base);
        if (fused_.armed && p.vec && g.batch == 1 && g.rough == 1) {
            p.fuse = fused_.af; p.fuse.sig0 = multi ? 0 : fused_.next_sig; p.fuse_rowpair = 1; fused_.used = true;
        }
        p.t1 = (cf4*)(t1buf ? t1buf : t1_.p);
        p.partials = partials ? partials : d_part();
        p.R2 = 0; p.rowpair = 0; p.Rcol = g.R; p.twR = nullptr; p.slab_elems = fold_slab_elems(g, true);
        if (g.fold == 4) {
            if (!p.vec) return fail(SMHIP_ERR_ARG, "internal: folded row pass on unaligned input");
            FftPlanDev colp;
            if ((rc = get_plan(g.R, colp))) return rc;
            p.R2 = g.R / 8; p.rowpair = 1; p.twR = colp.tw; p.ilv = F2S_ILV; p.nb = 4;      // units = row pairs
            const int gridq = (int)round_up((size_t)p.R2, 8 * p.ilv);
            if (multi) {
                if ((size_t)gridq * 2 * nm > PART_DOUBLES) return SMHIP_ERR_ARG;
                for (int i = 0; i < nm; ++i) multi->partials[i] = p.partials + (size_t)i * 2 * gridq;
                multi->b_off = (long long)((size_t)g.C * dt_size(sig.dtype));
                p.sigs = *multi;
            }
            launch_fft<KF1Q>(p.plan, gridq * nm, 4 * p.plan.T, (LDS_SCRATCH_FLOATS + (size_t)4 * p.plan.lds_floats) * 4, p);
            if (grid_out) *grid_out = gridq;
            return SMHIP_OK;
        }
        const int xgs = p.ilv > p.nb ? p.ilv / p.nb : 1;
        const bool joint = joint_rows(g);
        const int nlaunch = joint ? 1 : g.batch;
        if (joint) p.R = g.R * g.rough / 2;
        const int grid = (int)round_up((size_t)(p.R + p.nb - 1) / p.nb, 8 * xgs);
        if (multi && (nlaunch != 1 || g.rough > 1 || (size_t)grid * 2 * nm > PART_DOUBLES || !p.vec)) return SMHIP_ERR_ARG;
        if ((size_t)grid * 2 * nlaunch > PART_DOUBLES) return fail(SMHIP_ERR_SHAPE, "too many rows");
        const size_t lds = (LDS_SCRATCH_FLOATS + (size_t)p.nb * p.plan.lds_floats) * 4;
        cf4* const t1base = p.t1;
        double* const pbase = p.partials;
        if (multi) {
            for (int i = 0; i < nm; ++i) multi->partials[i] = pbase + (size_t)i * 2 * grid;
            multi->b_off = (long long)((size_t)g.C * dt_size(sig.dtype));
            p.sigs = *multi;
            launch_fft<KF1>(p.plan, grid * nm, p.nb * p.plan.T, lds, p, true);
            if (grid_out) *grid_out = grid;
            return SMHIP_OK;
        }
        for (int bi = 0; bi < nlaunch; ++bi) {
            p.a = sig_at(a, (size_t)bi * g.R * g.C); p.b = sig_at(b, (size_t)bi * g.R * g.C);
            p.t1 = t1base + (size_t)bi * (g.t1_slice / 2);
            p.partials = pbase + (size_t)bi * 2 * grid;
            if (blue) { F1BParams q; q.f = p; q.bt = bt; launch_fft<KF1B>(p.plan, grid, p.plan.T, lds, q); }
            else launch_fft<KF1>(p.plan, grid, p.nb * p.plan.T, lds, p, p.vec != 0);
        }
        if (grid_out) *grid_out = grid * nlaunch;
        if (g.rough > 1) return run_dftp(g, t1base, true, F2S_ILV, g.pitch4, g.t1_slice / 2, false);
        return SMHIP_OK;
    }
    // im_parts (optional, role a): the kernel also leaves sum w (Im a)^2 per work-group in d_part_im();
    // *im_parts = how many
    // second (optional): the pair's OTHER input in the same launch (its T1, role, scale); needs g.batch == 1
    struct F2SSecond { const void* t1; bool role_a; float scale; };
    int run_f2s(const Geo& g, bool role_a, float scale, bool hist, const void* t1buf = nullptr, int* im_parts = nullptr,
                const F2SSecond* second = nullptr) {
        F2SParams p;
        p.two = 0; p.t1_2 = nullptr; p.role_a_2 = 0; p.scale_2 = 0.f; p.re_2 = nullptr;
        if (second) {
            if (g.batch != 1 || second->role_a == role_a) return fail(SMHIP_ERR_ARG, "internal: paired column pass");
            p.two = 1; p.t1_2 = (const cf4*)second->t1; p.role_a_2 = second->role_a ? 1 : 0; p.scale_2 = second->scale;
            p.re_2 = plane(g, second->role_a ? P_REA : P_REB);
        }
        const int nsg = second ? 2 : 1;
        const int Rt = g.R / g.fold;                 // transform length (R2 on the folded path)
        int rc = get_plan(Rt, p.plan);
        if (rc) return rc;
        p.t1 = (const cf4*)(t1buf ? t1buf : t1_.p); p.pitch4 = g.pitch4 * g.fold; p.R = Rt; p.C = g.C;
        p.Cb = g.fold == 4 ? 4 * g.pitch4 : g.C / 2 + 1;
        p.slab = g.pitch4; p.Cb_real = g.C / 2 + 1; p.Rfull = g.R; p.slab_elems = fold_slab_elems(g, true);
        p.role_a = role_a ? 1 : 0; p.scale = scale;
        p.re = plane(g, role_a ? P_REA : P_REB); p.im = plane(g, P_IMA);
        p.hist = hist ? d_hist() : nullptr;
        const bool st = is_static_plan(p.plan);
        const int G = st ? (f2_nsig_for(p.plan.T) == 2 ? 2 * f2_bins_for(p.plan.T, p.plan.N) : 1) : 1;
        const int xg = G >= 8 ? 1 : 8 / G;
        const int grid = (int)round_up((size_t)(p.Cb + G - 1) / G, 8 * xg) * nsg;
        const size_t lds = (LDS_SCRATCH_FLOATS + (size_t)G * p.plan.lds_floats + HIST1_BINS) * 4;
        const bool any_a = role_a || (second && second->role_a);
        const bool want_im = im_parts && any_a && (size_t)grid * g.batch <= PART_IM_DOUBLES;
        if (im_parts) *im_parts = want_im ? grid * g.batch : -1;
        p.im_partials = want_im ? d_part_im() : nullptr;
        if (g.fold == 4) { launch_fft<KF2SQ>(p.plan, grid, G * p.plan.T, lds, p); return SMHIP_OK; }
        // every slice in one launch (im_partials is indexed by the global work-group id)
        if (g.batch > 1) { p.sl.wgs = grid; p.sl.t1_stride = g.t1_slice / 2; p.sl.plane_stride = (size_t)(g.C / 2 + 1) * g.R; }
        launch_fft<KF2S>(p.plan, grid * g.batch, G * p.plan.T, lds, p);
        return SMHIP_OK;
    }
    // sum over the full spectrum of |R_culled|^2 of the planes (re, im) -> (sum_re, sum_im), one sync
    void run_spec_norm(const Geo& g, const float* re, const float* im, const float* thr, double& sre, double& sim) {
        const size_t total = (size_t)g.Cb * g.R;
        SpecNormParams q;
        q.re = re; q.im = im; q.R = g.R; q.C = g.Cw; q.Cb = g.Cb; q.vec4 = vec4(g); q.thr = thr;
        q.chunks = pick_chunks((total + 3) / 4, 256, 8, 8);
        int grid = stream_grid((total + 3) / 4, 256, q.chunks);
        while ((size_t)grid * 2 > PART_DOUBLES) { q.chunks *= 2; grid = stream_grid((total + 3) / 4, 256, q.chunks); }
        q.partials = d_part();
        be.template launch<KSpecNorm>(grid, 256, LDS_SCRATCH_FLOATS * 4, q, stream);
        SumPartialsParams sp;
        sp.partials = d_part(); sp.nparts = grid; sp.out = mail_->norm2;
        be.template launch<KSumPartials>(1, 1024, LDS_SCRATCH_FLOATS * 4, sp, stream);
        // the cull threshold travels with the intermediate as a VALUE (d_thr(1) is reused by the next pair)
        PublishParams pp;
        pp.flags = d_flags(); pp.thr = d_thr(0); pp.consts = d_consts(); pp.mail = mail_; pp.zero_flags = nullptr;
        be.template launch<KPublish>(1, 64, LDS_SCRATCH_FLOATS * 4, pp, stream);
        be.sync(stream);
        sre = mail_->norm2[0]; sim = mail_->norm2[1];
    }
    void run_spec_rescale(const Geo& g, const float* re, const float* im, float thr, float scale, bool role_a, bool hist,
                          int* im_parts = nullptr) {
        const size_t total = (size_t)g.Cb * g.R;
        SpecRescaleParams q;
        q.re = re; q.im = role_a ? im : nullptr;
        q.dre = plane(g, role_a ? P_REA : P_REB); q.dim = plane(g, P_IMA);
        q.R = g.R; q.C = g.Cw; q.Cb = g.Cb; q.vec4 = vec4(g);
        q.thr = thr; q.scale = scale; q.sigma = noise_sigma; q.seed = ++noise_seed_;
        q.hist = hist ? d_hist() : nullptr;
        q.chunks = pick_chunks((total + 3) / 4, 256, 8, 8);
        const int grid = stream_grid((total + 3) / 4, 256, q.chunks);
        const bool want_im = im_parts && role_a && (size_t)grid <= PART_IM_DOUBLES;
        if (im_parts) *im_parts = want_im ? grid : -1;
        q.im_partials = want_im ? d_part_im() : nullptr;
        be.template launch<KSpecRescale>(grid, 256, (LDS_SCRATCH_FLOATS + HIST1_BINS) * 4, q, stream);
    }
    // the Parseval sums of a pair result that stays spectral, WITHOUT a pass of their own: sum w Re R_c^2
    // came out of the cull selection pass (`sel_parts` work-groups, + the candidates settled here), sum w
    // Im a^2 out of the role-a producer (`im_parts`).  One sync, as run_spec_norm.
    void run_spec_norm_fused(int sel_parts, int im_parts, double& sre, double& sim) {
        SumsqCandParams sc;
        sc.cand = cand_lists(); sc.thr = d_thr(1); sc.partials = d_part() + 4 * (size_t)sel_parts;
        be.template launch<KSumsqCand>(CAND_GRID, 256, LDS_SCRATCH_FLOATS * 4, sc, stream);
        SumSpecParams sp;
        sp.part4 = d_part(); sp.n4 = sel_parts + CAND_GRID; sp.part1 = d_part_im(); sp.n1 = im_parts; sp.out = mail_->norm2;
        sp.zero_u32 = d_candctr(); sp.zero_u32_count = 3;
        be.template launch<KSumSpec>(1, 1024, LDS_SCRATCH_FLOATS * 4, sp, stream);
        PublishParams pp;
        pp.flags = d_flags(); pp.thr = d_thr(0); pp.consts = d_consts(); pp.mail = mail_; pp.zero_flags = nullptr;
        be.template launch<KPublish>(1, 64, LDS_SCRATCH_FLOATS * 4, pp, stream);
        be.sync(stream);
        sre = mail_->norm2[0]; sim = mail_->norm2[1];
    }

    void read_norms(int grid, double& na, double& nb) {
        SumPartialsParams sp;
        sp.partials = d_part(); sp.nparts = grid; sp.out = mail_->norm2;     // host-mapped: no copy, one sync
        be.template launch<KSumPartials>(1, 1024, LDS_SCRATCH_FLOATS * 4, sp, stream);
        be.sync(stream);
        na = std::sqrt(mail_->norm2[0]); nb = std::sqrt(mail_->norm2[1]);
    }

    int run_f2(const Geo& g, float scale0, float scale1, int swap, bool hist) {
        F2Params p;
        int rc = get_plan(g.R, p.plan);
        if (rc) return rc;
        if (g.R == 1) {
            p.slab = g.pitch4; p.Cb_real = g.C / 2 + 1; p.Rfull = 1; p.slab_elems = 0;
            p.t1 = (const cf4*)t1_.p; p.pitch4 = g.pitch4; p.ilv = 1; p.R = 1; p.C = g.C; p.Cb = g.C / 2 + 1; p.nsig = 2;
            p.swap = swap; p.scale[0] = scale0; p.scale[1] = scale1;
            p.reA = plane(g, P_REA); p.imA = plane(g, P_IMA); p.reB = plane(g, P_REB);
            p.hist = hist ? d_hist() : nullptr;
            be.template launch<KF2R1>(std::max(1, std::min(64, (p.Cb + 255) / 256)), 256, (LDS_SCRATCH_FLOATS + HIST1_BINS) * 4, p, stream);
            return SMHIP_OK;
        }
        p.slab = g.pitch4; p.Cb_real = g.C / 2 + 1; p.Rfull = g.R; p.slab_elems = fold_slab_elems(g);
        if (g.fold == 4) {
            // R2-point transforms on 4 * pitch4 virtual columns (k1, bin); see k_f1q
            if ((rc = get_plan(g.R / 4, p.plan))) return rc;
            p.t1 = (const cf4*)t1_.p; p.pitch4 = 4 * g.pitch4; p.ilv = FOLD_ILV; p.R = g.R / 4; p.C = g.C; p.Cb = 4 * g.pitch4;
            p.nsig = 2; p.swap = swap; p.scale[0] = scale0; p.scale[1] = scale1;
            p.reA = plane(g, P_REA); p.imA = plane(g, P_IMA); p.reB = plane(g, P_REB);
            p.hist = hist ? d_hist() : nullptr;
            const int binsq = f2_bins_host(p.plan);
            const int gridq = (int)round_up((size_t)(p.Cb + binsq - 1) / binsq, 64);
            const size_t ldsq = (LDS_SCRATCH_FLOATS + (size_t)2 * binsq * p.plan.lds_floats + HIST1_BINS) * 4;
            launch_fft<KF2Q>(p.plan, gridq, 2 * binsq * p.plan.T, ldsq, p);
            return SMHIP_OK;
        }
        p.t1 = (const cf4*)t1_.p; p.pitch4 = g.pitch4; p.ilv = g.ilv; p.R = g.R; p.C = g.C; p.Cb = g.C / 2 + 1;
        // A and B of a bin in one work-group up to 512 threads; longer columns (T = 512: 14336,
        // 16384 rows) run one signal per work-group: two independent 512-thread groups per CU
        // overlap their phases, which beats the 16-byte reads of one lock-step 1024-thread group
        p.nsig = f2_nsig_for(p.plan.T);
        p.swap = swap; p.scale[0] = scale0; p.scale[1] = scale1;
        p.reA = plane(g, P_REA); p.imA = plane(g, P_IMA); p.reB = plane(g, P_REB);
        p.hist = hist ? d_hist() : nullptr;
        const int bins = p.nsig == 2 ? f2_bins_host(p.plan) : 1;
        const int ngroups = p.nsig == 2 ? 2 * bins : 1;
        const int grid = p.nsig == 2 ? (int)round_up((size_t)(p.Cb + bins - 1) / bins, 64) : (int)round_up((size_t)p.Cb * 2, 128);
        const size_t lds = (LDS_SCRATCH_FLOATS + (size_t)ngroups * p.plan.lds_floats + HIST1_BINS) * 4;
        if (g.batch > 1) { p.sl.wgs = grid; p.sl.t1_stride = g.t1_slice; p.sl.plane_stride = (size_t)(g.C / 2 + 1) * g.R; }
        launch_fft<KF2>(p.plan, grid * g.batch, ngroups * p.plan.T, lds, p);     // every slice in one launch
        return SMHIP_OK;
    }

    // exact k-th smallest of |X| (and |Y|) with bin multiplicities -> *thr_out.
    // Level 1 (11 bits) may already sit in d_hist() (fused into F2 / blend).  Fast path:
    // level 2 (10 bits) is one streaming pass that resolves level 1 itself, compacts the
    // ~1 % candidate keys and (fuse_reduce) takes the slerp-class sums; level 3 runs on the
    // candidate list and resolves level 2 itself; one single-work-group scan finishes.
    // A candidate-list overflow sets a sticky flag; the caller then redoes the work with
    // safe_select = true: three plain histogram passes, one scan after each.
    static constexpr int CAND_GRID = 256;
    // spec_slot >= 0: X came out of run_blend_spec(): k_spec_check decides whether its speculative level-2 work
    // stands; the plain pass below then returns at once (same grid, so the partials are the same either way)
    void run_select(const Geo& g, const float* X, const float* Y, unsigned long long rank, bool level1_done, float* thr_out,
                    bool fuse_reduce = false, int* nparts_out = nullptr, bool sumsq = false, int spec_slot = -1) {
        const size_t total = (size_t)g.Cb * g.R;
        HistParams h;
        h.X = X; h.Y = Y; h.R = g.R; h.C = g.Cw; h.Cb = g.Cb; h.vec4 = vec4(g); h.sel = d_sel(0);
        h.hist = d_hist(); h.chunks = pick_chunks((total + 3) / 4, 256, 8); h.only_if = nullptr;
        const int hgrid = stream_grid((total + 3) / 4, 256, h.chunks);
        const size_t hlds = (LDS_SCRATCH_FLOATS + HIST1_BINS) * 4;
        const size_t scan_lds = (LDS_SCRATCH_FLOATS + 2 * 256 + 32) * 4;
        ScanParams s;
        s.hist = d_hist(); s.sel = d_sel(0); s.value_out = thr_out; s.zero_also = nullptr; s.zero_count = 0;
        s.zero_u32 = nullptr; s.zero_u32_count = 0;
        if (!level1_done) { h.level = 1; be.template launch<KHist>(hgrid, 256, hlds, h, stream); }
        if (nparts_out) *nparts_out = 0;
        if (safe_select) {
            s.nbins = HIST1_BINS; s.shift = 11; s.final_level = 0; s.init = 1; s.rank_init = rank;
            be.template launch<KScan>(1, 256, scan_lds, s, stream);
            for (int level = 2; level <= 3; ++level) {
                h.level = level;
                be.template launch<KHist>(hgrid, 256, hlds, h, stream);
                s.nbins = HIST_LO_BINS; s.shift = 10; s.final_level = (level == 3); s.init = 0;
                be.template launch<KScan>(1, 256, scan_lds, s, stream);
            }
            return;
        }
        // (the list counters n_keys, n_pairs, overflow are zero here: the final scan clears them)
        Select2Params q;
        q.X = X; q.Y = Y; q.R = g.R; q.C = g.Cw; q.Cb = g.Cb; q.vec4 = vec4(g); q.sel = d_sel(0);
        q.hist1 = d_hist(); q.rank = rank; q.hist = d_hist2();
        q.cand = cand_lists(); q.fuse_reduce = (fuse_reduce && Y) ? 1 : 0; q.partials = d_part();
        q.sumsq = (sumsq && !Y) ? 1 : 0;
        q.flush_always = debug_flush_always ? 1 : 0;
        q.chunks = debug_sel_chunks ? (int)debug_sel_chunks : pick_chunks((total + 3) / 4, 256, 8, sel_wgs_per_cu);   // resident work-groups per CU: one round, no tail
        int grid2 = stream_grid((total + 3) / 4, 256, q.chunks);
        while ((size_t)(grid2 + CAND_GRID) * 4 > PART_DOUBLES) { q.chunks *= 2; grid2 = stream_grid((total + 3) / 4, 256, q.chunks); }
        const size_t lds2 = (LDS_SCRATCH_FLOATS + HIST_LO_BINS + 8 + STAGE_KEYS) * 4 + (size_t)STAGE_PAIRS * sizeof(cf4);
        q.skip = nullptr; q.blendB = nullptr; q.blendR = nullptr; q.consts = nullptr; q.t_sum = 0.f; q.hist1_out = nullptr; q.guess = nullptr;
        if (spec_slot >= 0) {
            SpecCheckParams sc;
            sc.hist1 = d_hist(); sc.rank = rank; sc.guess = d_spec_guess(spec_slot); sc.flag = d_spec_flag();
            sc.sel = d_sel(0); sc.hist2 = d_hist2(); sc.counters = d_candctr();
            be.template launch<KSpecCheck>(1, 256, LDS_SCRATCH_FLOATS * 4 + (256 + 16) * 8 + 16, sc, stream);
            q.skip = d_spec_flag();
            be.template launch<KSelect2Cull>(grid2, 256, lds2, q, stream);
        } else {
            be.template launch<KSelect2>(grid2, 256, lds2, q, stream);
        }

        Select3Params t3;
        t3.cand = q.cand; t3.sel = d_sel(0); t3.hist2 = d_hist2(); t3.hist = d_hist3();
        be.template launch<KSelect3>(CAND_GRID, 256, (LDS_SCRATCH_FLOATS + HIST_LO_BINS + 8 + 2 * 256 + 32) * 4, t3, stream);

        s.hist = d_hist3(); s.sel = d_sel(1); s.nbins = HIST_LO_BINS; s.shift = 10; s.final_level = 1; s.init = 0;
        s.zero_also = d_hist(); s.zero_count = HIST1_BINS + HIST_LO_BINS;      // hist1 and hist2 (hist3 is s.hist)
        // the list counters are cleared by the last kernel that reads them: this scan, or (when
        // the slerp sums come from the lists) slerp_consts after the candidate reduction
        if (!q.fuse_reduce && !q.sumsq) { s.zero_u32 = d_candctr(); s.zero_u32_count = 3; }   // not the sticky overflow word
        be.template launch<KScan>(1, 256, scan_lds, s, stream);

        if (q.sumsq && nparts_out) *nparts_out = grid2;          // run_spec_norm_fused() finishes (and clears the counters)
        if (q.fuse_reduce) {
            ReduceCandParams rc;
            rc.cand = q.cand; rc.thr = thr_out; rc.partials = d_part() + 4 * (size_t)grid2;
            be.template launch<KReduceCand>(CAND_GRID, 256, LDS_SCRATCH_FLOATS * 4, rc, stream);
            if (nparts_out) *nparts_out = grid2 + CAND_GRID;
        }
    }

    void run_blend(const Geo& g, int mode, int agreement, float t, float t_sum, bool hist) {
        const size_t total = (size_t)g.Cb * g.R;
        BlendParams b;
        b.reA = plane(g, P_REA); b.reB = plane(g, P_REB); b.reR = plane(g, P_RER);
        b.R = g.R; b.C = g.Cw; b.Cb = g.Cb; b.vec4 = vec4(g);
        b.mode = mode; b.agreement = agreement; b.t = t; b.t_sum = t_sum;
        b.consts = d_consts(); b.hist = hist ? d_hist() : nullptr; b.chunks = pick_chunks((total + 3) / 4, 256, 8, 8);
        be.template launch<KBlend>(stream_grid((total + 3) / 4, 256, b.chunks), 256, (LDS_SCRATCH_FLOATS + HIST1_BINS) * 4, b, stream);
    }

    // SLERP blend + the cull selection's level-2 pass on a guessed level-1 bin, in one sweep (k_select2<true>);
    // grid and element order are those of the plain selection pass that run_select(..., spec_slot) keeps as fallback
    void run_blend_spec(const Geo& g, float t_sum, bool sumsq, int spec_slot) {
        const size_t total = (size_t)g.Cb * g.R;
        Select2Params q;
        q.X = plane(g, P_REA); q.Y = nullptr; q.R = g.R; q.C = g.Cw; q.Cb = g.Cb; q.vec4 = vec4(g); q.sel = d_sel(0);
        q.hist1 = nullptr; q.rank = 0; q.hist = d_hist2();
        q.cand = cand_lists(); q.fuse_reduce = 0; q.partials = d_part(); q.sumsq = sumsq ? 1 : 0;
        q.flush_always = debug_flush_always ? 1 : 0;
        q.chunks = cull_select_chunks(total);
        q.skip = nullptr;
        q.blendB = plane(g, P_REB); q.blendR = plane(g, P_RER); q.consts = d_consts(); q.t_sum = t_sum;
        q.hist1_out = d_hist(); q.guess = d_spec_guess(spec_slot);
        const int grid = stream_grid((total + 3) / 4, 256, q.chunks);
        const size_t lds = (LDS_SCRATCH_FLOATS + HIST_LO_BINS + 8 + STAGE_KEYS + HIST1_BINS) * 4;
        be.template launch<KBlendSel>(grid, 256, lds, q, stream);
    }
    // test hook: did the last k_spec_check confirm the speculation? (-1: none ran on this context yet)
    long last_spec_verdict() {
        if (!small_.p) return -1;
        uint32_t v = 0;
        be.d2h(&v, d_spec_flag(), sizeof v, stream);
        be.sync(stream);
        return (long)v;
    }
    // speculations checked / confirmed on this context since it was created (bench.py reports the hit rate)
    void spec_totals(unsigned long long* checked, unsigned long long* hits) {
        uint32_t v[3] = {0, 0, 0};
        if (small_.p) { be.d2h(v, d_spec_flag(), sizeof v, stream); be.sync(stream); }
        *checked = v[1]; *hits = v[2];
    }
    // steps per thread of the cull's selection pass (and of the speculative blend in front of it)
    int cull_select_chunks(size_t total) const {
        int chunks = debug_sel_chunks ? (int)debug_sel_chunks : pick_chunks((total + 3) / 4, 256, 8, sel_wgs_per_cu);
        int grid2 = stream_grid((total + 3) / 4, 256, chunks);
        while ((size_t)(grid2 + CAND_GRID) * 4 > PART_DOUBLES) { chunks *= 2; grid2 = stream_grid((total + 3) / 4, 256, chunks); }
        return chunks;
    }

    // masked slerp sums + constants (reference functions.py:36-43 on the slerp class).
    // fused_parts > 0: the level-2 selection pass already left the sums in d_part().
    void run_slerp_consts(const Geo& g, bool have_thr, float t, int fused_parts, const float* ref_norms = nullptr,
                          const double* emf_part = nullptr, int emf_nparts = 0, int emf_elo = 0) {
        const size_t total = (size_t)g.Cb * g.R;
        SlerpConstParams c;
        c.ref_norms = ref_norms; c.emf_part = emf_part; c.emf_nparts = emf_nparts; c.emf_elo = emf_elo;
        c.fallback = nullptr; c.nfallback = 0; c.overflow = nullptr;
        c.thr = have_thr ? d_thr(0) : nullptr; c.t = t; c.out = d_consts();
        c.zero_u32 = nullptr; c.zero_u32_count = 0;
        if (fused_parts > 0) {
            c.zero_u32 = d_candctr(); c.zero_u32_count = 3;
            c.partials = d_part(); c.nparts = fused_parts;
        } else {
            ReduceParams r;
            r.reA = plane(g, P_REA); r.reB = plane(g, P_REB); r.R = g.R; r.C = g.Cw; r.Cb = g.Cb; r.vec4 = vec4(g);
            r.thr = c.thr; r.chunks = pick_chunks((total + 3) / 4, 256, 16, 8); r.only_if = nullptr;
            int grid = stream_grid((total + 3) / 4, 256, r.chunks);
            while ((size_t)grid * 4 > PART_DOUBLES) { r.chunks *= 2; grid = stream_grid((total + 3) / 4, 256, r.chunks); }
            r.partials = d_part();
            be.template launch<KReduce>(grid, 256, LDS_SCRATCH_FLOATS * 4, r, stream);
            c.partials = r.partials; c.nparts = grid;
        }
        be.template launch<KSlerpConsts>(1, 256, (2 * LDS_SCRATCH_FLOATS + 4 * EMF_VALS + 8) * 4, c, stream);
    }

    // norm_grid (optional): the row pass also leaves the sum of squares of what it stores in
    // d_part(), *norm_grid work-groups of it (read_norms() finishes the reduction)
    int run_inverse(const Geo& g, const float* reR, const float* imA, const float* cull_thr, const PairOut& o,
                    int* norm_grid = nullptr, float cull_val = 0.f) {
        I1Params a;
        int rc = get_plan(g.R, a.plan);
        if (rc) return rc;
        const int Cb = g.C / 2 + 1;
        a.cull_thr = cull_thr; a.cull_val = cull_val; a.R = g.R; a.Cb = Cb;
        a.G = (cf2*)t1_.p; a.pitchG = g.pitchG;
        a.s = g.R == 1 ? 1 : i1_bins_host(a.plan);
        const size_t lds1 = (LDS_SCRATCH_FLOATS + (size_t)a.s * a.plan.lds_floats) * 4;
        const int grid1 = (int)round_up((size_t)(Cb + a.s - 1) / a.s, 8 * (16 / a.s));

        I2Params b;
        bool blue; BluesteinTab bt;
        if ((rc = get_row_plan(g.C, b.plan, blue, bt))) return rc;
        b.G = (const cf2*)t1_.p; b.pitchG = g.pitchG; b.R = g.R; b.C = g.C; b.Cb = Cb;
        b.nb = blue ? 1 : std::max(1, 256 / b.plan.T);
        b.vec = !blue && (g.C % 8 == 0) && aligned16(o.out) && aligned16(o.base);
        b.inv_n = (float)(1.0 / ((double)g.R * (double)g.rough * (double)g.C));
        b.ifft_policy = o.ifft_policy;
        b.post = o.post; b.base_dtype = o.base_dtype; b.out_mode = o.out_mode;
        b.flags = d_flags();
        const bool joint = joint_rows(g);                 // the row blocks of a split column length: one row launch
        if (joint) b.R = g.R * g.rough;
        const int pairs = (b.R + 1) / 2;
        const int grid2 = (pairs + b.nb - 1) / b.nb;
        const int nrow_launch = joint ? 1 : g.batch;
        const bool want_norm = norm_grid && (size_t)grid2 * 2 * nrow_launch <= PART_DOUBLES;
        if (norm_grid) *norm_grid = want_norm ? grid2 * nrow_launch : -1;
        const size_t lds2 = (LDS_SCRATCH_FLOATS + (size_t)b.nb * b.plan.lds_floats) * 4;
        const size_t pslice = (size_t)Cb * g.R, eslice = (size_t)g.R * g.C;
        // a rank > 2 tensor goes slice by slice (columns, then rows; G is reused); the slices of a rough
        // column length all run their column pass first, k_dftp combines them, then the rows
        const int phases = g.rough > 1 ? 2 : 1;
        for (int phase = 0; phase < phases; ++phase) {
        if (phase == 1 && (rc = run_dftp(g, t1_.p, true, 1, g.pitchG, g.t1_slice / 2, true))) return rc;
        for (int bi = 0; bi < g.batch; ++bi) {
            if (g.rough > 1) { a.G = (cf2*)t1_.p + (size_t)bi * g.t1_slice; b.G = a.G; }
            a.reR = reR + bi * pslice; a.imA = imA + bi * pslice;
            if (phases == 2 && phase == 0 && g.R > 1) {      // the column pass of every row block in one launch
                if (bi > 0) break;
                a.G = (cf2*)t1_.p;
                a.sl.wgs = grid1; a.sl.t1_stride = g.t1_slice; a.sl.plane_stride = pslice;
                const int gridj = grid1 * g.batch;
                if (a.s >= 2) launch_fft<KI1x2>(a.plan, gridj, a.s * a.plan.T, lds1, a);
                else launch_fft<KI1x1>(a.plan, gridj, a.plan.T, lds1, a);
                continue;
            }
            if (phases == 2 && phase == 1) {
                // rows only
            } else
            if (g.R == 1) {
                be.template launch<KI1R1>(std::max(1, std::min(64, (Cb + 255) / 256)), 256, LDS_SCRATCH_FLOATS * 4, a, stream);
            } else if (g.fold == 4) {
                if (a.s >= 2) launch_fft<KI1x2Q>(a.plan, grid1, a.s * a.plan.T, lds1, a);
                else launch_fft<KI1x1Q>(a.plan, grid1, a.plan.T, lds1, a);
            } else if (a.s >= 2) launch_fft<KI1x2>(a.plan, grid1, a.s * a.plan.T, lds1, a);
            else launch_fft<KI1x1>(a.plan, grid1, a.plan.T, lds1, a);
            if (phases == 2 && phase == 0) continue;
            if (joint) { if (bi > 0) break; b.G = (const cf2*)t1_.p; }
            b.base = o.base ? (const char*)o.base + bi * eslice * dt_size(o.base_dtype) : nullptr;
            b.out = (char*)o.out + bi * eslice * (o.out_mode == OUT_BF16 ? 2 : 4);
            b.norm_partials = want_norm ? d_part() + (size_t)bi * 2 * grid2 : nullptr;
            if (blue) { I2BParams q; q.i = b; q.bt = bt; launch_fft<KI2B>(b.plan, grid2, b.plan.T, lds2, q); }
            else launch_fft<KI2>(b.plan, grid2, b.nb * b.plan.T, lds2, b, b.vec != 0);
        }
        }
        return SMHIP_OK;
    }

    // out = ca*a + cb*b (+ base, NaN/Inf policy, cast), optional sums of squares
    void run_combine(const SigDesc& a, const SigDesc& b, float ca, float cb, size_t n, float* out_f32,
                     const PairOut* fin, bool want_norms, int* grid_out = nullptr) {
        CombineParams c;
        c.a = a; c.b = b; c.ca = ca; c.cb = cb; c.n = n; c.out_f32 = out_f32;
        c.base = fin ? fin->base : nullptr; c.base_dtype = fin ? fin->base_dtype : DT_BF16;
        c.out_final = fin ? fin->out : nullptr; c.out_mode = fin ? fin->out_mode : OUT_F32;
        c.flags = d_flags();
        c.vec8 = (n % 8 == 0) && aligned16(a.x) && aligned16(a.base) && aligned16(b.x) && aligned16(b.base) &&
                 aligned16(out_f32) && aligned16(c.base) && aligned16(c.out_final);
        c.chunks = pick_chunks((n + 7) / 8, 256, 4, 8);
        int grid = stream_grid((n + 7) / 8, 256, c.chunks);
        while ((size_t)grid * 2 > PART_DOUBLES) { c.chunks *= 2; grid = stream_grid((n + 7) / 8, 256, c.chunks); }
        c.partials = want_norms ? d_part() : nullptr;
        be.template launch<KCombine>(grid, 256, LDS_SCRATCH_FLOATS * 4, c, stream);
        if (grid_out) *grid_out = grid;
    }

    // flags, thresholds and blend constants -> mailbox, one tiny kernel and one sync
    void publish(bool zero_flags) {
        PublishParams pp;
        pp.flags = d_flags(); pp.thr = d_thr(0); pp.consts = d_consts(); pp.mail = mail_;
        pp.zero_flags = zero_flags ? d_flags() : nullptr;
        be.template launch<KPublish>(1, 64, LDS_SCRATCH_FLOATS * 4, pp, stream);
        be.sync(stream);
        if (zero_flags) flags_clean_ = true;
    }
    void read_blend_info(smhip_blend_info* info, bool have_cut, bool have_cull, bool have_consts, bool published = false) {
        if (!info) return;
        if (!published) publish(false);
        const BlendConsts c = mail_->consts;
        float thr[4];
        for (int i = 0; i < 4; ++i) thr[i] = mail_->thr[i];
        info->cutoff_threshold = have_cut ? thr[0] : 0.0;
        info->cull_threshold = have_cull ? thr[1] : 0.0;
        info->dot = have_consts ? c.dot : 0.0;
        info->s00 = have_consts ? c.s00 : 0.0; info->s01 = have_consts ? c.s01 : 0.0; info->s11 = have_consts ? c.s11 : 0.0;
        info->n_slerp = have_consts ? c.n_slerp : 0;
        info->t = 0.0; info->cull_pct = 0.0;        // filled by callers that know them
    }

    static unsigned long long pct_index(unsigned long long len, double pct) {
        // Python: int(len(all_real) * pct), clamped to the last element (functions.py:115-119)
        unsigned long long idx = (unsigned long long)((double)len * pct);
        if (idx >= len) idx = len - 1;
        return idx;
    }

    // spectrum-domain part of a pair merge; planes P_REA/P_IMA/P_REB already hold
    // the (scaled) half spectra.  Leaves Re R in P_RER and the cull threshold in d_thr(1).
    void spectral_blend(const Geo& g, int mode, double t, double t_sum, double cutoff_pct, double cull_pct,
                        int agreement, bool level1_hist_done, bool& have_cull, int* sumsq_parts = nullptr, int spec_slot = 7) {
        const unsigned long long nfull = (unsigned long long)g.R * g.C * (unsigned long long)g.batch;
        have_cull = false;
        if (mode == BLEND_SLERP) {
            const bool have_cut = cutoff_pct > 0;
            int fused = 0;
            if (have_cut)
                run_select(g, plane(g, P_REA), plane(g, P_REB), pct_index(2 * nfull, cutoff_pct), level1_hist_done, d_thr(0),
                           true, &fused);
            // norm_mode = reference_cpu: the reference's cosine is taken with torch.norm's (biased) values of the gathered
            // class vectors (functions.py:36,40) - modelled from sampled statistics (k_class_emf); class_norms_mode 2
            // (test hook) runs the ordered emulation of sm_aten_norm.hpp over the planes instead
            const float* class_norms = nullptr;
            const double* emf_part = nullptr;
            int emf_nparts = 0, emf_elo = 0;
            if (ref_mode_ && class_norms_mode == 2 && !g.full) {
                AtenSrc cs[2];
                for (int w = 0; w < 2; ++w) {
                    memset(&cs[w], 0, sizeof(AtenSrc));
                    cs[w].kind = 1; cs[w].reA = plane(g, P_REA); cs[w].reB = plane(g, P_REB);
                    cs[w].thr = have_cut ? d_thr(0) : nullptr; cs[w].which = w;
                    cs[w].R = g.R; cs[w].C = g.Cw; cs[w].Cb = g.Cb; cs[w].n = (size_t)g.Cb * g.R;
                }
                if (run_aten_norms(cs, 2, false)) class_norms = d_aten_out();
            } else if (ref_mode_ && class_norms_mode == 1) {
                ClassEmfParams q;
                q.reA = plane(g, P_REA); q.reB = plane(g, P_REB); q.thr = have_cut ? d_thr(0) : nullptr;
                q.R = g.R; q.C = g.Cw; q.Cb = g.Cb; q.n = (size_t)g.Cb * g.R;
                // a lane ends near (class share ~ 1/4) * (sum of Re^2 over the full spectrum ~ n / 2) / 8: the
                // levels reach 4 binades above that and 11 below (further down the sum loses nothing)
                q.elo = (int)std::floor(std::log2(0.03 * (double)nfull + 1.0)) - 11;
                q.sample = 1;
                while (q.sample < emf_max_sample && q.n / (2 * (size_t)q.sample) >= EMF_MIN_SAMPLED) q.sample *= 2;
                const size_t rows = (q.n + 7) / 8, npieces = (rows + 8 * (size_t)q.sample - 1) / (8 * (size_t)q.sample);
                q.iters = (int)std::max<size_t>(1, std::min<size_t>(64, (npieces + (size_t)32 * 384 - 1) / ((size_t)32 * 384)));   // ~384 work-groups: their
                                                                         // 36-value partials are summed by ONE work-group (k_slerp_consts)
                const int grid = (int)std::max<size_t>(1, (npieces + (size_t)32 * q.iters - 1) / ((size_t)32 * q.iters));
                if (!ensure(emf_, (size_t)grid * 2 * EMF_VALS * sizeof(double))) {
                    q.partials = (double*)emf_.p;
                    be.template launch<KClassEmf>(grid, 256, LDS_SCRATCH_FLOATS * 4 * 2, q, stream);
                    emf_part = q.partials; emf_nparts = grid; emf_elo = q.elo;
                }
            }
            run_slerp_consts(g, have_cut, (float)t, fused, class_norms, emf_part, emf_nparts, emf_elo);
            const bool spec = cull_pct > 0 && spec_cull && !safe_select && !g.full && (size_t)g.Cb * g.R >= spec_min_bins;
            if (spec) run_blend_spec(g, (float)t_sum, sumsq_parts != nullptr, spec_slot);
            else run_blend(g, BLEND_SLERP, 1, (float)t, (float)t_sum, cull_pct > 0);
            if (sumsq_parts) *sumsq_parts = 0;
            if (cull_pct > 0) {
                run_select(g, plane(g, P_RER), nullptr, pct_index(nfull, cull_pct), true, d_thr(1), false, sumsq_parts,
                           sumsq_parts != nullptr, spec ? spec_slot : -1);
                have_cull = true;
            }
        } else {
            run_blend(g, BLEND_ARITH, agreement, (float)t, 1.f, false);
        }
    }

    // ---- a whole SLERP pair merge of 1-D tensors in one launch (k_pair1d) ------------------------------
    bool pair1d_enabled = true;
    bool pair1d_ok(const Geo& g) const {
        return pair1d_enabled && g.R == 1 && g.batch == 1 && !g.full && g.C >= 2 && g.C <= PAIR1D_MAX_C && !row_bluestein(g.C);
    }
    // A = the larger-norm input.  Leaves the thresholds in d_thr(0/1) and the constants in d_consts() as the
    // multi-kernel path does; norm_parts (optional): one [2] partial of the stored values' squares in d_part()
    int run_pair1d(const Geo& g, const SigDesc& A, const SigDesc& Bs, double na, double nb, double t, double t_sum,
                   double cutoff_pct, double cull_pct, const PairOut& o, int* norm_parts) {
        Pair1dParams q;
        int rc = get_plan(g.C, q.plan);
        if (rc) return rc;
        q.a = A; q.b = Bs; q.C = g.C;
        q.sa = (float)(1.0 / na); q.sb = (float)(1.0 / nb);
        q.t = (float)t; q.t_sum = (float)t_sum;
        const unsigned long long nfull = (unsigned long long)g.C;
        q.have_cut = cutoff_pct > 0; q.rank_cut = pct_index(2 * nfull, cutoff_pct);
        q.have_cull = cull_pct > 0; q.rank_cull = pct_index(nfull, cull_pct);
        memset(&q.fin, 0, sizeof q.fin);
        q.fin.R = 1; q.fin.C = g.C; q.fin.Cb = g.C / 2 + 1;
        q.fin.inv_n = (float)(1.0 / (double)g.C);
        q.fin.ifft_policy = o.ifft_policy; q.fin.post = o.post;
        q.fin.base = o.base; q.fin.base_dtype = o.base_dtype; q.fin.out = o.out; q.fin.out_mode = o.out_mode;
        q.fin.flags = d_flags();
        q.fin.norm_partials = norm_parts ? d_part() : nullptr;
        if (norm_parts) *norm_parts = 1;
        q.thr = d_thr(0); q.consts = d_consts();
        const size_t lds = (LDS_SCRATCH_FLOATS + (size_t)q.plan.lds_floats + 2 * (size_t)g.C + HIST1_BINS + 1024 + 64 + 16) * 4;
        launch_fft<KPair1d>(q.plan, 1, q.plan.T, lds, q);
        return SMHIP_OK;
    }

    // ---- A9: merge_tensors_fft2_slerp on fp32 inputs ---------------------------------
    int merge_pair_slerp(const float* v0, const float* v1, int R, int C, double t, double bthr, double t_sum,
                         double cutoff_pct, double cull_pct, float* out, double* n0o, double* n1o, int* branch,
                         smhip_blend_info* info) {
        return with_select_retry([&] {
            return merge_pair_slerp_once(v0, v1, R, C, t, bthr, t_sum, cutoff_pct, cull_pct, out, n0o, n1o, branch, info);
        });
    }
    int merge_pair_slerp_once(const float* v0, const float* v1, int R, int C, double t, double bthr, double t_sum,
                              double cutoff_pct, double cull_pct, float* out, double* n0o, double* n1o, int* branch,
                              smhip_blend_info* info) {
        const Geo g = geo(R, C, false, aligned16(v0) && aligned16(v1) && aligned16(out));
        int rc = reserve(R, C);
        if (rc) return rc;
        clear_flags();
        SigDesc a{v0, nullptr, DT_F32, 1.f}, b{v1, nullptr, DT_F32, 1.f};
        int grid;
        if ((rc = run_f1(g, a, b, grid))) return rc;
        double n0, n1;
        read_norms(grid, n0, n1);
        // reference normalises in fp32: norm().item() is an fp32 value
        n0 = (double)(float)n0; n1 = (double)(float)n1;
        if (n0o) *n0o = n0;
        if (n1o) *n1o = n1;
        const size_t n = (size_t)R * C;
        if (n1 < 1e-4 || n0 < 1e-4) {              // functions.py:184-190: normalised v0 comes back
            if (branch) *branch = SMHIP_BRANCH_EARLY_V0;
            SigDesc none{nullptr, nullptr, DT_F32, 1.f};
            run_combine(a, none, n0 != 0 ? (float)(1.0 / n0) : 1.f, 0.f, n, out, nullptr, false);
            read_blend_info(info, false, false, false);
            return SMHIP_OK;
        }
        const double ratio = n1 / (n0 + 1e-10);
        PairOut po;
        po.out = out; po.out_mode = OUT_F32; po.post = 1.f;
        if (ratio < bthr) {
            // functions.py:199-202: R = F0 + t*F1, Im included: linear in the spectrum.
            if (branch) *branch = SMHIP_BRANCH_LINEAR;
            if ((rc = run_f2_linear(g, (float)(1.0 / n0), (float)(1.0 / n1), (float)t))) return rc;
            if ((rc = run_inverse(g, plane(g, P_RER), plane(g, P_IMA), nullptr, po))) return rc;
            read_blend_info(info, false, false, false);
            return check_flags(true, false);
        }
        if (branch) *branch = SMHIP_BRANCH_SLERP;
        if ((rc = run_f2(g, (float)(1.0 / n0), (float)(1.0 / n1), 0, cutoff_pct > 0))) return rc;
        bool have_cull;
        spectral_blend(g, BLEND_SLERP, t, t_sum, cutoff_pct, cull_pct, 1, true, have_cull);
        if ((rc = run_inverse(g, plane(g, P_RER), plane(g, P_IMA), have_cull ? d_thr(1) : nullptr, po))) return rc;
        read_blend_info(info, cutoff_pct > 0, have_cull, true);
        if (info) { info->t = t; info->cull_pct = cull_pct; }
        return check_flags(true, false);
    }

    // ratio < b path: spectra added with weight t (needs Im of both inputs): do it in
    // the spectral domain exactly as the reference: Re R = ra + t rb, Im R = ia + t ib.
    int run_f2_linear(const Geo& g, float s0, float s1, float t) {
        // Im b is not kept by F2, so run F2 twice with roles swapped: first pass
        // leaves (Re a, Im a, Re b); second pass with swap=1 leaves Im b in P_IMA.
        int rc = run_f2(g, s0, s1, 0, false);
        if (rc) return rc;
        const size_t total = (size_t)g.Cb * g.R;
        // Re R = ra + t*rb  (arithmetic blend without agreement)
        run_blend(g, BLEND_ARITH, 0, t, 1.f, false);
        // keep Im a in tmp, fetch Im b, then Im R = ia + t*ib
        if ((rc = ensure(tmpC_, g.plane_floats * sizeof(float)))) return rc;
        SigDesc ia{plane(g, P_IMA), nullptr, DT_F32, 1.f}, none{nullptr, nullptr, DT_F32, 1.f};
        run_combine(ia, none, 1.f, 0.f, total, (float*)tmpC_.p, nullptr, false);
        // second F2 with swapped roles overwrites P_REA/P_IMA with b's planes and P_REB with a's
        if ((rc = run_f2(g, s0, s1, 1, false))) return rc;
        SigDesc ib{plane(g, P_IMA), nullptr, DT_F32, 1.f}, ia2{tmpC_.p, nullptr, DT_F32, 1.f};
        run_combine(ia2, ib, 1.f, t, total, plane(g, P_IMA), nullptr, false);
        return SMHIP_OK;
    }

    // NaN/Inf flags and, right behind them, the candidate-list counters with their sticky overflow word
    // (every check_flags() leaves the device flags cleared; a memset is needed only after a
    // call that ended without one)
    void clear_flags() {
        if (!small_.p && reserve(1, 1)) return;      // first call on this context: the small block and the mailbox
        if (!flags_clean_) be.memset(d_flags(), 0, 48, stream);
        flags_clean_ = false;
        overflow_seen_ = false;
    }
    // did a candidate list overflow since clear_flags()?  (sticky word, read with a sync)
    bool select_overflowed() {
        uint32_t v = 0;
        publish(false);
        v = mail_->flags[11];
        return v != 0;
    }
    // Run `body` (a whole API call); if one of its selections overflowed its candidate
    // lists the results are void: redo the call once with full-pass selection.
    template <class F>
    int with_select_retry(F&& body) {
        int rc = body();
        if (!safe_select && overflow_seen_) {
            safe_select = true;
            rc = body();
            safe_select = false;
        }
        return rc;
    }

    int check_flags(bool ifft_stage, bool final_stage, uint32_t* nan_ifft = nullptr, uint32_t* nan_final = nullptr) {
        uint32_t f[12];                      // flags[8] + candidate counters[4]
        publish(true);
        for (int i = 0; i < 12; ++i) f[i] = mail_->flags[i];
        if (f[11] && !safe_select) { overflow_seen_ = true; return SMHIP_OK; }   // with_select_retry() redoes the call
        if (nan_ifft) *nan_ifft = f[0];
        if (nan_final) *nan_final = f[2];
        if (ifft_stage && f[1]) return fail(SMHIP_ERR_INF_IFFT, "Inf in ifft output");
        if (final_stage && f[3]) return fail(SMHIP_ERR_INF_MERGED, "Inf in merged tensor");
        return SMHIP_OK;
    }

    // ---- A10: task_arithmetic_fft2 on fp32 inputs ------------------------------------
    int pair_arith(const SigDesc& a_in, const SigDesc& b_in, int R, int C, float sa, float sb, double t, int agreement,
                   const PairOut& po, double na_hint, double nb_hint, const Geo* layer_geo = nullptr) {
        // (layer_geo: the layer's own geometry - slices of a rank > 2 tensor or of a rough column length)
        const Geo g = layer_geo ? *layer_geo
                                : geo(R, C, false, aligned16(a_in.x) && aligned16(a_in.base) && aligned16(b_in.x) && aligned16(b_in.base) &&
                                                   aligned16(po.out) && aligned16(po.base));
        int rc = reserve(g.R, g.C, false, g.batch);
        if (rc) return rc;
        SigDesc a = a_in, b = b_in;
        int grid;
        // two-for-one packs a and b into one complex transform: rounding noise of the
        // larger contaminates the smaller at ~1e-7 of the larger, so lift a tiny b by
        // an exact power of two first (its sign pattern decides the blend, quirk Q3).
        double na = na_hint, nb = nb_hint;
        if (na < 0 || nb < 0) {
            if ((rc = run_f1(g, a, b, grid))) return rc;
            read_norms(grid, na, nb);
            na *= std::fabs(sa); nb *= std::fabs(sb);
        } else {
            na *= std::fabs(sa); nb *= std::fabs(sb);
            grid = -1;
        }
        float pre = 1.f;
        if (nb == 0) { b.x = nullptr; b.base = nullptr; }     // exactly zero: F1 writes exact zeros
        if (nb > 0 && na > 0 && (nb < na * 0.25 || nb > na * 4.0)) {
            int e;
            std::frexp(na / nb, &e);
            pre = std::ldexp(1.f, e - 1);
        }
        if (pre != 1.f || grid < 0 || nb == 0) {
            b.prescale = b_in.prescale * pre;
            if ((rc = run_f1(g, a, b, grid))) return rc;
        }
        if ((rc = run_f2(g, sa, sb / pre, 0, false))) return rc;
        bool have_cull;
        spectral_blend(g, BLEND_ARITH, t, 1.0, 0, 0, agreement, false, have_cull);
        return run_inverse(g, plane(g, P_RER), plane(g, P_IMA), nullptr, po);
    }

    // ---- A1-A13: the layer tournament ---------------------------------------------------
    // greedy pairing, reference functions.py:316-365 with way="least"
    static void correlated_pairs_least(const std::vector<float>& norms, int m, std::vector<std::pair<int, int>>& out) {
        out.clear();
        std::vector<char> used(m, 0);
        std::vector<float> corr((size_t)m * m, 0.f);
        for (int i = 0; i < m; ++i)
            for (int j = i + 1; j < m; ++j) corr[(size_t)i * m + j] = norms[i] * norms[j];
        for (;;) {
            float best = INFINITY;
            bool any = false;
            for (int i = 0; i < m; ++i)
                for (int j = i + 1; j < m; ++j)
                    if (!used[i] && !used[j]) { any = true; best = std::min(best, std::fabs(corr[(size_t)i * m + j])); }
            if (!any) break;
            int bx = -1, by = -1;
            for (int i = 0; i < m && bx < 0; ++i)
                for (int j = i + 1; j < m; ++j)
                    if (!used[i] && !used[j] && std::fabs(corr[(size_t)i * m + j]) == best) { bx = i; by = j; break; }
            if (bx < 0) break;      // NaN norms: nothing compares equal (reference breaks too)
            out.emplace_back(bx, by);
            used[bx] = used[by] = 1;
        }
        for (int i = 0; i < m; ++i)
            if (!used[i]) out.emplace_back(i, -1);
    }

    struct Slot {           // one entry of the reference's layer_stack
        SigDesc sig;        // where its values live (spatial)
        double weight;
        double norm;        // ||.||_2 if known, else < 0
        // spectral intermediate (result of an earlier SLERP pair merge, K >= 3): the pair's Re R and
        // Im a planes, detached from the working set; the cull is still to be applied (thr)
        bool spectral = false;
        int re_id = -1, im_id = -1;     // pool buffers
        float thr = 0.f;                // cull threshold (0: none)
        double spec_scale = 1.0;        // 1 / sqrt(sum |R_culled|^2 / n): brings the spectrum to unit spatial norm
        double post = 1.0;              // spatial values = post * ifft(R)   (target_norm)
        int rows_id = -1;               // >= 0: rowspec_[rows_id] holds this raw delta's row spectra (row-pair layout)
    };
    // spectral -> spatial fp32 (the branches that need spatial inputs: add, Arithmetic-FFT, early-out)
    int materialise(const Geo& g, Slot& s, std::vector<char>& inter_busy) {
        if (!s.spectral) return SMHIP_OK;
        const size_t n = (size_t)g.R * g.C;
        int id = -1;
        for (size_t q = 0; q < inter_.size(); ++q) if (!inter_busy[q]) { id = (int)q; break; }
        if (id < 0) { inter_.emplace_back(); inter_busy.push_back(0); id = (int)inter_.size() - 1; }
        int rc = ensure(inter_[id], n * sizeof(float));
        if (rc) return rc;
        inter_busy[id] = 1;
        PairOut po;
        po.out = inter_[id].p; po.out_mode = OUT_F32; po.post = (float)s.post;
        if ((rc = run_inverse(g, (const float*)pool_[s.re_id].p, (const float*)pool_[s.im_id].p, nullptr, po, nullptr, s.thr))) return rc;
        pool_release(s.re_id); pool_release(s.im_id);
        s.spectral = false; s.re_id = s.im_id = -1;
        s.sig = SigDesc{inter_[id].p, nullptr, DT_F32, 1.f};
        return SMHIP_OK;
    }

    // all delta norms in one pass (16-bit inputs, at most two distinct bases, aligned, n % 8 == 0);
    // false: not applicable, the caller takes the pairwise path
    bool run_delta_norms(const smhip_layer_desc& d, size_t n, std::vector<Slot>& stack) {
        if (d.k > NORMS_K || d.in_dtype == DT_F32 || (n % 8) != 0) return false;
        DeltaNormsParams p;
        p.k = d.k; p.dtype = d.in_dtype; p.n = n;
        p.ubase[0] = p.ubase[1] = nullptr;
        int nb = 0;
        for (int i = 0; i < NORMS_K; ++i) { p.ft[i] = d.finetune[i < d.k ? i : 0]; p.base_of[i] = 0; }
        for (int i = 0; i < d.k; ++i) {
            if (!d.base[i] || !aligned16(d.base[i]) || !aligned16(d.finetune[i])) return false;
            int j = 0;
            while (j < nb && p.ubase[j] != d.base[i]) ++j;
            if (j == nb) { if (nb == 2) return false; p.ubase[nb++] = d.base[i]; }
            p.base_of[i] = j;
        }
        p.chunks = pick_chunks(n / 8, 256, 4, 8);
        int grid = stream_grid(n / 8, 256, p.chunks);
        while ((size_t)grid * NORMS_MAX > PART_DOUBLES) { p.chunks *= 2; grid = stream_grid(n / 8, 256, p.chunks); }
        p.partials = d_part();
        be.template launch<KDeltaNorms>(grid, 256, LDS_SCRATCH_FLOATS * 4, p, stream);
        SumNParams sp;
        sp.partials = d_part(); sp.nparts = grid; sp.out = mail_->norm2;
        be.template launch<KSumPartialsN>(1, 256, LDS_SCRATCH_FLOATS * 4, sp, stream);
        be.sync(stream);
        for (int i = 0; i < d.k; ++i) stack[i].norm = std::sqrt(mail_->norm2[i]);
        return true;
    }

    // norm_mode = reference_cpu: norms as torch's CPU kernel returns them (sm_aten_norm.hpp), for up to
    // ATEN_MAX_SIGS row streams at once.  Results land in d_aten_out()[0..nsig) on the device and, with
    // to_mail, in mail_->snorm after the caller's next sync.  Needs 16-byte aligned inputs for the
    // vector loads (false: not applicable, the caller keeps what it has).
    float* d_aten_out() { return (float*)aten_.p; }
    float* d_aten_lanes() { return (float*)aten_.p + ATEN_MAX_SIGS; }
    uint32_t* d_aten_stats() { return (uint32_t*)((float*)aten_.p + ATEN_MAX_SIGS + ATEN_MAX_SIGS * 8); }
    static constexpr size_t ATEN_HEAD_BYTES = 4096;
    bool aten_serial = false;          // test hook: the old single-work-group serial chain (k_serial_norm) instead
    int class_norms_mode = 1;          // reference_cpu and the slerp class's norms: 1 = modelled from sampled statistics
                                       // (k_class_emf), 2 = ordered emulation over the planes, 0 = left exact (test hooks)
    bool ref_mode_ = false;            // the current layer runs with norm_mode = reference_cpu
    // walk_on: a side stream for the walker and the finish (forked off the caller's stream behind the summaries;
    // the CALLER joins it): those two are chains of dependent steps on 8 work-groups per signal - latency, no load
    bool run_aten_norms(const AtenSrc* srcs, int nsig, bool to_mail, void* walk_on = nullptr) {
        if (nsig < 1 || nsig > ATEN_MAX_SIGS) return false;
        size_t max_rows = 0;
        const int kind = srcs[0].kind;
        AtenSrc local[ATEN_MAX_SIGS];
        for (int i = 0; i < nsig; ++i) {
            local[i] = srcs[i];
            AtenSrc& a = local[i];
            if (a.kind != kind) return false;
            // (an unaligned signal is loaded element by element: the mode must never fall back to other numerics silently)
            if (a.kind == 0) a.unaligned = (!aligned16(a.sig.x) || !aligned16(a.sig.base)) ? 1 : 0;
            else if (!aligned16(a.reA) || !aligned16(a.reB)) return false;      // planes of the workspace: always aligned
            max_rows = std::max(max_rows, aten_rows(a));
        }
        srcs = local;
        const size_t nchunks = std::max<size_t>(1, (max_rows + ATEN_CHUNK_ROWS - 1) / ATEN_CHUNK_ROWS);
        if (nchunks * nsig > (size_t)1 << 30) return false;
        const size_t pre_bytes = round_up((size_t)nsig * nchunks * 8 * sizeof(double), 256);
        const size_t rec_bytes = round_up((size_t)nsig * nchunks * 16 * sizeof(AtenSum), 256);
        const size_t grp_bytes = round_up((size_t)nsig * nchunks * 16 * ATEN_GROUPS * sizeof(AtenSum), 256);
        const size_t ep_bytes = round_up((size_t)nsig * nchunks * 8 * sizeof(int), 256);
        if (ensure(aten_, ATEN_HEAD_BYTES + pre_bytes + rec_bytes + grp_bytes + ep_bytes)) return false;
        char* basep = (char*)aten_.p + ATEN_HEAD_BYTES;
        double* pre = (double*)basep;
        AtenSum* rec = (AtenSum*)(basep + pre_bytes);
        AtenSum* grp = (AtenSum*)(basep + pre_bytes + rec_bytes);
        int* epred = (int*)(basep + pre_bytes + rec_bytes + grp_bytes);
        const bool summaries = nchunks > 1;
        if (summaries) {
            AtenPreParams a;
            a.nsig = nsig; a.nchunks = nchunks; a.pre = pre;
            for (int i = 0; i < ATEN_MAX_SIGS; ++i) a.src[i] = srcs[i < nsig ? i : 0];
            if (kind) be.template launch<KAtenPreC>((int)(nchunks * nsig), ATEN_THREADS, LDS_SCRATCH_FLOATS * 4, a, stream);
            else be.template launch<KAtenPre>((int)(nchunks * nsig), ATEN_THREADS, LDS_SCRATCH_FLOATS * 4, a, stream);
            AtenScanParams sc;
            sc.pre = pre; sc.nchunks = nchunks;
            be.template launch<KAtenScan>(nsig, ATEN_THREADS, (LDS_SCRATCH_FLOATS + 2 * 32 * 8) * 4, sc, stream);
            AtenPartParams b;
            b.nsig = nsig; b.nchunks = nchunks; b.prefix = pre; b.rec = rec; b.grp = grp; b.epred = epred;
            for (int i = 0; i < ATEN_MAX_SIGS; ++i) b.src[i] = srcs[i < nsig ? i : 0];
            int mode = aten_part_mode(srcs[0]);
            for (int i = 1; i < nsig; ++i) if (aten_part_mode(srcs[i]) != mode || srcs[i].sig.dtype != srcs[0].sig.dtype) mode = kind ? ATEN_PART_CLASS : ATEN_PART_SIGNAL;
            const int pgrid = (int)round_up(nchunks * nsig, (size_t)8 * nsig);
            const size_t plds = (LDS_SCRATCH_FLOATS + ATEN_PART_LDS_FLOATS) * 4;
            if (mode == ATEN_PART_RAW16) be.template launch<KAtenPart16>(pgrid, ATEN_THREADS, plds, b, stream);
            else if (mode == ATEN_PART_RAW32) be.template launch<KAtenPart32>(pgrid, ATEN_THREADS, plds, b, stream);
            else if (mode == ATEN_PART_CLASS) be.template launch<KAtenPartC>(pgrid, ATEN_THREADS, plds, b, stream);
            else be.template launch<KAtenPart>(pgrid, ATEN_THREADS, plds, b, stream);
        }
        void* const main_stream = stream;
        struct StreamSwap { void*& s; void* keep; ~StreamSwap() { s = keep; } } swap_back{stream, main_stream};
        if (walk_on) { be.fork(main_stream, walk_on); stream = walk_on; }
        AtenWalkParams w;
        w.nsig = nsig; w.nchunks = nchunks; w.rec = summaries ? rec : nullptr; w.grp = grp; w.epred = epred;
        w.lanes = d_aten_lanes(); w.stats = d_aten_stats();
        for (int i = 0; i < ATEN_MAX_SIGS; ++i) w.src[i] = srcs[i < nsig ? i : 0];
        if (kind) be.template launch<KAtenWalkC>(nsig * 8, ATEN_THREADS, (LDS_SCRATCH_FLOATS + ATEN_WALK_LDS_FLOATS) * 4, w, stream);
        else be.template launch<KAtenWalk>(nsig * 8, ATEN_THREADS, (LDS_SCRATCH_FLOATS + ATEN_WALK_LDS_FLOATS) * 4, w, stream);
        AtenFinishParams f;
        f.nsig = nsig; f.lanes = d_aten_lanes(); f.out = d_aten_out(); f.mail = to_mail ? mail_->snorm : nullptr;
        for (int i = 0; i < ATEN_MAX_SIGS; ++i) f.src[i] = srcs[i < nsig ? i : 0];
        be.template launch<KAtenFinish>(1, 64, LDS_SCRATCH_FLOATS * 4, f, stream);
        return true;
    }
    // ---- reference_cpu, fused: the forward row pass summarises the deltas it forms (AtenFuse, sm_aten_core.hpp) ------
    // begin_fused_norms(): sampled prefix + scan (the binade predictions), buffers; the row pass launched next writes
    // the group summaries; finish_fused_norms(): chunk summaries, walker, finish -> mail_->snorm after the next sync.
    bool fused_done_ = false;
    struct FusedNorms {
        bool armed = false, used = false;
        AtenFuse af;
        int nsig = 0, next_sig = 0;
        size_t nchunks = 0, groups = 0;
        AtenSum* rec = nullptr; int* epred = nullptr;
        AtenSrc srcs[ATEN_MAX_SIGS];
    } fused_;
    bool fuse_norms = true;            // test hook "fuse_norms" = 0: the separate summary pass (k_aten_part) as before
    // the row plan of this geometry summarises in its natural scatter: a static plan, whole groups of 256 rows-of-8
    // per matrix row, one wave per group
    bool fusable_rows(const Geo& g) {
        FftPlanDev pl;
        // (not the folded row pass: k_f1q's transform phase is fully exposed and the summaries cost it more than the
        //  separate pass they replace - see k_f1q)
        if (g.batch != 1 || g.rough != 1 || g.full || g.fold != 1 || row_bluestein(g.C) || get_plan(g.C, pl) || !is_static_plan(pl)) return false;
        return aten_fusable(g.C, pl.T);
    }
    bool begin_fused_norms(const Geo& g, const std::vector<Slot>& stack, size_t n) {
        fused_.armed = fused_.used = false;
        const int k = (int)stack.size();
        if (!fuse_norms || aten_serial || k < 1 || k > ATEN_MAX_SIGS || !fusable_rows(g) || (n % (8 * ATEN_GROUP_ROWS)) != 0) return false;
        const size_t rows = n / 8;
        const size_t nchunks = (rows + ATEN_CHUNK_ROWS - 1) / ATEN_CHUNK_ROWS;
        if (nchunks < 2 || nchunks * k > (size_t)1 << 30) return false;       // (one chunk: the walker works from the data)
        for (int i = 0; i < k; ++i) {
            const SigDesc& sg = stack[i].sig;
            if (!sg.x || !sg.base || sg.dtype == DT_F32 || sg.prescale != 1.f || sg.dtype != stack[0].sig.dtype ||
                !aligned16(sg.x) || !aligned16(sg.base)) return false;
            memset(&fused_.srcs[i], 0, sizeof(AtenSrc));
            fused_.srcs[i].kind = 0; fused_.srcs[i].sig = sg; fused_.srcs[i].n = n; fused_.srcs[i].C = -1;
        }
        const size_t pre_bytes = round_up((size_t)k * nchunks * 8 * sizeof(double), 256);
        const size_t rec_bytes = round_up((size_t)k * nchunks * 16 * sizeof(AtenSum), 256);
        const size_t grp_bytes = round_up((size_t)k * nchunks * 16 * ATEN_GROUPS * sizeof(AtenSum), 256);
        const size_t ep_bytes = round_up((size_t)k * nchunks * 8 * sizeof(int), 256);
        if (ensure(aten_, ATEN_HEAD_BYTES + pre_bytes + rec_bytes + grp_bytes + ep_bytes)) return false;
        char* basep = (char*)aten_.p + ATEN_HEAD_BYTES;
        double* pre = (double*)basep;
        AtenPreParams a;
        a.nsig = k; a.nchunks = nchunks; a.pre = pre;
        for (int i = 0; i < ATEN_MAX_SIGS; ++i) a.src[i] = fused_.srcs[i < k ? i : 0];
        be.template launch<KAtenPre>((int)(nchunks * k), ATEN_THREADS, LDS_SCRATCH_FLOATS * 4, a, stream);
        AtenScanParams sc;
        sc.pre = pre; sc.nchunks = nchunks;
        be.template launch<KAtenScan>(k, ATEN_THREADS, (LDS_SCRATCH_FLOATS + 2 * 32 * 8) * 4, sc, stream);
        fused_.af.prefix = pre; fused_.af.grp = (AtenSum*)(basep + pre_bytes + rec_bytes); fused_.af.nchunks = nchunks; fused_.af.sig0 = 0;
        fused_.rec = (AtenSum*)(basep + pre_bytes); fused_.epred = (int*)(basep + pre_bytes + rec_bytes + grp_bytes);
        fused_.nsig = k; fused_.nchunks = nchunks; fused_.groups = rows / ATEN_GROUP_ROWS; fused_.next_sig = 0;
        fused_.armed = true;
        return true;
    }
    // behind the row pass(es): false when no row pass took the summaries up (the caller falls back to the separate pass)
    bool finish_fused_norms() {
        const bool ok = fused_.armed && fused_.used;
        fused_.armed = false;
        if (!ok) return false;
        const int k = fused_.nsig;
        AtenRecParams r;
        r.nsig = k; r.nchunks = fused_.nchunks; r.prefix = fused_.af.prefix; r.grp = fused_.af.grp; r.rec = fused_.rec;
        r.epred = fused_.epred; r.groups = fused_.groups;
        const size_t total = (size_t)k * fused_.nchunks * 16;
        be.template launch<KAtenRec>((int)((total + ATEN_THREADS - 1) / ATEN_THREADS), ATEN_THREADS, LDS_SCRATCH_FLOATS * 4, r, stream);
        AtenWalkParams w;
        w.nsig = k; w.nchunks = fused_.nchunks; w.rec = fused_.rec; w.grp = fused_.af.grp; w.epred = fused_.epred;
        w.lanes = d_aten_lanes(); w.stats = d_aten_stats();
        for (int i = 0; i < ATEN_MAX_SIGS; ++i) w.src[i] = fused_.srcs[i < k ? i : 0];
        be.template launch<KAtenWalk>(k * 8, ATEN_THREADS, (LDS_SCRATCH_FLOATS + ATEN_WALK_LDS_FLOATS) * 4, w, stream);
        AtenFinishParams f;
        f.nsig = k; f.lanes = d_aten_lanes(); f.out = d_aten_out(); f.mail = mail_->snorm;
        for (int i = 0; i < ATEN_MAX_SIGS; ++i) f.src[i] = fused_.srcs[i < k ? i : 0];
        be.template launch<KAtenFinish>(1, 64, LDS_SCRATCH_FLOATS * 4, f, stream);
        return true;
    }

    // the delta / fp32 norms of `k` signals of n elements each -> out[] (one sync)
    bool run_serial_norms(const SigDesc* sigs, int k, size_t n, double* out) {
        if (k < 1 || k > 16) return false;
        if (aten_serial) {
            if ((n % 8) != 0) return false;
            SerialNormParams q;
            q.k = k; q.n = n; q.out = mail_->snorm;
            for (int i = 0; i < 16; ++i) {
                q.sig[i] = sigs[i < k ? i : 0];
                if (!aligned16(q.sig[i].x) || !aligned16(q.sig[i].base)) return false;
            }
            be.template launch<KSerialNorm>(k, 256, (LDS_SCRATCH_FLOATS + 2 * SER_CHUNK) * 4, q, stream);
        } else {
            AtenSrc srcs[ATEN_MAX_SIGS];
            for (int i = 0; i < k; ++i) {
                memset(&srcs[i], 0, sizeof(AtenSrc));
                srcs[i].kind = 0; srcs[i].sig = sigs[i]; srcs[i].n = n; srcs[i].C = -1;
            }
            if (!run_aten_norms(srcs, k, true)) return false;
        }
        be.sync(stream);
        for (int i = 0; i < k; ++i) out[i] = (double)mail_->snorm[i];
        return true;
    }
    // norm_mode = reference_cpu, the deltas' norms: the summaries are computed on the caller's stream, the walker (8
    // work-groups per signal, ~200 us of dependent steps) runs on a side stream of the backend beside the row pass
    // and is joined before anything reads its result.  false: not started (the caller takes the synchronous path)
    void* aten_walk_pending_ = nullptr;
    bool aten_overlap = true;
    int emf_max_sample = EMF_MAX_SAMPLE;   // k_class_emf reads at most one piece of 8 rows in this many (debug option "emf_max_sample")
    bool begin_delta_ref_norms(const std::vector<Slot>& stack, size_t n) {
        const int k = (int)stack.size();
        void* aux = aten_overlap && !aten_serial ? be.aux_stream() : nullptr;
        if (!aux || k < 1 || k > ATEN_MAX_SIGS) return false;
        AtenSrc srcs[ATEN_MAX_SIGS];
        for (int i = 0; i < k; ++i) {
            memset(&srcs[i], 0, sizeof(AtenSrc));
            srcs[i].kind = 0; srcs[i].sig = stack[i].sig; srcs[i].n = n; srcs[i].C = -1;
        }
        aten_walk_pending_ = run_aten_norms(srcs, k, true, aux) ? aux : nullptr;
        return aten_walk_pending_ != nullptr;
    }
    void end_delta_ref_norms() {
        if (aten_walk_pending_) be.join(aten_walk_pending_, stream);
        aten_walk_pending_ = nullptr;
    }

    // test hook: walker statistics of the last run_aten_norms, summed over lanes: chunks composed from their
    // summaries / crossed with the help of the group summaries / walked cooperatively
    void aten_stats(int nsig, unsigned long long* out3) {
        std::vector<uint32_t> h((size_t)nsig * 32);
        be.d2h(h.data(), d_aten_stats(), h.size() * 4, stream);
        out3[0] = out3[1] = out3[2] = 0;
        for (int i = 0; i < nsig * 8; ++i) for (int q = 0; q < 3; ++q) out3[q] += h[4 * i + q];
    }

    // row spectra of every raw delta (row-pair F1 each) + their norms with ONE sync
    int rows_first(const Geo& g, std::vector<Slot>& stack) {
        const int k = (int)stack.size();
        if ((int)rowspec_.size() < k) rowspec_.resize(k);
        const size_t bytes = (g.t1_slice / 2) * g.batch * sizeof(cf4) + 4096;
        int rc;
        std::vector<int> grids(k);
        size_t poff = 0;
        for (int i = 0; i < k; ++i) if ((rc = ensure(rowspec_[i], bytes))) return rc;
        // one launch for all of them when they are alike (same dtype, aligned): the signals of a row block on one XCD
        bool alike = f1_multi && k >= 2 && k <= F1_MAX_SIGS;
        for (int i = 0; i < k && alike; ++i)
            alike = stack[i].sig.dtype == stack[0].sig.dtype && stack[i].sig.prescale == 1.f && stack[i].sig.x &&
                    (stack[i].sig.base != nullptr) == (stack[0].sig.base != nullptr) &&
                    aligned16(stack[i].sig.x) && aligned16(stack[i].sig.base);
        if (alike) {
            F1Sigs ms;
            memset(&ms, 0, sizeof ms);
            ms.n = k;
            for (int i = 0; i < k; ++i) { ms.x[i] = stack[i].sig.x; ms.base[i] = stack[i].sig.base; ms.t1[i] = (cf4*)rowspec_[i].p; }
            int gq = 0;
            rc = run_f1_rowpairs(g, stack[0].sig, rowspec_[0].p, d_part(), &gq, &ms);
            if (rc == SMHIP_OK) {
                for (int i = 0; i < k; ++i) {
                    SumPartialsParams sp;
                    sp.partials = d_part() + (size_t)i * 2 * gq; sp.nparts = gq; sp.out = mail_->norm2 + 2 * i;
                    be.template launch<KSumPartials>(1, 1024, LDS_SCRATCH_FLOATS * 4, sp, stream);
                }
                fused_done_ = finish_fused_norms();
                be.sync(stream);
                for (int i = 0; i < k; ++i) {
                    stack[i].norm = std::sqrt(mail_->norm2[2 * i] + mail_->norm2[2 * i + 1]);
                    stack[i].rows_id = i;
                }
                return SMHIP_OK;
            }
            if (rc != SMHIP_ERR_ARG) return rc;
        }
        for (int i = 0; i < k; ++i) {
            fused_.next_sig = i;
            if ((rc = run_f1_rowpairs(g, stack[i].sig, rowspec_[i].p, d_part() + poff, &grids[i]))) return rc;
            SumPartialsParams sp;
            sp.partials = d_part() + poff; sp.nparts = grids[i]; sp.out = mail_->norm2 + 2 * i;
            be.template launch<KSumPartials>(1, 1024, LDS_SCRATCH_FLOATS * 4, sp, stream);
            poff += 2 * (size_t)grids[i];
        }
        fused_done_ = finish_fused_norms();
        be.sync(stream);
        for (int i = 0; i < k; ++i) {
            stack[i].norm = std::sqrt(mail_->norm2[2 * i] + mail_->norm2[2 * i + 1]);      // even rows + odd rows
            stack[i].rows_id = i;
        }
        return SMHIP_OK;
    }

    int merge_layer(const smhip_layer_desc& d, void* out_bf16, float* delta_out, smhip_layer_report* rep) {
        bool transposed = false;
        if (d.k > 1 && d.batch <= 1 && shape_support(d.rows, d.cols, &transposed) && transposed)
            return merge_layer_transposed(d, out_bf16, delta_out, rep);
        return with_select_retry([&] { return merge_layer_once(d, out_bf16, delta_out, rep); });
    }
    void run_transpose(const void* src, void* dst, int R, int C, int esize) {
        TransposeParams q;
        q.src = src; q.dst = dst; q.R = R; q.C = C; q.esize = esize;
        const int grid = ((R + TR_TILE - 1) / TR_TILE) * ((C + TR_TILE - 1) / TR_TILE);
        be.template launch<KTranspose>(grid, 256, LDS_SCRATCH_FLOATS * 4 + TR_TILE * (TR_TILE + 1) * 4, q, stream);
    }
    // The ROW length is the one without a plan, the column length has one: merge the transposed
    // tensors ([C x R]: the rough length becomes the column length, which k_dftp handles) and
    // transpose the result back.  fft2 commutes with the transpose, and norms, order statistics
    // and blends do not care where a bin sits.
    int merge_layer_transposed(const smhip_layer_desc& d, void* out_bf16, float* delta_out, smhip_layer_report* rep) {
        if (d.k < 1 || d.k > SMHIP_MAX_MODELS) return fail(SMHIP_ERR_ARG, "k out of range");
        const size_t n = (size_t)d.rows * d.cols;
        std::vector<std::pair<const void*, int>> srcs;        // distinct (pointer, element size)
        auto slot_of = [&](const void* ptr, int es) {
            for (size_t i = 0; i < srcs.size(); ++i) if (srcs[i].first == ptr) return (int)i;
            srcs.emplace_back(ptr, es);
            return (int)srcs.size() - 1;
        };
        const int es_in = (int)dt_size(d.in_dtype), es_base = (int)dt_size(d.base_out_dtype);
        int ft_slot[SMHIP_MAX_MODELS], base_slot[SMHIP_MAX_MODELS];
        for (int i = 0; i < d.k; ++i) { ft_slot[i] = slot_of(d.finetune[i], es_in); base_slot[i] = slot_of(d.base[i], es_in); }
        int bo_slot = -1;
        for (size_t i = 0; i < srcs.size(); ++i)
            if (srcs[i].first == d.base_out && srcs[i].second == es_base) bo_slot = (int)i;
        if (bo_slot < 0) { srcs.emplace_back(d.base_out, es_base); bo_slot = (int)srcs.size() - 1; }
        const size_t nsrc = srcs.size();
        if (tr_.size() < nsrc + 2) tr_.resize(nsrc + 2);
        int rc;
        for (size_t i = 0; i < nsrc; ++i) {
            if ((rc = ensure(tr_[i], n * srcs[i].second))) return rc;
            run_transpose(srcs[i].first, tr_[i].p, d.rows, d.cols, srcs[i].second);
        }
        if ((rc = ensure(tr_[nsrc], n * 2))) return rc;
        if (delta_out && (rc = ensure(tr_[nsrc + 1], n * 4))) return rc;
        smhip_layer_desc t = d;
        t.rows = d.cols; t.cols = d.rows;
        for (int i = 0; i < d.k; ++i) { t.finetune[i] = tr_[ft_slot[i]].p; t.base[i] = tr_[base_slot[i]].p; }
        t.base_out = tr_[bo_slot].p;
        float* dt = delta_out ? (float*)tr_[nsrc + 1].p : nullptr;
        rc = with_select_retry([&] { return merge_layer_once(t, tr_[nsrc].p, dt, rep); });
        if (rc) return rc;
        run_transpose(tr_[nsrc].p, out_bf16, d.cols, d.rows, 2);
        if (delta_out) run_transpose(dt, delta_out, d.cols, d.rows, 4);
        return SMHIP_OK;
    }
    int merge_layer_once(const smhip_layer_desc& d, void* out_bf16, float* delta_out, smhip_layer_report* rep) {
        if (d.k < 1 || d.k > SMHIP_MAX_MODELS) return fail(SMHIP_ERR_ARG, "k out of range");
        const int R = d.rows, C = d.cols;
        if (R < 1 || C < 1) return fail(SMHIP_ERR_ARG, "bad shape");
        const int batch = d.batch > 1 ? d.batch : 1;          // rank > 2: `batch` slices [R x C], statistics over all of them
        if (batch > 1 && R < 2) return fail(SMHIP_ERR_ARG, "a batch of 1-D slices: pass it as a 2-D tensor");
        const size_t n = (size_t)R * C * batch;
        bool all_aligned = aligned16(d.base_out) && aligned16(out_bf16) && aligned16(delta_out);
        for (int i = 0; i < d.k; ++i) all_aligned = all_aligned && aligned16(d.finetune[i]) && aligned16(d.base[i]);
        // a column length without a plan: `rough` row blocks of Rs rows, combined by k_dftp
        int rough = 1, Rs = R;
        if (d.k > 1 && R > 1) {
            if (!rough_split(R, rough, Rs) || (rough > 1 && batch > 1)) {
                char buf[200];
                snprintf(buf, sizeof buf, "unsupported column length %d%s (needs p * M, p <= %d, M even with factors in "
                         "{2,3,5,7,11,13} and <= %d)", R, batch > 1 ? " for a rank > 2 tensor" : "", DFTP_MAX_P, EMAX * 1024);
                return fail(SMHIP_ERR_SHAPE, buf);
            }
        }
        if (rough == 1 && debug_force_split > 1 && batch == 1 && d.k > 1 && R % debug_force_split == 0) {
            int T; std::vector<int> rad;
            const int M = R / debug_force_split;
            if (M % 2 == 0 && plan_shape(M, T, rad)) { rough = debug_force_split; Rs = M; }
        }
        Geo g = geo(Rs, C, false, all_aligned && rough == 1, rough > 1 ? rough : batch);
        g.rough = rough;
        int rc;
        if (!small_.p && (rc = reserve(1, 1))) return rc;
        clear_flags();
        smhip_layer_report local;
        smhip_layer_report& rp = rep ? *rep : local;
        memset(&rp, 0, sizeof rp);
        rp.merged_delta_norm = -1;

        const bool ref_norms = d.norm_mode == 1;       // torch's CPU norm kernel emulated for every norm the reference takes
        struct RefModeGuard { bool& f; ~RefModeGuard() { f = false; } } ref_guard{ref_mode_};
        ref_mode_ = ref_norms;
        const bool spectral_ok = spectral_inter && (Rs % 2 == 0) && Rs >= 2;
        std::vector<Slot> stack(d.k);
        for (int i = 0; i < d.k; ++i) {
            stack[i].sig = SigDesc{d.finetune[i], d.base[i], d.in_dtype, 1.f};
            stack[i].weight = d.alpha[i];
            stack[i].norm = -1;
        }
        PairOut fin;
        fin.out = out_bf16; fin.out_mode = OUT_BF16; fin.base = d.base_out; fin.base_dtype = d.base_out_dtype; fin.post = 1.f;

        // K = 1: result = base_out + (ft - base)        (tournament loop is skipped)
        if (d.k == 1) {
            SigDesc none{nullptr, nullptr, DT_F32, 1.f};
            int grid;
            run_combine(stack[0].sig, none, 1.f, 0.f, n, delta_out, &fin, true, &grid);
            double na, nb;
            read_norms(grid, na, nb);
            rp.delta_norm[0] = na; rp.target_norm = (double)(float)na + d.target_norm_offset; rp.merged_delta_norm = na;
            return check_flags(false, true, &rp.nan_ifft, &rp.nan_final);
        }
        if ((rc = reserve(g.R, C, false, g.batch))) return rc;

        // norms of every delta.  K == 2: fused into the (speculative) F1 of the only pair.
        std::vector<float> norms32(d.k);
        int f1_grid = -1;
        bool f1_ready = false;
        bool ref_started = false;          // reference_cpu: the deltas' torch.norm emulation runs beside the row passes
        fused_done_ = false;
        fused_.armed = false;
        if (d.k == 2) {
            // reference_cpu: the row pass summarises its deltas for the torch.norm emulation when its plan can
            // (begin_fused_norms), else the separate summary pass runs in front of it with the walker beside it
            const bool fusing = ref_norms && begin_fused_norms(g, stack, n);
            if (ref_norms && !fusing) ref_started = begin_delta_ref_norms(stack, n);
            rc = run_f1(g, stack[0].sig, stack[1].sig, f1_grid);
            if (fusing) ref_started = finish_fused_norms();
            end_delta_ref_norms();
            if (rc) return rc;
            double na, nb;
            read_norms(f1_grid, na, nb);
            stack[0].norm = na; stack[1].norm = nb;
            f1_ready = true;
        } else if (spectral_ok && all_aligned && (C % 8 == 0) && (size_t)d.k * 2 * (size_t)(g.R / 2 + 8) * g.batch <= PART_DOUBLES &&
                   ((ref_started = ref_norms && !begin_fused_norms(g, stack, n) && begin_delta_ref_norms(stack, n)),
                    rows_first(g, stack) == SMHIP_OK)) {
            if (fused_done_) ref_started = true;        // (the row passes took the summaries with them; rows_first has synced)
            // K >= 3: every delta's ROWS are transformed up front, one signal at a time (row pairs);
            // the norms come with it (no separate pass over the inputs), and whichever deltas the
            // pairing puts together only need their column passes afterwards
        } else {
            if (!run_delta_norms(d, n, stack)) {
                SigDesc none{nullptr, nullptr, DT_F32, 1.f};
                for (int i = 0; i < d.k; i += 2) {
                    int grid;
                    const bool two = i + 1 < d.k;
                    run_combine(stack[i].sig, two ? stack[i + 1].sig : none, 0.f, 0.f, n, nullptr, nullptr, true, &grid);
                    double na, nb;
                    read_norms(grid, na, nb);
                    stack[i].norm = na;
                    if (two) stack[i + 1].norm = nb;
                }
            }
        }
        fused_.armed = false;
        if (ref_started && d.k != 2 && !fused_done_) { end_delta_ref_norms(); be.sync(stream); }      // (K = 2: joined in front of read_norms' sync)
        if (ref_started) {
            for (int i = 0; i < d.k; ++i) stack[i].norm = (double)mail_->snorm[i];
        } else if (ref_norms) {
            SigDesc sg[16];
            double nr[16];
            for (int i = 0; i < d.k; ++i) sg[i] = stack[i].sig;
            // (never a silent fall-back to accurate norms: the caller asked for the reference's device="cpu" numerics)
            if (!run_serial_norms(sg, d.k, n, nr))
                return fail(SMHIP_ERR_ARG, "norm_mode = reference_cpu: the torch.norm emulation does not apply to these inputs");
            for (int i = 0; i < d.k; ++i) stack[i].norm = nr[i];
        }
        double mean = 0;
        for (int i = 0; i < d.k; ++i) {
            norms32[i] = (float)stack[i].norm;           // torch.norm of an fp32 tensor is fp32
            rp.delta_norm[i] = norms32[i];
        }
        // A NaN/Inf norm (NaN or Inf element in a finetune or base) makes the pairing find no
        // pair, every slot is carried and the loop below would never end - the reference does
        // spin there.  Deviation: a loud error.
        for (int i = 0; i < d.k; ++i) {
            if (!std::isfinite(norms32[i])) {
                char buf[96];
                snprintf(buf, sizeof buf, "||finetune[%d] - base[%d]|| = %g", i, i, (double)norms32[i]);
                check_flags(false, false);
                return fail(SMHIP_ERR_NONFINITE, buf);
            }
        }
        {   // torch.tensor(layer_norms).mean(): fp32 mean
            float acc = 0.f;
            for (int i = 0; i < d.k; ++i) acc += norms32[i];
            mean = (double)(acc / (float)d.k);
        }
        const double target_norm = mean + d.target_norm_offset;
        rp.target_norm = target_norm;
        double cull_pct = d.cull_start_pct;

        std::vector<char> inter_busy(inter_.size(), 0);
        for (size_t q = 0; q < pool_.size(); ++q) {          // planes an aborted call left detached
            bool working = false;
            for (int i = 0; i < 4; ++i) working = working || pidx_[i] == (int)q;
            pool_busy_[q] = working ? 1 : 0;
        }
        noise_seed_ = noise_seed_base;           // the noise model is a function of (layer step, bin): runs repeat bit for bit
        int step = 0;
        int round_idx = 0;                       // tournament round: the cull fraction goes with it
        int pair_idx = 0;                        // pair merge of this layer, in order: its speculation slot (one guess per
                                                 // (round, position): two pairs of one round have different thresholds - with
                                                 // one guess per round a K = 4 layer missed 22 % of its speculations)
        int deferred_step = -1;
        bool deferred_cut = false, deferred_cull = false;
        while (stack.size() > 1) {
            const int m = (int)stack.size();
            std::vector<std::pair<int, int>> pairs;
            correlated_pairs_least(norms32, m, pairs);   // Q1: first m entries of the ORIGINAL norm list
            std::vector<Slot> next;
            const bool last_round = (pairs.size() == 1 && pairs[0].second >= 0);
            bool any_pair = false;
            for (auto& pr : pairs) any_pair = any_pair || pr.second >= 0;
            if (!any_pair) { check_flags(false, false); return fail(SMHIP_ERR_NONFINITE, "no pair can be formed (non-finite norms)"); }
            for (auto& pr : pairs) {
                const int x = pr.first, y = pr.second;
                if (step < SMHIP_MAX_PAIRS) { rp.step_x[step] = x; rp.step_y[step] = y; }
                if (y < 0) {
                    next.push_back(stack[x]);
                    if (step < SMHIP_MAX_PAIRS) rp.step_branch[step] = SMHIP_BRANCH_CARRY;
                    ++step;
                    continue;
                }
                Slot A = stack[x], Bs = stack[y];
                const double a_w = stack[x].weight, b_w = stack[y].weight;   // Q4: not swapped
                if (A.norm < 0 || Bs.norm < 0) return fail(SMHIP_ERR_ARG, "internal: norm unknown");
                double na = (double)(float)A.norm, nb = (double)(float)Bs.norm;
                bool swapped = false;
                if (std::fabs(na) < std::fabs(nb)) { std::swap(A, Bs); std::swap(na, nb); swapped = true; }
                const double ca = std::fabs(na / target_norm), cb = std::fabs(nb / target_norm);
                const double ratio = cb / (ca + 1e-10);
                // destination: final output when this is the last merge, else an fp32 intermediate
                PairOut po;
                float* inter = nullptr;
                // merge_tensors_fft2_slerp's own ratio test (functions.py:196-202): linear blend below b
                const bool linear = (nb / (na + 1e-10)) < d.b;
                const bool slerp_proper = !(ca < 1e-6) && !(cb < 1e-6 || ratio < 0.1) && !(nb < 1e-4 || na < 1e-4) && !linear;
                if (last_round) {
                    po = fin;
                } else if (slerp_proper && spectral_ok) {
                    // the result stays in the spectral domain: no fp32 buffer
                } else {
                    int id = -1;
                    for (size_t q = 0; q < inter_.size(); ++q) if (!inter_busy[q]) { id = (int)q; break; }
                    if (id < 0) { inter_.emplace_back(); inter_busy.push_back(0); id = (int)inter_.size() - 1; }
                    Buffer* bb = &inter_[id];
                    if ((rc = ensure(*bb, n * sizeof(float)))) return rc;
                    inter_busy[id] = 1;
                    inter = (float*)bb->p;
                    po.out = inter; po.out_mode = OUT_F32; po.post = 1.f;
                }
                int branch;
                int inv_grid = -1;      // >= 0: work-groups of the inverse row pass that left norm partials
                smhip_blend_info info;
                memset(&info, 0, sizeof info);
                double out_norm = -1;
                Slot spec_slot;
                if (!slerp_proper && (A.spectral || Bs.spectral)) {
                    // these branches work on spatial values
                    if ((rc = materialise(g, stack[x], inter_busy))) return rc;
                    if ((rc = materialise(g, stack[y], inter_busy))) return rc;
                    A = swapped ? stack[y] : stack[x]; Bs = swapped ? stack[x] : stack[y];
                }
                bool out_spectral = false;
                if (ca < 1e-6) {
                    branch = SMHIP_BRANCH_ADD;                       // merged = a + b
                    int grid;
                    if (last_round) run_combine(A.sig, Bs.sig, 1.f, 1.f, n, delta_out, &po, false, &grid);
                    else run_combine(A.sig, Bs.sig, 1.f, 1.f, n, inter, nullptr, false, &grid);
                } else if (cb < 1e-6 || ratio < 0.1) {
                    branch = SMHIP_BRANCH_ARITH;
                    const double s = target_norm / na;
                    const double w = b_w / (a_w + 1e-10);
                    PairOut pa = po;
                    float* dtmp = nullptr;
                    if (last_round && delta_out) { pa = PairOut(); pa.out = delta_out; dtmp = delta_out; }
                    pa.ifft_policy = 0;
                    if ((rc = pair_arith(A.sig, Bs.sig, R, C, (float)s, (float)(w * s), 1.0, 1, pa, na, nb, &g))) return rc;
                    if (dtmp) {     // add-back from the fp32 delta
                        SigDesc ds{dtmp, nullptr, DT_F32, 1.f}, none{nullptr, nullptr, DT_F32, 1.f};
                        run_combine(ds, none, 1.f, 0.f, n, nullptr, &fin, false);
                    }
                    f1_ready = false;
                } else {
                    branch = SMHIP_BRANCH_SLERP;
                    const double t = a_w / (a_w + b_w);
                    // merge_tensors_fft2_slerp early-outs on the fp32 norms (functions.py:184-190)
                    if (nb < 1e-4 || na < 1e-4) {
                        SigDesc none{nullptr, nullptr, DT_F32, 1.f};
                        const float sc = (float)(target_norm / na);
                        if (last_round) run_combine(A.sig, none, sc, 0.f, n, delta_out, &po, false);
                        else run_combine(A.sig, none, sc, 0.f, n, inter, nullptr, false);
                        branch = SMHIP_BRANCH_EARLY_V0;
                    } else if (linear) {
                        // R = Fa + t Fb on the normalised spectra, Im included; then * target_norm
                        branch = SMHIP_BRANCH_LINEAR;
                        int grid;
                        if ((rc = run_f1(g, A.sig, Bs.sig, grid))) return rc;          // a in slot 0
                        f1_ready = false;
                        if ((rc = run_f2_linear(g, (float)(1.0 / na), (float)(1.0 / nb), (float)t))) return rc;
                        PairOut ps = po;
                        ps.post = (float)target_norm;
                        float* dtmp = nullptr;
                        if (last_round && delta_out) { ps = PairOut(); ps.out = delta_out; ps.post = (float)target_norm; dtmp = delta_out; }
                        if ((rc = run_inverse(g, plane(g, P_RER), plane(g, P_IMA), nullptr, ps, last_round ? nullptr : &inv_grid))) return rc;
                        if (dtmp) {
                            SigDesc ds{dtmp, nullptr, DT_F32, 1.f}, none{nullptr, nullptr, DT_F32, 1.f};
                            run_combine(ds, none, 1.f, 0.f, n, nullptr, &fin, false);
                        }
                    } else {
                        int im_parts = -1;
                        const bool any_spec = stack[x].spectral || stack[y].spectral ||
                                              (stack[x].rows_id >= 0 && stack[y].rows_id >= 0);
                        const bool fused1d = !any_spec && pair1d_ok(g);       // 1-D: the whole pair merge is one launch
                        if (fused1d) {
                            f1_ready = false;
                        } else if (!any_spec) {
                            if (!(f1_ready && d.k == 2)) {
                                int grid;
                                if ((rc = run_f1(g, stack[x].sig, stack[y].sig, grid))) return rc;
                            }
                            f1_ready = false;
                            // T1 slot 0 holds stack[x], slot 1 holds stack[y]; role "a" is the larger norm
                            const float s0 = (float)(1.0 / (double)(float)stack[x].norm);
                            const float s1 = (float)(1.0 / (double)(float)stack[y].norm);
                            if ((rc = run_f2(g, s0, s1, swapped ? 1 : 0, d.cutoff_pct > 0))) return rc;
                        } else {
                            // at least one input stayed in the spectral domain: bring each input's
                            // planes in on its own (the level-1 histogram accumulates over both)
                            f1_ready = false;
                            int* imp = (!last_round && spectral_ok && fuse_spec_norm) ? &im_parts : nullptr;
                            // both inputs raw with their rows done: ONE column-pass launch for the two of them
                            // (measured on one box, per layer: 1024x8192 -19 %, 14336x4096 -6 %, 8192^2 -2 %, 8192x28672 -0.6 %,
                            //  the folded 28672x8192 +1 %: the launch boundary it saves matters for short passes only)
                            if (f2s_pair && g.batch == 1 && g.fold == 1 && (size_t)g.Cb * g.R <= f2s_pair_max_bins &&
                                !stack[x].spectral && !stack[y].spectral &&
                                stack[x].rows_id >= 0 && stack[y].rows_id >= 0) {
                                const bool xa = !swapped;
                                F2SSecond sec{rowspec_[stack[y].rows_id].p, !xa, (float)(1.0 / (double)(float)stack[y].norm)};
                                if ((rc = run_f2s(g, xa, (float)(1.0 / (double)(float)stack[x].norm), d.cutoff_pct > 0,
                                                  rowspec_[stack[x].rows_id].p, imp, &sec))) return rc;
                            } else
                            for (int side = 0; side < 2; ++side) {
                                const Slot& in = side == 0 ? stack[x] : stack[y];
                                const bool role_a = (side == 0) != swapped;
                                int* ip = role_a ? imp : nullptr;
                                if (in.spectral) {
                                    run_spec_rescale(g, (const float*)pool_[in.re_id].p, (const float*)pool_[in.im_id].p, in.thr,
                                                     (float)in.spec_scale, role_a, d.cutoff_pct > 0, ip);
                                } else if (in.rows_id >= 0) {
                                    if ((rc = run_f2s(g, role_a, (float)(1.0 / (double)(float)in.norm), d.cutoff_pct > 0,
                                                      rowspec_[in.rows_id].p, ip))) return rc;
                                } else {
                                    if ((rc = run_f1_rowpairs(g, in.sig))) return rc;
                                    if ((rc = run_f2s(g, role_a, (float)(1.0 / (double)(float)in.norm), d.cutoff_pct > 0, nullptr, ip))) return rc;
                                }
                            }
                        }
                        bool have_cull;
                        int sel_parts = 0;
                        const bool fused_norm = im_parts > 0 && !safe_select && cull_pct > 0;
                        if (fused1d) have_cull = cull_pct > 0;
                        else spectral_blend(g, BLEND_SLERP, t, d.t_sum, d.cutoff_pct, cull_pct, 1, true, have_cull,
                                            fused_norm ? &sel_parts : nullptr, std::min(pair_idx++, 6));
                        for (int side = 0; side < 2; ++side) {          // consumed spectral inputs give their planes back
                            Slot& in = side == 0 ? stack[x] : stack[y];
                            if (in.spectral) { pool_release(in.re_id); pool_release(in.im_id); in.re_id = in.im_id = -1; }
                        }
                        if (!last_round && spectral_ok) {
                            // the result stays spectral: its norm by Parseval, its planes detached
                            double sre, sim;
                            if (fused_norm && sel_parts > 0) run_spec_norm_fused(sel_parts, im_parts, sre, sim);
                            else run_spec_norm(g, plane(g, P_RER), plane(g, P_IMA), have_cull ? d_thr(1) : nullptr, sre, sim);
                            read_blend_info(&info, d.cutoff_pct > 0, have_cull, true, /*published=*/true);
                            const double ssum = (sre + sim) / ((double)R * (double)C);     // Parseval per [R x C] transform (slices add up)
                            out_spectral = true;
                            spec_slot = Slot();
                            spec_slot.spectral = true;
                            spec_slot.re_id = pidx_[P_RER]; spec_slot.im_id = pidx_[P_IMA];
                            spec_slot.thr = have_cull ? mail_->thr[1] : 0.f;
                            // reference_cpu: the reference takes torch.norm of the materialised tensor
                            // (fast_fourier.py:209-210, functions.py:85); here the tensor does not exist - its norm
                            // is modelled from its exact one (aten_gauss_norm_ratio: the values are Gaussian-like)
                            const double nrm_exact = target_norm * std::sqrt(ssum);
                            const double bias = ref_norms ? aten_gauss_norm_ratio((double)n, nrm_exact / std::sqrt((double)n)) : 1.0;
                            spec_slot.spec_scale = ssum > 0 ? 1.0 / (std::sqrt(ssum) * bias) : 1.0;
                            spec_slot.post = target_norm;
                            spec_slot.norm = nrm_exact * bias;
                            spec_slot.sig = SigDesc{nullptr, nullptr, DT_F32, 1.f};
                            const int r1 = pool_acquire(g.plane_floats * sizeof(float));
                            const int r2 = pool_acquire(g.plane_floats * sizeof(float));
                            if (r1 < 0 || r2 < 0) return SMHIP_ERR_NOMEM;
                            pidx_[P_RER] = r1; pidx_[P_IMA] = r2;
                        } else {
                        PairOut ps = po;
                        ps.post = (float)target_norm;                       // merged * target_norm (fast_fourier.py:243)
                        float* dtmp = nullptr;
                        if (last_round && delta_out) { ps = PairOut(); ps.out = delta_out; ps.post = (float)target_norm; dtmp = delta_out; }
                        if (fused1d) {
                            if ((rc = run_pair1d(g, A.sig, Bs.sig, na, nb, t, d.t_sum, d.cutoff_pct, cull_pct, ps,
                                                 last_round ? nullptr : &inv_grid))) return rc;
                        } else
                        if ((rc = run_inverse(g, plane(g, P_RER), plane(g, P_IMA), have_cull ? d_thr(1) : nullptr, ps,
                                              last_round ? nullptr : &inv_grid))) return rc;
                        if (dtmp) {
                            SigDesc ds{dtmp, nullptr, DT_F32, 1.f}, none{nullptr, nullptr, DT_F32, 1.f};
                            run_combine(ds, none, 1.f, 0.f, n, nullptr, &fin, false);
                        }
                        if (last_round) {      // read with the flags at the end of the layer: one sync less
                            deferred_step = step; deferred_cut = d.cutoff_pct > 0; deferred_cull = have_cull;
                        } else {
                            read_blend_info(&info, d.cutoff_pct > 0, have_cull, true);
                        }
                        }
                    }
                }
                if (branch == SMHIP_BRANCH_SLERP || branch == SMHIP_BRANCH_LINEAR) { info.t = a_w / (a_w + b_w); info.cull_pct = cull_pct; }
                if (step < SMHIP_MAX_PAIRS) { rp.step_branch[step] = branch; rp.step_info[step] = info; }
                ++step;
                for (size_t q = 0; q < inter_.size(); ++q)     // inputs that were intermediates are dead now
                    if (inter_[q].p && (inter_[q].p == stack[x].sig.x || inter_[q].p == stack[y].sig.x)) inter_busy[q] = 0;
                if (!last_round && out_spectral) {
                    spec_slot.weight = (a_w + b_w) / 2.0;
                    next.push_back(spec_slot);
                } else if (!last_round) {
                    // the next round needs ||merged|| (fast_fourier.py:209-210)
                    SigDesc ms{inter, nullptr, DT_F32, 1.f}, none{nullptr, nullptr, DT_F32, 1.f};
                    int grid = inv_grid;             // the inverse row pass summed the squares it stored
                    if (grid < 0) run_combine(ms, none, 0.f, 0.f, n, nullptr, nullptr, true, &grid);
                    double nm, dummy;
                    read_norms(grid, nm, dummy);
                    if (ref_norms) {
                        double nr;
                        if (!run_serial_norms(&ms, 1, n, &nr))
                            return fail(SMHIP_ERR_ARG, "norm_mode = reference_cpu: the torch.norm emulation does not apply to an intermediate");
                        nm = nr;
                    }
                    out_norm = nm;
                    Slot s;
                    s.sig = ms; s.weight = (a_w + b_w) / 2.0; s.norm = out_norm;
                    next.push_back(s);
                }
            }
            stack.swap(next);
            cull_pct = cull_pct / 2.0;
            ++round_idx;
            if (last_round) break;
        }
        rp.n_steps = std::min(step, (int)SMHIP_MAX_PAIRS);
        if (delta_out) {
            SigDesc ds{delta_out, nullptr, DT_F32, 1.f}, none{nullptr, nullptr, DT_F32, 1.f};
            int grid;
            run_combine(ds, none, 0.f, 0.f, n, nullptr, nullptr, true, &grid);
            double nm, dummy;
            read_norms(grid, nm, dummy);
            rp.merged_delta_norm = nm;
        }
        rc = check_flags(true, true, &rp.nan_ifft, &rp.nan_final);
        if (deferred_step >= 0 && deferred_step < SMHIP_MAX_PAIRS) {
            const double t_keep = rp.step_info[deferred_step].t, c_keep = rp.step_info[deferred_step].cull_pct;
            read_blend_info(&rp.step_info[deferred_step], deferred_cut, deferred_cull, true, /*published=*/true);
            rp.step_info[deferred_step].t = t_keep; rp.step_info[deferred_step].cull_pct = c_keep;
        }
        return rc;
    }

    // ---- N3: AdditionMerge / TaskAdditionMerge (one streaming kernel) ------------------------
    int addition_merge(int k, const void* const* fts, const void* base, int dtype, size_t n, int mode, void* out) {
        if (k < 1 || k > ADD_MAX_MODELS) return fail(SMHIP_ERR_ARG, "k out of range");
        AdditionParams a;
        a.k = k; a.base = base; a.dtype = dtype; a.n = n; a.mode = mode; a.out = out;
        bool al = aligned16(base) && aligned16(out);
        for (int i = 0; i < ADD_MAX_MODELS; ++i) { a.ft[i] = fts[i < k ? i : 0]; al = al && aligned16(a.ft[i]); }
        a.vec8 = (n % 8 == 0) && al;
        a.chunks = pick_chunks((n + 7) / 8, 256, 2, 8);
        be.template launch<KAddition>(stream_grid((n + 7) / 8, 256, a.chunks), 256, LDS_SCRATCH_FLOATS * 4, a, stream);
        return SMHIP_OK;
    }

    // ---- function level: slerp (functions.py:24-43), tensor / scalar and the exact norm (functions.py:75-88) ----
    int fn_slerp(const float* v0, const float* v1, size_t rows, size_t cols, float t, float* out) {
        const size_t n = rows * cols;
        if (n == 0) return SMHIP_OK;
        int rc;
        if (!small_.p && (rc = reserve(1, 1))) return rc;
        const int seg = 256 * 64;
        const size_t cch = (cols + seg - 1) / seg;
        if (rows * cch > (size_t)1 << 30) return fail(SMHIP_ERR_ARG, "slerp: tensor too large");
        FnSumsParams a;
        a.v0 = v0; a.v1 = v1; a.n = n; a.chunks = pick_chunks(n, 256, 2, 8);
        const int grid = stream_grid(n, 256, a.chunks);
        if ((rc = ensure(tmpA_, ((size_t)grid * 4 + rows * cch) * sizeof(double) + (rows + 4) * sizeof(float)))) return rc;
        a.partials = (double*)tmpA_.p;
        double* part = a.partials + (size_t)grid * 4;
        float* den = (float*)(part + rows * cch);
        float* consts = den + rows;
        be.template launch<KFnSums>(grid, 256, LDS_SCRATCH_FLOATS * 4, a, stream);
        FnSlerpFinParams f;
        f.partials = a.partials; f.nparts = grid; f.t = t; f.consts = consts;
        be.template launch<KFnSlerpFin>(1, 256, LDS_SCRATCH_FLOATS * 4, f, stream);
        FnSlerpRowsParams r;
        r.v0 = v0; r.v1 = v1; r.out = out; r.rows = rows; r.cols = cols; r.cchunks = (int)cch; r.seg = seg;
        r.consts = consts; r.part = part; r.den = den;
        be.template launch<KFnSlerpRows0>((int)(rows * cch), 256, LDS_SCRATCH_FLOATS * 4, r, stream);
        FnSlerpDenParams d;
        d.part = part; d.rows = rows; d.cchunks = (int)cch; d.den = den;
        be.template launch<KFnSlerpDen>((int)((rows + 255) / 256), 256, LDS_SCRATCH_FLOATS * 4, d, stream);
        be.template launch<KFnSlerpRows1>((int)(rows * cch), 256, LDS_SCRATCH_FLOATS * 4, r, stream);
        return SMHIP_OK;
    }
    int fn_div_scalar(const void* x, int dtype, size_t n, float s, void* out) {
        if (n == 0) return SMHIP_OK;
        DivScalarParams q;
        q.x = x; q.out = out; q.dtype = dtype; q.n = n; q.s = s; q.chunks = pick_chunks(n, 256, 2, 8);
        be.template launch<KDivScalar>(stream_grid(n, 256, q.chunks), 256, LDS_SCRATCH_FLOATS * 4, q, stream);
        return SMHIP_OK;
    }
    int fn_exact_norm(const void* x, int dtype, size_t n, double* norm_out) {
        *norm_out = 0.0;
        if (n == 0) return SMHIP_OK;
        int rc;
        if (!small_.p && (rc = reserve(1, 1))) return rc;
        SumsqAnyParams q;
        q.x = x; q.dtype = dtype; q.n = n; q.chunks = pick_chunks(n, 256, 2, 8);
        const int grid = stream_grid(n, 256, q.chunks);
        if ((rc = ensure(tmpA_, (size_t)grid * 2 * sizeof(double)))) return rc;
        q.partials = (double*)tmpA_.p;
        be.template launch<KSumsqAny>(grid, 256, LDS_SCRATCH_FLOATS * 4, q, stream);
        SumPartialsParams sp;
        sp.partials = q.partials; sp.nparts = grid; sp.out = mail_->norm2;
        be.template launch<KSumPartials>(1, 256, LDS_SCRATCH_FLOATS * 4, sp, stream);
        be.sync(stream);
        *norm_out = std::sqrt(mail_->norm2[0]);
        return SMHIP_OK;
    }

    // ---- correlate_pairs (legacy operator's pairing matrix, functions.py:304-314) -------------
    int correlate_pairs(int k, const void* const* tensors, int dtype, size_t rows, size_t cols, float* matrix_out) {
        if (k < 2 || k > CORR_MAX_K) return fail(SMHIP_ERR_ARG, "correlate_pairs: 2 <= k <= 8");
        int rc;
        if (!small_.p && (rc = reserve(1, 1))) return rc;
        const int P = k * (k + 1) / 2;
        const size_t cblocks = (cols + 255) / 256;
        int splits = (int)std::max<size_t>(1, std::min<size_t>(64, (size_t)1024 / cblocks));
        splits = (int)std::min<size_t>((size_t)splits, std::max<size_t>(rows, 1));
        if ((rc = ensure(tmpA_, (size_t)splits * P * cols * sizeof(double)))) return rc;
        CorrPartialParams a;
        a.k = k; a.dtype = dtype; a.rows = rows; a.cols = cols; a.splits = splits; a.part = (double*)tmpA_.p;
        for (int i = 0; i < CORR_MAX_K; ++i) a.t[i] = tensors[i < k ? i : 0];
        be.template launch<KCorrPartial>((int)(cblocks * splits), 256, LDS_SCRATCH_FLOATS * 4, a, stream);
        CorrFinishParams f;
        f.k = k; f.cols = cols; f.splits = splits; f.part = (const double*)tmpA_.p; f.eps = 1e-8f;
        f.out = (float*)mail_->corr;
        for (int i = 0; i < k * k; ++i) mail_->corr[i] = 0.f;
        be.template launch<KCorrFinish>(k * (k - 1) / 2, 256, LDS_SCRATCH_FLOATS * 4, f, stream);
        be.sync(stream);
        for (int i = 0; i < k * k; ++i) matrix_out[i] = mail_->corr[i];
        return SMHIP_OK;
    }

    // ---- transforms for the function-level API -----------------------------------------
    int fft_transform(const float* x, int R, int C, float* spectrum) {
        const Geo g = geo(R, C);
        int rc = reserve(R, C);
        if (rc) return rc;
        SigDesc a{x, nullptr, DT_F32, 1.f}, none{nullptr, nullptr, DT_F32, 1.f};
        int grid;
        if ((rc = run_f1(g, a, none, grid))) return rc;
        if ((rc = run_f2(g, 1.f, 1.f, 0, false))) return rc;
        ExpandParams e;
        e.re = plane(g, P_REA); e.im = plane(g, P_IMA); e.R = R; e.C = C; e.Cb = g.Cb; e.full = (cf2*)spectrum; e.chunks = 8;
        be.template launch<KExpand>(stream_grid((size_t)R * C, 256, e.chunks), 256, LDS_SCRATCH_FLOATS * 4, e, stream);
        return SMHIP_OK;
    }
    // real part of the inverse transform of an arbitrary complex spectrum:
    // Hermitian-symmetrise while packing, then the c2r inverse.
    int ifft_transform(const float* spectrum, int R, int C, float* out) {
        const Geo g = geo(R, C);
        int rc = reserve(R, C);
        if (rc) return rc;
        be.memset(d_flags(), 0, 32, stream);         // not the selection counters: blend_full calls this mid-way
        PackParams pk;
        pk.full = (const cf2*)spectrum; pk.R = R; pk.C = C; pk.Cb = g.Cb; pk.re = plane(g, P_RER); pk.im = plane(g, P_IMA);
        pk.chunks = 8; pk.sym = 1;
        be.template launch<KPack>(stream_grid((size_t)g.Cb * R, 256, pk.chunks), 256, LDS_SCRATCH_FLOATS * 4, pk, stream);
        PairOut po;
        po.out = out; po.out_mode = OUT_F32; po.post = 1.f;
        return run_inverse(g, plane(g, P_RER), plane(g, P_IMA), nullptr, po);
    }

    // A5-A7 on full complex spectra [R][C] (no Hermitian assumption: every bin has
    // weight 1 and the planes simply keep the row-major order of the input).
    int blend_full(const float* f0, const float* f1, int R, int C, int mode, double t, double t_sum, double cutoff_pct,
                   double cull_pct, int agreement, int do_imag, float* out_spec, smhip_blend_info* info) {
        clear_flags();
        int rc = blend_full_once(f0, f1, R, C, mode, t, t_sum, cutoff_pct, cull_pct, agreement, do_imag, out_spec, info);
        if (rc == SMHIP_OK && !safe_select && select_overflowed()) {
            safe_select = true;
            rc = blend_full_once(f0, f1, R, C, mode, t, t_sum, cutoff_pct, cull_pct, agreement, do_imag, out_spec, info);
            safe_select = false;
        }
        return rc;
    }
    int blend_full_once(const float* f0, const float* f1, int R, int C, int mode, double t, double t_sum, double cutoff_pct,
                        double cull_pct, int agreement, int do_imag, float* out_spec, smhip_blend_info* info) {
        const Geo g = geo(R, C, true);
        int rc = reserve(R, C, true);
        if (rc) return rc;
        const size_t total = (size_t)R * C;
        const int sgrid = stream_grid(total, 256, 8);
        const size_t slds = LDS_SCRATCH_FLOATS * 4;
        if ((rc = ensure(fullS_, 2 * total * sizeof(float)))) return rc;
        float* im0 = (float*)fullS_.p;
        float* im1 = im0 + total;
        SplitParams sp;
        sp.n = total; sp.chunks = 8;
        sp.full = (const cf2*)f0; sp.re = plane(g, P_REA); sp.im = im0;
        be.template launch<KSplit>(sgrid, 256, slds, sp, stream);
        sp.full = (const cf2*)f1; sp.re = plane(g, P_REB); sp.im = im1;
        be.template launch<KSplit>(sgrid, 256, slds, sp, stream);
        bool have_cull;
        spectral_blend(g, mode, t, t_sum, cutoff_pct, cull_pct, agreement, false, have_cull);
        if (have_cull) {   // Re R[|Re R| < thr] = 0   (functions.py:147)
            CullParams cp;
            cp.x = plane(g, P_RER); cp.n = total; cp.thr = d_thr(1); cp.chunks = 8;
            be.template launch<KCull>(sgrid, 256, slds, cp, stream);
        }
        read_blend_info(info, mode == BLEND_SLERP && cutoff_pct > 0, have_cull, mode == BLEND_SLERP);
        if (info) { info->t = t; info->cull_pct = cull_pct; }
        JoinParams jp;
        jp.n = total; jp.chunks = 8; jp.full = (cf2*)out_spec;
        if (!do_imag) {
            jp.re = plane(g, P_RER); jp.im = im0;
            be.template launch<KJoin>(sgrid, 256, slds, jp, stream);
            return SMHIP_OK;
        }
        // imaginary detour (functions.py:152-158 / :290-298): transform both imaginary
        // planes, blend them with interp_imag / do_imag = False, inverse, keep .real
        if ((rc = ensure(saveR_, total * sizeof(float)))) return rc;
        if ((rc = ensure(saveI_, total * sizeof(float)))) return rc;
        if ((rc = ensure(tmpA_, 2 * total * sizeof(float)))) return rc;
        if ((rc = ensure(tmpB_, 2 * total * sizeof(float)))) return rc;
        if ((rc = ensure(tmpC_, 2 * total * sizeof(float)))) return rc;
        SigDesc rr{plane(g, P_RER), nullptr, DT_F32, 1.f}, none{nullptr, nullptr, DT_F32, 1.f};
        run_combine(rr, none, 1.f, 0.f, total, (float*)saveR_.p, nullptr, false);
        if ((rc = fft_transform(im0, R, C, (float*)tmpA_.p))) return rc;
        if ((rc = fft_transform(im1, R, C, (float*)tmpB_.p))) return rc;
        if ((rc = blend_full_once((const float*)tmpA_.p, (const float*)tmpB_.p, R, C, mode, t, 1.0, 0, 0, agreement, 0,
                                  (float*)tmpC_.p, nullptr))) return rc;
        if ((rc = ifft_transform((const float*)tmpC_.p, R, C, (float*)saveI_.p))) return rc;
        jp.re = (const float*)saveR_.p; jp.im = (const float*)saveI_.p;
        be.template launch<KJoin>(sgrid, 256, slds, jp, stream);
        return SMHIP_OK;
    }

    // profiling table lives in the backend
  private:
    std::map<int, HostPlan> plans_;
    Buffer t1_, small_, tmpA_, tmpB_, tmpC_, fullS_, saveR_, saveI_, cand_, aten_, emf_;
    std::vector<Buffer> pool_ = std::vector<Buffer>(4);
    std::vector<char> pool_busy_ = std::vector<char>(4, 1);
    int pidx_[4] = {0, 1, 2, 3};
    uint32_t noise_seed_ = 0;
    uint32_t cap_keys_ = 0, cap_pairs_ = 0;
    bool overflow_seen_ = false;
    bool flags_clean_ = false;
    Mailbox* mail_ = nullptr;
    std::vector<Buffer> inter_;
    std::vector<Buffer> rowspec_;          // K >= 3: row spectra of the raw deltas (rows_first)
    std::vector<Buffer> tr_;               // transposed copies of a layer's tensors (rough ROW length)
    std::map<int, void*> rough_tw_;        // twiddles of the column lengths without a plan
    std::map<int, BlueHost> blue_;         // chirp tables of the row lengths without a plan
    std::vector<double> host_part_;
};

}  // namespace smhip
