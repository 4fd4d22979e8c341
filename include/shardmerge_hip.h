/* shardmerge_hip.h - C ABI of the MI355X-native spectral-merge hot path.
 *
 * This is the drop-in boundary for shardmerge's per-layer merge
 * (reference: shard/merge/fast_fourier.py:103-276 calling
 * shard/tensor/functions.py:24-302).  Every pointer marked "device" is a HIP
 * device pointer (e.g. torch.Tensor.data_ptr() on PyTorch-ROCm); `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  The library borrows
 * input pointers for the duration of a call, never mutates inputs, writes only
 * into caller-provided outputs and owns nothing but its internal workspace
 * (grown on demand with hipMalloc, reused across calls).  One context serves
 * one caller thread / one device.
 *
 * All functions return SMHIP_OK (0) or an error code; smhip_last_error() gives
 * the message.  The Python wrapper maps SMHIP_ERR_INF_IFFT / _INF_MERGED to the
 * reference's ValueError texts (functions.py:217, fast_fourier.py:274).
 */
#ifndef SHARDMERGE_HIP_H
#define SHARDMERGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smhip_ctx smhip_ctx;

enum {
    SMHIP_OK = 0,
    SMHIP_ERR_HIP = 1,         /* a HIP runtime call failed */
    SMHIP_ERR_SHAPE = 2,       /* unsupported shape, see smhip_length_supported / smhip_shape_supported */
    SMHIP_ERR_INF_IFFT = 3,    /* "Inf in ifft output"            (functions.py:215-217) */
    SMHIP_ERR_INF_MERGED = 4,  /* "Inf in merged tensor for ..."  (fast_fourier.py:273-274) */
    SMHIP_ERR_ARG = 5,
    SMHIP_ERR_NOMEM = 6,
    SMHIP_ERR_NONFINITE = 7    /* a delta norm is NaN/Inf (K >= 2): the reference's tournament loop
                                  (fast_fourier.py:171-254) never terminates on such input */
};

enum { SMHIP_BF16 = 0, SMHIP_F16 = 1, SMHIP_F32 = 2 };

/* branch taken by a pair merge (fast_fourier.py:223 / :226 / :233) */
enum { SMHIP_BRANCH_ADD = 0, SMHIP_BRANCH_ARITH = 1, SMHIP_BRANCH_SLERP = 2, SMHIP_BRANCH_CARRY = 3,
       SMHIP_BRANCH_EARLY_V0 = 4 /* functions.py:184-190 */, SMHIP_BRANCH_LINEAR = 5 /* functions.py:199-202 */ };

#define SMHIP_MAX_MODELS 16
#define SMHIP_MAX_PAIRS 32

/* ---- lifetime ---------------------------------------------------------- */
int smhip_create(int device, smhip_ctx** out);
void smhip_destroy(smhip_ctx* ctx);
const char* smhip_last_error(smhip_ctx* ctx);
const char* smhip_version(void);
/* pre-size the workspace for [rows x cols] tensors (optional; it grows on demand) */
int smhip_reserve(smhip_ctx* ctx, int rows, int cols);
size_t smhip_workspace_bytes(smhip_ctx* ctx);
/* 0 if a length-n transform has a work-group plan (n <= 32768 with prime factors <= 13; what the
 * function-level entry points A4-A10 need of both lengths), SMHIP_ERR_SHAPE otherwise */
int smhip_length_supported(int n);
/* 0 if smhip_merge_layer takes a [rows x cols] tensor (rows = 1 for 1-D): one length must have a
 * plan; the other may also be p * M with M planned and even and p <= 256 - 11008 = 43 * 256,
 * 18944 = 37 * 512, 65536 = 2 * 32768, 128256 = 167 * 768 - which costs one extra pass over the
 * row spectra each way (and, when it is the ROW length, a transpose of the operands).  A tensor without
 * any planned length (Falcon-7B: 4544 = 71 * 64, 4672 = 73 * 64) is taken when one length splits as
 * above and the other, of any factorisation and parity, is <= 16384: its rows go through the chirp-z
 * row passes (sm_bluestein.hpp), 3-4x the arithmetic of a planned length.  The function-level entry
 * points A4-A10 take such a ROW length too (their column length needs a plan). */
int smhip_shape_supported(int rows, int cols);

/* ---- A4 / A8: transforms (reference fft_transform / ifft_transform,
 *      functions.py:45-73).  x: device float[rows*cols] (rows = 1 for 1-D);
 *      spectrum: device interleaved complex64 [rows][cols]. ------------------- */
int smhip_fft_transform(smhip_ctx* ctx, const float* x, int rows, int cols, float* spectrum, void* stream);
int smhip_ifft_transform(smhip_ctx* ctx, const float* spectrum, int rows, int cols, float* real_out, void* stream);

/* ---- A5-A7: spectrum-level blends on full complex spectra
 *      (interpolate_fft_components functions.py:90-162,
 *       arithmetic_fft_components functions.py:256-302). --------------------- */
typedef struct {
    double cutoff_threshold, cull_threshold;
    double dot, s00, s01, s11;
    uint64_t n_slerp;
    double t;          /* slerp fraction used (fast_fourier.py:234: weights NOT swapped with a/b, quirk Q4) */
    double cull_pct;   /* cull fraction used (halved every tournament round, fast_fourier.py:254) */
} smhip_blend_info;
int smhip_interpolate_fft_components(smhip_ctx* ctx, const float* f0, const float* f1, int rows, int cols,
                                     double t, double t_sum, double cutoff_pct, double cull_pct, int interp_imag,
                                     float* out_spectrum, smhip_blend_info* info, void* stream);
int smhip_arithmetic_fft_components(smhip_ctx* ctx, const float* f0, const float* f1, int rows, int cols,
                                    double t, int agreement, int do_imag, float* out_spectrum, void* stream);

/* ---- A9: merge_tensors_fft2_slerp (functions.py:164-221).
 *      v0, v1, out: device float[rows*cols].  *branch reports which path ran
 *      (SLERP, EARLY_V0 or LINEAR). ------------------------------------------ */
int smhip_merge_tensors_fft2_slerp(smhip_ctx* ctx, const float* v0, const float* v1, int rows, int cols,
                                   double t, double b, double t_sum, double cutoff_pct, double cull_pct,
                                   float* out, double* norm0, double* norm1, int* branch,
                                   smhip_blend_info* info, void* stream);

/* ---- A10: task_arithmetic_fft2 (functions.py:224-254) ------------------- */
int smhip_task_arithmetic_fft2(smhip_ctx* ctx, const float* v0, const float* v1, int rows, int cols,
                               double t, int agreement, float* out, void* stream);

/* ---- A1-A13 fused: the block-tensor branch of FourierMerge._merge_layer
 *      (fast_fourier.py:132-276 + base.py:117-137). ------------------------- */
typedef struct {
    int k;                                  /* models that pass use_layer_index */
    const void* finetune[SMHIP_MAX_MODELS]; /* device, in_dtype, [rows*cols] */
    const void* base[SMHIP_MAX_MODELS];     /* device, in_dtype: each model's own base */
    double alpha[SMHIP_MAX_MODELS];
    int in_dtype;
    const void* base_out;                   /* device: output_base_model's tensor */
    int base_out_dtype;
    int rows, cols;                         /* rows = 1 for 1-D tensors */
    double target_norm_offset;              /* 1e-10 */
    double cull_start_pct;                  /* 0.20 */
    double cutoff_pct;                      /* 0.08 */
    double t_sum;                           /* 1.0  */
    double b;                               /* 0.1: merge_tensors_fft2_slerp's norm-ratio threshold below which
                                               the pair is blended linearly, R = F0 + t F1 (functions.py:164,196-202);
                                               unreachable at the default (the layer's own rule routes ratios < 0.1
                                               to Arithmetic-FFT first, fast_fourier.py:226) */
    int norm_mode;                          /* 1 "reference_cpu" - what every host entry point of this package passes by
                                               default (shardmerge_amd/constants.py: DEFAULT_NORM_MODE): every norm the
                                               reference takes as torch's CPU kernel returns it - acc = fma(x, x, acc)
                                               serially in 8 fp32 lanes, biased by -5e-3 at 67 M elements - which the
                                               reference's device="cpu" output depends on at the 1e-2 level.  EXACT (bit
                                               for bit, csrc/sm_aten_norm.hpp) for spatial tensors: the deltas and
                                               materialised intermediates; MODELLED for what has no spatial form here:
                                               the gathered slerp-class vectors (sampled statistics, 1e-6 ... 4e-6 of
                                               torch's value) and a K >= 3 intermediate kept in the spectral domain
                                               (Gaussian model of its exact Parseval norm, ~1e-5).  Costs ~0.3 ms per
                                               8192 x 8192 layer.  Inputs that are not 16-byte aligned are loaded element
                                               by element; the mode never falls back to other numerics silently (an
                                               input it cannot handle is SMHIP_ERR_ARG).
                                               0 "exact": accurate L2 norms - the reference's device="cuda" numerics */
    int batch;                              /* 0 / 1: one [rows x cols] tensor.  > 1: a rank > 2 tensor, `batch` contiguous
                                               slices [rows x cols]: each slice is transformed on its own (the reference's
                                               fftn(dim=(-2,-1)), functions.py:58) while every norm, order statistic and
                                               slerp sum runs over the whole tensor, as the reference's flat ops do */
} smhip_layer_desc;

typedef struct {
    double target_norm;
    double delta_norm[SMHIP_MAX_MODELS];
    int n_steps;                            /* pair merges + carries, in order */
    int step_x[SMHIP_MAX_PAIRS], step_y[SMHIP_MAX_PAIRS];   /* stack indices (y = -1: carry) */
    int step_branch[SMHIP_MAX_PAIRS];
    smhip_blend_info step_info[SMHIP_MAX_PAIRS];
    uint32_t nan_ifft, nan_final;           /* NaNs replaced by 0 (functions.py:211-213, fast_fourier.py:270-271) */
    double merged_delta_norm;               /* || result before add-back || when available, else -1 */
} smhip_layer_report;

/* out: device bf16 [rows*cols] (the reference hard-casts block tensors to bf16,
 * fast_fourier.py:276).  delta_out (optional, may be NULL): device float of the
 * merged delta before add-back, for parity checks. */
int smhip_merge_layer(smhip_ctx* ctx, const smhip_layer_desc* desc, void* out_bf16, float* delta_out,
                      smhip_layer_report* report, void* stream);

/* ---- N3: AdditionMerge / TaskAdditionMerge (reference shard/merge/addition.py:70-76,
 *      shard/merge/taskaddition.py:69-79): out = sum_i (finetune_i - base), optionally keeping per
 *      element only the deltas whose sign equals the majority sign.  All tensors: device, `dtype`,
 *      n elements; the arithmetic is the dtype's, as torch does it on CPU (16-bit ops are fp32 ops
 *      rounded to the dtype).  The base is NOT added back (the reference does not). ------------ */
int smhip_addition_merge(smhip_ctx* ctx, int k, const void* const* finetunes, const void* base, int dtype, size_t n,
                         int sign_agreement, void* out, void* stream);

/* ---- slerp (reference shard/tensor/functions.py:24-43) on fp32 device tensors of rows x cols elements (1-D:
 *      rows = 1): the cosine is taken between the UN-normalised vectors over the whole tensor, the relative vector
 *      v1 - v0 dot is normalised along the LAST dimension (F.normalize(dim=-1), eps 1e-12), out = v0 cos + rel sin.
 *      rows * cols = 0: nothing to do.  A zero vector gives NaN, as in the reference. ---- */
int smhip_slerp(smhip_ctx* ctx, const float* v0, const float* v1, size_t rows, size_t cols, float t, float* out, void* stream);

/* ---- the two halves of normalize_tensor (functions.py:75-88: norm = tensor.norm().item(); tensor / norm):
 *      smhip_exact_norm - ||x||_2 accumulated in fp64 (the norm as torch.norm computes it on CPU, rounding bias
 *      included, is smhip_reference_cpu_norm); norm_out: HOST double.
 *      smhip_div_scalar - out = x / s in x's dtype (a 16-bit tensor over a Python float: an fp32 division rounded
 *      to the dtype, as torch does it); out may be x. ---- */
int smhip_exact_norm(smhip_ctx* ctx, const void* x, int dtype, size_t n, double* norm_out, void* stream);
int smhip_div_scalar(smhip_ctx* ctx, const void* x, int dtype, size_t n, float s, void* out, void* stream);

/* ---- correlate_pairs (reference shard/tensor/functions.py:304-314, the legacy fourier.py operator's
 *      pairing matrix): matrix[i][j] = mean over the trailing positions of
 *      cosine_similarity(t_i, t_j, dim=0).nan_to_num(0); zero diagonal.  tensors: k (2..8) device
 *      tensors of `dtype`, viewed as [rows = shape[0]] x [cols = the rest] (1-D: cols = 1).
 *      matrix_out: HOST float[k*k].  (The legacy operator's task_add_models post-pass,
 *      fourier.py:191-196, is smhip_task_arithmetic_fft2(result, delta, t = 1, agreement = 0).) ---- */
int smhip_correlate_pairs(smhip_ctx* ctx, int k, const void* const* tensors, int dtype, size_t rows, size_t cols,
                          float* matrix_out, void* stream);

/* ---- torch.norm(x - base) as ATen's CPU kernel returns it for a contiguous fp32 tensor (the reference's
 *      device="cpu" norms: shard/tensor/functions.py:36,40,85, shard/merge/fast_fourier.py:152,209-210):
 *      acc = fma(x, x, acc) - one rounding per element - serially in 8 fp32 lanes, lanes added in order, the n % 8 tail,
 *      sqrt - reproduced bit for bit by a parallel algorithm (csrc/sm_aten_norm.hpp).  x, base (may be
 *      NULL): device, `dtype`, n elements, 16-byte aligned.  norm_out: HOST float.  This is what
 *      smhip_layer_desc::norm_mode = 1 uses for every spatial norm. -------------------------------- */
int smhip_reference_cpu_norm(smhip_ctx* ctx, const void* x, const void* base, int dtype, size_t n, float* norm_out,
                             void* stream);

/* ---- test hooks: "cand_cap" clamps the capacity of the selection passes' candidate
 *      lists (0 = default) so that the overflow fallback can be exercised;
 *      "sel_chunks" sets the steps per thread of the level-2 selection pass,
 *      "force_split" = p splits the column length of smhip_merge_layer into p row blocks (the path of
 *      lengths without a plan, smhip_shape_supported) although it has one, 0 = off;
 *      "force_bluestein" = 1 sends every row transform through the chirp-z row passes (the path of ROW lengths
 *      without a plan);
 *      "pair1d" = 0 sends 1-D tensors through the multi-kernel pipeline instead of the one-launch pair merge;
 *      "spec_cull" = 0 keeps the SLERP blend and the cull's selection pass apart (no speculation on the
 *      threshold's level-1 bin); "spec_min_bins" = the smallest spectrum (bins) that speculates (default 2^20);
 *      "dftp_pairs" = 0 runs the generic p-point DFT kernel also for p <= 126;
 *      "sel_wgs_per_cu" the resident work-groups per CU its grid is sized for (0 = default 5), and
 *      "sel_flush_always" flushes its staged candidates after every round (the
 *      mid-stream flush that only very large tensors reach otherwise);
 *      "noise_seed" = s offsets the seed of the rounding-noise model that a spectral intermediate's culled bins
 *      receive (another realisation of the same statistics; default 0);
 *      "spectral_intermediates" = 0 makes a K >= 3 tournament materialise every intermediate
 *      (inverse transform to fp32, forward again) as round 1 of this library did, instead of
 *      keeping it in the spectral domain (default 1). ----------------------------------- */
int smhip_debug_option(smhip_ctx* ctx, const char* key, long value);
/* test hook, read side: "spec_hit" = 1 if the last speculative blend's guess was confirmed, 0 if it was voided;
 * "spec_checked" / "spec_hits": how many speculations this context has checked / confirmed since it was created;
 * "aten_fast" / "aten_group" / "aten_slow": chunks the last smhip_reference_cpu_norm composed from their summaries /
 * crossed with the group summaries / walked cooperatively
 * ("aten_serial" = 1 as an option selects the old serial single-work-group chain instead) */
int smhip_debug_query(smhip_ctx* ctx, const char* key, long* value);

/* ---- profiling: per-kernel device time measured with HIP events on the
 *      caller's stream (bench.py's roofline leg) ---------------------------- */
int smhip_profile_enable(smhip_ctx* ctx, int on);
int smhip_profile_reset(smhip_ctx* ctx);
int smhip_profile_count(smhip_ctx* ctx);
/* i-th kernel: name, number of launches, total milliseconds */
int smhip_profile_get(smhip_ctx* ctx, int i, const char** name, uint64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* SHARDMERGE_HIP_H */
