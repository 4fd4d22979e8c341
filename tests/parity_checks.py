"""Parity checks shared by the two test tiers:

  * tests/test_emul_parity.py (no GPU): the product's kernel bodies and host
    orchestration run in the CPU work-group emulator (tests/emul);
  * tests/test_gpu_parity.py (-m gpu): the same checks through
    libshardmerge_hip.so on a real MI355X.

Every check compares against the golden vectors produced by the reference
(tests/golden) and/or the CPU oracle (oracle/spectral_oracle.py).

Tolerances.  Transforms and all branch-free paths: 2e-6 normwise.  The SLERP
branch takes two order statistics of the data; exactly one bin (and its
conjugate twin, which the reference computes separately and which differs by an
ulp) sits ON each threshold, so any implementation that is not bit-identical to
torch's CPU FFT may classify that bin differently (SURVEY 8a "parity floor").
`spectral_residual` therefore measures the difference after dropping the few
largest bins of its spectrum: the rest must agree to 2e-5, and at most
MAX_TIE_BINS bins may carry the rest of the difference.  For K >= 3 the
reference's later rounds take decisions on bins that round 1 culled to zero and
that come back from its ifft->fft round trip as rounding noise (see
oracle/chaos_probe.py): there the bar is the reference's own reproducibility
floor, measured when the goldens were generated (manifest "layer_self_floor").
"""
import math

import torch

from oracle import spectral_oracle as so
from tests.golden import inputs as gi

TOL = 2e-6
MAX_TIE_BINS = 8


def spectral_residual(x: torch.Tensor, ref: torch.Tensor, drop: int = MAX_TIE_BINS):
    """(total relative error, relative error after removing the `drop` largest
    bins of the difference's spectrum)."""
    d = x.double().cpu() - ref.double().cpu()
    r = ref.double().cpu()
    D = torch.fft.fftn(d) if d.ndim > 1 else torch.fft.fft(d)
    mag2 = (D.real ** 2 + D.imag ** 2).flatten()
    ref2 = float((r ** 2).sum()) * r.numel()
    tot = float(mag2.sum())
    top = torch.topk(mag2, min(drop, mag2.numel())).values.sum().item()
    if ref2 == 0:
        return math.sqrt(tot), math.sqrt(max(tot - top, 0.0))
    return math.sqrt(tot / ref2), math.sqrt(max(tot - top, 0.0) / ref2)


def check_fft(engine, golden, case):
    x = gi.fft_input(case)
    f = engine.fft_transform(x).cpu()
    ref = torch.complex(golden.get("g1_fft.safetensors", case["id"] + ".fft.re"),
                        golden.get("g1_fft.safetensors", case["id"] + ".fft.im"))
    assert f.shape == ref.shape and f.dtype == torch.complex64
    assert so.rel_err(torch.view_as_real(f), torch.view_as_real(ref)) < TOL
    back = engine.ifft_transform(ref).cpu()
    assert so.rel_err(back, golden.get("g1_fft.safetensors", case["id"] + ".ifft")) < TOL
    assert torch.allclose(engine.ifft_transform(f).cpu(), x, atol=1e-4)   # reference test_functions.py:92-121


def check_interp(engine, golden, case):
    a, b = gi.pair_input(case)
    fa, fb = so.fft_transform(a), so.fft_transform(b)
    tr = so.BlendTrace()
    so.interpolate_fft_components(fa, fb, t=case["t"], t_sum=case["t_sum"], cutoff_pct=case["cutoff"],
                                  cull_pct=case["cull"], interp_imag=False, trace=tr)
    out, rep = engine.interpolate_fft_components(fa, fb, case["t"], case["t_sum"], case["cutoff"], case["cull"], case["imag"])
    out = out.cpu()
    ref_re = golden.get("g2_interp.safetensors", case["id"] + ".re")
    ref_im = golden.get("g2_interp.safetensors", case["id"] + ".im")
    # full complex spectra in, nothing recomputed: no FFT rounding between us and
    # the reference, so thresholds and classes must agree exactly
    assert so.rel_err(out.real, ref_re) < TOL
    # interp_imag=True: the reference blends Re(fft(Im F)), which for a real input is
    # pure rounding noise.  Its effect - the distance of the reference's
    # Im R from Im F0 - is not reproducible; we must land in the same noise band.
    noise_band = so.rel_err(ref_im, fa.imag)
    assert so.rel_err(out.imag, ref_im) <= (3 * noise_band + 1e-5 if case["imag"] else TOL)
    assert rep.n_slerp == tr.n_slerp
    assert rep.cutoff_threshold == tr.cutoff_threshold
    assert abs(rep.cull_threshold - tr.cull_threshold) <= 2e-6 * abs(tr.cull_threshold)   # a blended value: ulps
    if not case["imag"]:
        assert torch.equal(out.imag, fa.imag)           # reference test_functions.py:389-402


def check_pair(engine, golden, case):
    a, b = gi.pair_input(case)
    tr = so.BlendTrace()
    ref_o, n0_o, n1_o = so.merge_tensors_fft2_slerp(a, b, t=case["t"], b=case["b"], t_sum=case["t_sum"],
                                                    cutoff_pct=case["cutoff"], cull_pct=case["cull"], trace=tr)
    out, n0, n1, rep = engine.merge_tensors_fft2_slerp(a, b, case["t"], b=case["b"], t_sum=case["t_sum"],
                                                       cutoff_pct=case["cutoff"], cull_pct=case["cull"])
    out = out.cpu()
    ref = golden.get("g4_pair.safetensors", case["id"])
    meta = golden.manifest["pair_meta"][case["id"]]
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert n0 == torch.tensor(meta["n0"]).item() or abs(n0 - meta["n0"]) <= 2e-6 * abs(meta["n0"])
    assert abs(n1 - meta["n1"]) <= 2e-6 * abs(meta["n1"]) + 1e-30
    assert not torch.isnan(out).any() and not torch.isinf(out).any()
    total, resid = spectral_residual(out, ref)
    if rep.branch == "slerp" and (case["cutoff"] > 0 or case["cull"] > 0):
        assert resid < 2e-5, f"beyond the threshold-tie bins the results differ by {resid:.2e}"
        assert total < 8.0 / math.sqrt(a.numel()), f"total {total:.2e}"
        assert abs(rep.cutoff_threshold - tr.cutoff_threshold) <= 1e-5 * tr.cutoff_threshold
        assert abs(rep.cull_threshold - tr.cull_threshold) <= 1e-5 * tr.cull_threshold
        assert abs(rep.n_slerp - tr.n_slerp) <= 4 + 1e-4 * tr.n_slerp
    else:
        assert total < 5e-6, f"{rep.branch}: {total:.2e}"
    return rep


def check_arith(engine, golden, case):
    a, b = gi.pair_input(case)
    out = engine.task_arithmetic_fft2(a, b, case["t"], agreement=case["agreement"]).cpu()
    ref = golden.get("g4_pair.safetensors", case["id"])
    assert out.shape == ref.shape
    # sign agreement is decided per bin on values that differ from the reference's by
    # FFT rounding: bins with |Re| ~ 1e-7 of the typical size can flip; they carry no energy
    assert so.rel_err(out, ref) < 5e-6


def models_in_window(case):
    tensors, models, cfg, lname = gi.layer_inputs(case)
    layer = int(lname.split(".")[2])
    use = [m for m in models
           if not (m.get("start_layer", 0) > layer or (m.get("end_layer", -1) != -1 and m.get("end_layer", -1) < layer))]
    return tensors, use, cfg, lname


# ---- K >= 3: what CAN be compared tightly -------------------------------------------------------
# From the second tournament round on, an input is an intermediate whose spectrum holds the bins
# the previous round culled.  In the reference those bins come back from its ifft -> fft round
# trip as +-1e-7 rounding noise, and the round takes its sign-agreement decisions, its 8 %
# quantile (which lands INSIDE the noise values: >= 16 % of the concatenated magnitudes are
# noise) and therefore its slerp-class membership on that noise (oracle/chaos_probe.py).  That
# part is not reproducible by anything but a bit-identical FFT.  Everything else is:
#   * the pairing, the branch, t (quirk Q4), the round's cull fraction        -> exact
#   * first-round thresholds / class counts / cosine                            -> as for K = 2
#   * later rounds: the cutoff threshold must be noise-level; the cull threshold, the class
#     count and the cosine move only by what the noise bins' membership moves them (measured
#     against the real reference: <= 3e-2, 3e-2, 2e-2) - a wrong cull fraction moves the cull
#     threshold by 2x, a wrong weight moves t
#   * the merged delta's spectrum OUTSIDE the bins that earlier rounds culled (and outside the
#     bins whose final cull decision differs because the cull threshold moved: those must sit
#     on the threshold and be few) agrees to ~1e-3 (K = 3: the slerp constants see the noise
#     bins' membership) / 1e-4 (K = 4 goldens); INSIDE, only the statistical floor holds.
LATER_ROUND_CULL_TOL = 5e-3        # one input is an intermediate
BOTH_INTER_CULL_TOL = 1.5e-1        # both are (K >= 4): the 10 % quantile lies among the ~b/||rel|| leftovers of noise bins
LATER_ROUND_NSLERP_TOL = 3e-2
LATER_ROUND_DOT_TOL = 2e-2
OUTSIDE_CULLED_TOL = 2.5e-3


def step_intermediate_inputs(tr, k):
    """Per pairing step: how many of its two inputs are intermediates (products of earlier
    pair merges).  Replays the reference's stack bookkeeping (fast_fourier.py:171-254)."""
    kinds = [0] * k                      # 0: raw delta, 1: intermediate
    out, nxt, pos = [], [], 0
    stack = list(kinds)
    for (x, y) in tr.pairs:
        if y < 0:
            out.append(stack[x])
            nxt.append(stack[x])
        else:
            out.append(stack[x] + stack[y])
            nxt.append(1)
        pos += 1
        if pos == (len(stack) + 1) // 2:             # round complete
            stack, nxt, pos = nxt, [], 0
    return out


def check_layer_steps(rep, tr, numel, reported_fields=True, biased_slerp_norms=False):
    """Every pairing step of the HIP path's report against the oracle's trace.
    reported_fields=False skips t and the cull fraction (values the library merely reports
    back) so that a test can show the MEASURED quantities alone catch a slip.
    biased_slerp_norms=True: `tr` is the AS-IS oracle at a size where torch's CPU norm is visibly
    biased (>= 16 M elements); norm_mode = reference_cpu reproduces its SPATIAL norms (the cutoff
    threshold then matches tightly) but not the norms of the gathered slerp-class vectors, so the
    cosine and what follows from it (the cull threshold) agree to the size of that bias only."""
    assert rep.branches == tr.branches
    assert [(s[0], s[1]) for s in rep.steps] == tr.pairs
    k = len(rep.delta_norms)
    n_inter = step_intermediate_inputs(tr, k)
    first_cut = None
    for i, (info, bt) in enumerate(zip(rep.infos, tr.steps)):
        if bt is None or tr.branches[i] != "slerp":
            continue
        if reported_fields:
            assert abs(info.t - bt.t) <= 1e-12, f"step {i}: t {info.t} vs {bt.t} (weights / quirk Q4)"
            assert info.cull_pct == bt.cull_pct, f"step {i}: cull fraction {info.cull_pct} vs {bt.cull_pct}"
        if n_inter[i] == 0:
            first_cut = bt.cutoff_threshold if first_cut is None else first_cut
            # neighbouring order statistics are ~1/(0.2 n) apart (relative): a rank shifted by one
            # through FFT rounding shows at small sizes
            gran = 8.0 / max(numel, 1)
            assert abs(info.cutoff_threshold - bt.cutoff_threshold) <= max(1e-5, gran) * bt.cutoff_threshold + 1e-30, f"step {i} cutoff"
            loose = 100.0 if biased_slerp_norms else 1.0
            assert abs(info.cull_threshold - bt.cull_threshold) <= loose * max(5e-5, gran) * bt.cull_threshold + 1e-30, f"step {i} cull"
            assert abs(info.n_slerp - bt.n_slerp) <= 4 + 1e-4 * bt.n_slerp, f"step {i} n_slerp"
            # (one bin in or out of the class - a rank shifted through FFT rounding, as above - moves the cosine by ~1/n)
            assert abs(info.dot - bt.dot) <= loose * max(1e-4, 1.0 / max(numel, 1)), f"step {i} dot"
        else:
            if first_cut:
                assert info.cutoff_threshold < 1e-3 * first_cut and bt.cutoff_threshold < 1e-3 * first_cut, \
                    f"step {i}: the cutoff threshold of a later round lies in the culled bins' noise"
            tol = LATER_ROUND_CULL_TOL if n_inter[i] == 1 else BOTH_INTER_CULL_TOL
            assert abs(info.cull_threshold - bt.cull_threshold) <= tol * bt.cull_threshold, \
                f"step {i} cull threshold {info.cull_threshold} vs {bt.cull_threshold}"
            both = 2.0 if n_inter[i] == 2 else 1.0          # two noisy inputs: twice the irreproducible membership
            assert abs(info.n_slerp - bt.n_slerp) <= both * LATER_ROUND_NSLERP_TOL * bt.n_slerp, \
                f"step {i} n_slerp {info.n_slerp} vs {bt.n_slerp}"
            assert abs(info.dot - bt.dot) <= both * LATER_ROUND_DOT_TOL, f"step {i} dot {info.dot} vs {bt.dot}"


def masked_spectral_check(delta, tr, tol_outside=OUTSIDE_CULLED_TOL, device="cpu"):
    """Spectrum of (delta - oracle's merged delta), split by the bins earlier rounds culled.
    Returns (relative error outside, relative error inside, number of final-cull flips).
    device: where this CHECK runs its three double-precision transforms (the metric's 235 M-element tensors take
    minutes and ~25 GB on the host; on the GPU box the checker may use the card - the product never does)."""
    ref = tr.merged_delta.to(device).double()
    hip = delta.to(device).double()
    d = hip - ref
    dims = tuple(range(ref.ndim))
    D, Rf, Hf = torch.fft.fftn(d), torch.fft.fftn(ref), torch.fft.fftn(hip)
    del hip, d
    slerp_steps = [i for i, b in enumerate(tr.steps) if b is not None and b.culled_mask is not None]
    assert len(slerp_steps) >= 2
    union = torch.zeros(ref.shape, dtype=torch.bool, device=device)
    for i in slerp_steps[:-1]:
        union |= tr.steps[i].culled_mask.reshape(ref.shape).to(device)
    mirror = lambda m: torch.roll(torch.flip(m, dims=dims), shifts=tuple([1] * len(dims)), dims=dims)
    union |= mirror(union)
    # final cull: a bin zeroed on one side only must sit on the (moved) threshold
    last = tr.steps[slerp_steps[-1]]
    scale = tr.target_norm
    thr = last.cull_threshold * scale
    zero_ref = Rf.real.abs() < 1e-3 * thr
    zero_hip = Hf.real.abs() < 1e-3 * thr
    flips = (zero_ref ^ zero_hip) & ~union
    mag = torch.where(zero_ref, Hf.real.abs(), Rf.real.abs())
    band = BOTH_INTER_CULL_TOL if len(slerp_steps) >= 3 else LATER_ROUND_CULL_TOL
    # ... except the handful of bins that sat ON an EARLIER round's cull threshold (SURVEY 8a's tie floor: a bin and
    # its conjugate twin, culled there on one side only): one side carries the value, the other rounding noise, and
    # the last round treats them accordingly.  Their number grows with the tensor (~5e-8 n bins within rounding of
    # a threshold: DESIGN 6.1).
    far = flips & ((mag - thr).abs() > 1.5 * band * thr)
    n_far = int(far.sum())
    assert n_far <= 2 * (4 + ref.numel() // (1 << 23)), \
        f"{n_far} bins culled on one side only do not sit on the cull threshold"
    n_flips = int(flips.sum())
    assert n_flips <= 16 + 2 * band * ref.numel() * last.cull_pct, f"{n_flips} final-cull flips"
    keep = ~union & ~flips & ~mirror(flips)
    e2 = (D.real ** 2 + D.imag ** 2)
    r2 = (Rf.real ** 2 + Rf.imag ** 2)
    # the same tie bins when the HIP side culled them earlier and the oracle did not (not in `union`): the few largest
    # bins of the difference are left out, as spectral_residual() does for a single pair merge
    e_keep = e2[keep]
    n_drop = min(2 * (4 + ref.numel() // (1 << 23)), max(e_keep.numel() - 1, 0))
    dropped = float(torch.topk(e_keep, n_drop).values.sum()) if n_drop > 0 else 0.0
    outside = math.sqrt(max(float(e_keep.sum()) - dropped, 0.0) / float(r2[keep].sum()))
    inside = math.sqrt(float(e2[union].sum()) / max(float(r2[union].sum()), 1e-300)) if bool(union.any()) else 0.0
    assert outside <= tol_outside, f"outside the culled bins the merged delta differs by {outside:.2e}"
    return outside, inside, n_flips


def check_layer(engine, golden, case, norm_mode=None):
    tensors, use, cfg, lname = models_in_window(case)
    fts = [tensors[m["model"]] for m in use]
    bases = [tensors[m["base"]] for m in use]
    alphas = [m["alpha"] for m in use]
    tr = so.LayerTrace()
    so.merge_layer(fts, bases, alphas, tensors[cfg["output_base_model"]], names=[m["model"] for m in use], trace=tr)
    out, rep, delta = engine.merge_layer(fts, bases, alphas, tensors[cfg["output_base_model"]], want_delta=True, layer_name=lname,
                                         norm_mode=norm_mode)       # None: the product's default (constants.DEFAULT_NORM_MODE)
    out, delta = out.cpu(), delta.cpu()
    ref = golden.get("g7_layer.safetensors", case["id"])
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape            # reference test_fast_fourier.py:320
    assert rep.branches == tr.branches
    assert [(s[0], s[1]) for s in rep.steps] == tr.pairs
    assert abs(rep.target_norm - tr.target_norm) <= 1e-5 * tr.target_norm + 1e-12
    floor = golden.manifest["layer_self_floor"][case["id"]]
    total, resid = spectral_residual(out.float(), ref.float())
    n_slerp_rounds = sum(1 for b in tr.branches if b == "slerp")
    if n_slerp_rounds == 0:
        mism = (out.view(torch.int16) != ref.view(torch.int16)).float().mean().item()
        assert mism < 2e-3 and total < 1e-4, f"{mism:.2e} of bf16 outputs differ, {total:.2e}"
        assert so.rel_err(delta, tr.merged_delta) < 5e-6 or tr.merged_delta.norm() == 0
    elif n_slerp_rounds == 1:
        # one slerp merge: only the threshold-tie bins may differ
        d_total, d_resid = spectral_residual(delta, tr.merged_delta)
        assert d_resid < 2e-5, f"merged delta differs by {d_resid:.2e} beyond the tie bins"
        assert total <= 2.5 * floor + 1e-4, f"{total:.2e} vs reference self-floor {floor:.2e}"
    else:
        # later rounds decide on rounding noise in the reference (module docstring): the total
        # is bounded by the reference's own floor, everything reproducible is compared tightly
        assert total <= 2.5 * floor + 1e-4, f"{total:.2e} vs reference self-floor {floor:.2e}"
        masked_spectral_check(delta, tr)
    check_layer_steps(rep, tr, out.numel())
    return rep


# ---- K >= 3: the reference's own floor as reference-held DATA (tests/golden/g11_floor.safetensors) ----------------
FLOOR_RATIO = 1.25                  # the HIP path may sit this multiple of d(reference with fp64 FFTs, reference) away
FLOOR_RATIO_EXACT = 1.6             # norm_mode = exact against the reference's device="cpu" output: plus its norm artefact
                                    # (measured 1.25 / 1.54 on the output / merged delta at 1024^2, K = 3)
FLOOR_DELTA_STORE = 4e-4            # the fixtures' merged deltas are fp16(delta * 256)


def floor_fixture(golden, case):
    cid = case["id"]
    g = lambda key: golden.get("g11_floor.safetensors", f"{cid}/{key}")
    return {"out": g("out").float(), "out64": g("out_fp64").float(),
            "delta": g("delta_f16").float() / gi.FLOOR_DELTA_SCALE, "delta64": g("delta_fp64_f16").float() / gi.FLOOR_DELTA_SCALE}


def check_floor(engine, golden, case, norm_mode=None):
    """K = 3 / K = 4 at 1024 x 1024 against the REFERENCE's outputs (run here by oracle/gen_golden.py): the distance of
    the HIP path from the reference must not exceed FLOOR_RATIO times the distance of the reference from ITSELF when
    its FFTs are evaluated in float64 - on the bf16 output and on the fp32 merged delta.  Both distances come from
    committed data, none from a number quoted in prose."""
    tensors, use, cfg, lname = models_in_window(case)
    fts = [tensors[m["model"]] for m in use]
    bases = [tensors[m["base"]] for m in use]
    alphas = [m["alpha"] for m in use]
    fx = floor_fixture(golden, case)
    out, rep, delta = engine.merge_layer(fts, bases, alphas, tensors[cfg["output_base_model"]], want_delta=True, layer_name=lname,
                                         norm_mode=norm_mode)
    out, delta = out.cpu().float(), delta.cpu()
    floor_out, floor_delta = so.rel_err(fx["out64"], fx["out"]), so.rel_err(fx["delta64"], fx["delta"])
    rec = {"id": case["id"], "floor_out": floor_out, "floor_delta": floor_delta,
           "out_vs_ref": so.rel_err(out, fx["out"]), "out_vs_ref_fp64": so.rel_err(out, fx["out64"]),
           "delta_vs_ref": so.rel_err(delta, fx["delta"]), "delta_vs_ref_fp64": so.rel_err(delta, fx["delta64"])}
    assert abs(floor_out - golden.manifest["floor_meta"][case["id"]]["out_floor"]) < 1e-6
    ratio = FLOOR_RATIO_EXACT if norm_mode == "exact" else FLOOR_RATIO
    assert rec["out_vs_ref"] <= ratio * floor_out, rec
    assert rec["out_vs_ref_fp64"] <= ratio * floor_out, rec
    assert rec["delta_vs_ref"] <= ratio * floor_delta + FLOOR_DELTA_STORE, rec
    assert rec["delta_vs_ref_fp64"] <= (ratio + 0.05) * floor_delta + FLOOR_DELTA_STORE, rec
    return rec


def check_noise_seed_sensitivity(engine, golden, case):
    """Two realisations of the rounding-noise model (debug option "noise_seed") give outputs no further apart than the
    reference is from itself: the model's seed is not a tuning knob that happens to land on the reference."""
    tensors, use, cfg, lname = models_in_window(case)
    fts = [tensors[m["model"]] for m in use]
    bases = [tensors[m["base"]] for m in use]
    alphas = [m["alpha"] for m in use]
    fx = floor_fixture(golden, case)
    floor_out = so.rel_err(fx["out64"], fx["out"])
    outs = []
    try:
        for seed in (0, 977, 40503):
            engine.ctx.debug_option("noise_seed", seed)
            out, rep = engine.merge_layer(fts, bases, alphas, tensors[cfg["output_base_model"]], layer_name=lname)
            outs.append(out.cpu().float())
    finally:
        engine.ctx.debug_option("noise_seed", 0)
    d01, d02 = so.rel_err(outs[1], outs[0]), so.rel_err(outs[2], outs[0])
    assert d01 > 0 and d02 > 0, "the seed does not reach the noise model"
    assert max(d01, d02) <= FLOOR_RATIO * floor_out, (d01, d02, floor_out)
    for o in outs[1:]:
        assert so.rel_err(o, fx["out"]) <= FLOOR_RATIO * floor_out
    return d01, d02, floor_out


# ---- the reference's legacy in-RAM operator (shard/merge/fourier.py) ------------------------------------------------
def check_legacy(engine, golden, case):
    """`merge_options.operator: fourier_legacy` (shardmerge_amd/merge/fourier_legacy.py) against the outputs of the
    reference's own class (G12, oracle/gen_golden.py --only-legacy): median target norm, cosine pairing, deltas and
    first-round norms in the models' dtype, the task_add_models post-pass, fp32 result."""
    import asyncio
    from shardmerge_amd.config import MergeConfig, MergeModel
    from shardmerge_amd.merge.fourier_legacy import LegacyFourierMerge
    from shardmerge_amd.writer import ShardLayer
    tensors, models, cfg_kw, lname = gi.layer_inputs(case)
    cfg = MergeConfig(finetune_merge=[MergeModel(**m) for m in models], output_base_model=cfg_kw["output_base_model"], output_dir="unused")
    op = LegacyFourierMerge(cfg, task_add_models=case.get("task_add"), engine=engine)
    dev = engine.device

    async def fetch(uri, name, device):
        return tensors[uri].to(dev)
    op._fetch = fetch
    out = asyncio.run(op._merge_layer(ShardLayer(1, "s", lname, False), str(dev))).cpu()
    ref = golden.get("g12_legacy.safetensors", case["id"])
    assert out.dtype == torch.float32 and out.shape == ref.shape              # (fourier.py:205: no bf16 cast)
    floor = golden.manifest["legacy_self_floor"][case["id"]]
    total, resid = spectral_residual(out, ref)
    if floor < 1e-3:                      # one slerp round (or none) decides nothing on rounding noise
        assert resid < 2e-5 and total < 1e-3, (total, resid)
    else:
        assert total <= 2.5 * floor + 1e-4, (total, floor)
    return total, resid, floor


# ---- A1 / A8 at function level: smhip_slerp, smhip_exact_norm / smhip_reference_cpu_norm + smhip_div_scalar ----------
def check_fn_slerp(engine, golden, case):
    """functions.py:24-43 against the reference's own outputs (g3_slerp) and, on a 2-D tensor, against the oracle
    (the relative vector is normalised along the LAST dimension there)"""
    a, b = gi.pair_input(case)
    out = engine.slerp(a, b, case["t"]).cpu()
    assert so.rel_err(out, golden.get("g3_slerp.safetensors", case["id"])) < 2e-6
    g = torch.Generator().manual_seed(case["seed"] + 1000)
    x, y = torch.randn(37, 129, generator=g), torch.randn(37, 129, generator=g)
    y = 0.6 * x + 0.8 * y
    assert so.rel_err(engine.slerp(x, y, 0.35).cpu(), so.slerp(x, y, 0.35)) < 2e-6
    e = torch.zeros(0)
    assert engine.slerp(e, e, 0.5).numel() == 0


def check_fn_normalize(engine):
    """functions.py:75-88 as the reference's CPU run computes it: norm = tensor.norm().item() (a 16-bit tensor's norm is
    a 16-bit value), tensor / norm in the tensor's dtype - bit for bit in reference_cpu mode; exact mode: the fp64 norm"""
    g = torch.Generator().manual_seed(77)
    for dtype, n in ((torch.float32, 8 * 1237), (torch.bfloat16, 8 * 4099), (torch.float16, 8 * 515)):
        x = (torch.randn(n, generator=g) * 0.02).to(dtype)
        out, norm = engine.normalize_tensor(x, norm_mode="reference_cpu")
        ref_norm = x.norm().item()
        assert norm == ref_norm, (dtype, norm, ref_norm)
        assert torch.equal(out.cpu(), x / ref_norm), dtype
        out_e, norm_e = engine.normalize_tensor(x, norm_mode="exact")
        exact = x.double().pow(2).sum().sqrt().item()
        assert abs(norm_e - exact) <= 1e-12 * exact
        assert torch.equal(out_e.cpu(), x / exact)
    z = torch.zeros(16)
    out, norm = engine.normalize_tensor(z)
    assert norm == 0.0 and torch.equal(out.cpu(), z)
