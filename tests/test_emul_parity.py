"""CPU tier: the product's kernel bodies + host orchestration, executed by the
work-group emulator (tests/emul), against the reference's golden vectors and
the oracle.  No GPU needed; sizes are the golden fixtures' (<= 448 x 256)."""
import pytest
import torch

from tests import parity_checks as pc
from tests.golden import inputs as gi


@pytest.fixture(scope="module")
def engine():
    from tests.emul.loader import emul_engine
    return emul_engine()


@pytest.mark.parametrize("case", gi.FFT_CASES, ids=lambda c: c["id"])
def test_fft(engine, golden, case):
    pc.check_fft(engine, golden, case)


@pytest.mark.parametrize("case", gi.INTERP_CASES, ids=lambda c: c["id"])
def test_interpolate(engine, golden, case):
    pc.check_interp(engine, golden, case)


@pytest.mark.parametrize("case", gi.SLERP_CASES, ids=lambda c: c["id"])
def test_function_level_slerp(engine, golden, case):
    pc.check_fn_slerp(engine, golden, case)


def test_function_level_normalize_tensor(engine):
    pc.check_fn_normalize(engine)


@pytest.mark.parametrize("case", gi.PAIR_CASES, ids=lambda c: c["id"])
def test_pair_slerp(engine, golden, case):
    pc.check_pair(engine, golden, case)


@pytest.mark.parametrize("case", gi.ARITH_CASES, ids=lambda c: c["id"])
def test_pair_arith(engine, golden, case):
    pc.check_arith(engine, golden, case)


@pytest.mark.parametrize("norm_mode", ["reference_cpu", "exact"])
@pytest.mark.parametrize("case", gi.LAYER_CASES, ids=lambda c: c["id"])
def test_layer(engine, golden, case, norm_mode):
    """the golden tier in BOTH norm modes (the goldens are the reference's device=cpu outputs at sizes where torch's
    norm is still accurate to 1e-6, so both must meet them)"""
    pc.check_layer(engine, golden, case, norm_mode=norm_mode)


def test_unsupported_length_is_loud(engine):
    x = torch.randn(17 * 19, 4)          # 323 = 17*19: no radix for it (as a ROW length the chirp-z passes would take it)
    with pytest.raises(NotImplementedError):
        engine.fft_transform(x)
    assert engine.lib.length_supported(14336) and engine.lib.length_supported(28672)
    assert not engine.lib.length_supported(17) and not engine.lib.length_supported(65536)


def test_inf_raises_like_reference(engine):
    base = torch.zeros(8, 8, dtype=torch.bfloat16)
    ft = base.clone()
    ft[0, 0] = float("inf")
    with pytest.raises(ValueError, match="Inf in merged tensor for model.layers.3.w"):
        engine.merge_layer([ft], [base], [1.0], base, layer_name="model.layers.3.w")
    a = torch.randn(8, 8)
    b = torch.randn(8, 8)
    a[1, 1] = 3e38
    b[1, 1] = -3e38
    # the overflow in the transform gives Inf/NaN; the reference zeroes NaN and raises on Inf
    try:
        engine.merge_tensors_fft2_slerp(a, b, 0.5)
    except ValueError as e:
        assert "Inf in ifft output" in str(e)


def test_nan_is_zeroed(engine):
    base = torch.randn(16, 16).to(torch.bfloat16)
    ft = (base.float() + 0.01 * torch.randn(16, 16)).to(torch.bfloat16)
    ft2 = ft.clone()
    ft2[2, 3] = float("nan")
    out, rep = engine.merge_layer([ft2], [base], [1.0], base)
    assert rep.nan_final == 1 and out[2, 3] == 0 and not torch.isnan(out.float()).any()


def test_candidate_list_overflow_falls_back(engine, golden):
    """The level-2 selection pass compacts ~1 % of the data into candidate lists; when
    a list overflows, the plain full passes must produce the very same result."""
    case = [c for c in gi.PAIR_CASES if c["id"] == "pair_256_mix"][0]
    a, b = gi.pair_input(case)
    kw = dict(b=case["b"], t_sum=case["t_sum"], cutoff_pct=case["cutoff"], cull_pct=case["cull"])
    ref, _, _, rep0 = engine.merge_tensors_fft2_slerp(a, b, case["t"], **kw)
    engine.ctx.debug_option("cand_cap", 7)
    engine.ctx.profile(True)
    engine.ctx.profile_reset()
    try:
        out, _, _, rep1 = engine.merge_tensors_fft2_slerp(a, b, case["t"], **kw)
        launches = engine.ctx.profile_table()
    finally:
        engine.ctx.debug_option("cand_cap", 0)
        engine.ctx.profile(False)
    # the call was redone with the plain histogram passes (level 2 and 3 of both thresholds)
    assert launches["select_hist"][0] == 4 and launches["f1_rows_fwd"][0] == 2
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())   # sums associate differently: ulps
    assert rep1.cutoff_threshold == rep0.cutoff_threshold and rep1.cull_threshold == rep0.cull_threshold
    assert rep1.n_slerp == rep0.n_slerp and rep1.s01 == pytest.approx(rep0.s01, rel=1e-9)


def test_selection_rounds_and_midstream_flush(engine, golden):
    """The level-2 selection pass streams in rounds and flushes its staged candidates to the
    global lists between rounds when the stage fills up (only 100 M-element tensors get there
    by themselves): force many rounds and a flush after each, the result must not change."""
    case = [c for c in gi.PAIR_CASES if c["id"] == "pair_256_mix"][0]
    a, b = gi.pair_input(case)
    kw = dict(b=case["b"], t_sum=case["t_sum"], cutoff_pct=case["cutoff"], cull_pct=case["cull"])
    ref, _, _, rep0 = engine.merge_tensors_fft2_slerp(a, b, case["t"], **kw)
    engine.ctx.debug_option("sel_chunks", 64)           # one work-group, 4 rounds of 16 steps
    engine.ctx.debug_option("sel_flush_always", 1)
    engine.ctx.profile(True)
    engine.ctx.profile_reset()
    try:
        out, _, _, rep1 = engine.merge_tensors_fft2_slerp(a, b, case["t"], **kw)
        launches = engine.ctx.profile_table()
    finally:
        engine.ctx.debug_option("sel_chunks", 0)
        engine.ctx.debug_option("sel_flush_always", 0)
        engine.ctx.profile(False)
    assert "select_hist" not in launches                                   # still the fast path
    assert rep1.cutoff_threshold == rep0.cutoff_threshold and rep1.cull_threshold == rep0.cull_threshold
    assert rep1.n_slerp == rep0.n_slerp and rep1.s01 == pytest.approx(rep0.s01, rel=1e-9)
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


def test_heavy_ties(engine):
    """Quantised inputs give spectra with many exactly equal magnitudes (and exact
    zeros): the order statistics must still be the reference's."""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-2, 3, (64, 64), generator=g).float() * 0.25
    b = torch.randint(-2, 3, (64, 64), generator=g).float() * 0.25
    a[:, ::2] = 0
    tr = so.BlendTrace()
    ref, n0, n1 = so.merge_tensors_fft2_slerp(a, b, 0.5, cutoff_pct=0.08, cull_pct=0.2, trace=tr)
    out, m0, m1, rep = engine.merge_tensors_fft2_slerp(a, b, 0.5, cutoff_pct=0.08, cull_pct=0.2)
    total, resid = pc.spectral_residual(out, ref, drop=64)
    assert abs(rep.cutoff_threshold - tr.cutoff_threshold) <= 1e-5 * max(tr.cutoff_threshold, 1e-6)
    assert resid < 1e-4


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384, 14336, 28672, 7168, 3072, 5120, 6144, 3584, 12288, 13824, 27648, 2304, 1536, 2560])
def test_static_and_dynamic_plan_lengths(engine, n):
    """Every straight-line (static plan) length and a few run-time planned ones, as a
    row transform (2 x n) and as a column transform (n x 2)."""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(n)
    for shape in [(2, n), (n, 2)]:
        x = torch.randn(*shape, generator=g)
        f = engine.fft_transform(x)
        assert so.rel_err(torch.view_as_real(f), torch.view_as_real(so.fft_transform(x))) < TOL_FFT
        assert so.rel_err(engine.ifft_transform(f), x) < TOL_FFT


TOL_FFT = 3e-6


def test_complex_packed_engine_mode(golden):
    """fft_engine.hpp's PACK mode 2 (one vf2 = re, im of one value; off by default - measured on MI355X: the same
    kernel times, profiles/r04_ab_complex_packed_butterflies.txt) is the same transform: every radix through the row,
    column and inverse kernels of an alternate emulator build, and a golden layer through the whole pipeline.  (The
    device spelling of its three swizzled ops is checked on the GPU by tools/cx_ops_check.hip.)"""
    from oracle import spectral_oracle as so
    from tests.emul.loader import emul_engine_variant
    eng = emul_engine_variant(["-DSM_ROW_PACK_BIG=2", "-DSM_I1_PACK=2", "-DSM_F1Q_PACK=2", "-DSM_F2_PACK=2", "-DSM_F2S_PACK=2",
                               "-DSM_ROW_PACK_MAX_T=0"], "cpx")
    for n in (8192, 14336, 7168, 3072, 5120, 13824, 2304, 1408):      # radices 32 16 8 7 4 3 5; 1408 = 11 * 128 at run time
        g = torch.Generator().manual_seed(n)
        for shape in [(2, n), (n, 2)]:
            x = torch.randn(*shape, generator=g)
            f = eng.fft_transform(x)
            assert so.rel_err(torch.view_as_real(f), torch.view_as_real(so.fft_transform(x))) < TOL_FFT
            assert so.rel_err(eng.ifft_transform(f), x) < TOL_FFT
    pc.check_layer(eng, golden, gi.LAYER_CASES[0])


def test_misaligned_inputs_take_the_elementwise_path(engine):
    """Static-plan lengths assume 16-byte aligned rows; an unaligned view must fall back
    to the run-time kernel and give the same answer."""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(9)
    big = (torch.randn(3 + 64 * 1024, generator=g) * 0.01).to(torch.bfloat16)
    base = big[3:3 + 32 * 1024].view(32, 1024)              # 6-byte offset: not 16-byte aligned
    ft0 = (base.float() + torch.randn(32, 1024, generator=g) * 0.002).to(torch.bfloat16)
    ft1 = (base.float() + torch.randn(32, 1024, generator=g) * 0.003).to(torch.bfloat16)
    out_u, rep_u = engine.merge_layer([ft0, ft1], [base, base], [0.3, 0.5], base)
    out_a, rep_a = engine.merge_layer([ft0, ft1], [base.clone(), base.clone()], [0.3, 0.5], base.clone())
    assert rep_u.branches == rep_a.branches == ["slerp"]
    assert torch.equal(out_u, out_a)


def test_non_finite_delta_norm_is_an_error_not_a_hang(engine):
    """K >= 2 with a NaN (or Inf) element: no pair can ever be formed and the reference's
    tournament loop spins forever (fast_fourier.py:171-254); the library returns an error."""
    g = torch.Generator().manual_seed(123)
    base = torch.randn(16, 16, generator=g).to(torch.bfloat16)
    fts = [(base.float() + 0.01 * torch.randn(16, 16, generator=g)).to(torch.bfloat16) for _ in range(3)]
    keep = fts[1][2, 3].clone()
    fts[1][2, 3] = float("nan")
    for k in (2, 3):
        with pytest.raises(ValueError, match="Non-finite delta norm in model.layers.0.w"):
            engine.merge_layer(fts[:k], [base] * k, [0.3, 0.5, 0.2][:k], base, layer_name="model.layers.0.w")
    fts[1][2, 3] = float("inf")
    with pytest.raises(ValueError, match="Non-finite delta norm"):
        engine.merge_layer(fts[:2], [base] * 2, [0.3, 0.5], base)
    # the context stays usable
    fts[1][2, 3] = keep
    out, rep = engine.merge_layer(fts[:2], [base] * 2, [0.3, 0.5], base)
    assert rep.branches == ["slerp"] and not torch.isnan(out.float()).any()


def test_mixed_input_dtypes_are_promoted_not_demoted(engine):
    """fp32 base next to bf16 finetunes (and mixed bf16/fp32 finetunes): the reference upcasts
    everything to fp32 (base.py:128-131); demoting the base to bf16 changed 93 % of the outputs."""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(11)
    base32 = torch.randn(64, 128, generator=g) * 0.02                     # NOT bf16-representable
    fts = [(base32 + torch.randn(64, 128, generator=g) * s_).to(torch.bfloat16) for s_ in (0.002, 0.003)]
    for f_list in (fts, [fts[0], fts[1].float()]):
        ref = so.merge_layer(f_list, [base32, base32], so.ALPHAS[:2], base32)
        out, rep = engine.merge_layer(f_list, [base32, base32], so.ALPHAS[:2], base32)
        assert rep.branches == ["slerp"]
        out = out.cpu()
        total, resid = pc.spectral_residual(out.float(), ref.float())
        assert resid < 1e-4 and (out.view(torch.int16) != ref.view(torch.int16)).float().mean().item() < 0.02


# ---- K >= 3: the checks are sharp enough to tell a slip from the reference's own chaos ----------
def _layer_run(engine, case, mutation=None):
    from oracle import spectral_oracle as so
    tensors, use, cfg, lname = pc.models_in_window(case)
    fts = [tensors[m["model"]] for m in use]
    bases = [tensors[m["base"]] for m in use]
    alphas = [m["alpha"] for m in use]
    tr = so.LayerTrace()
    so.merge_layer(fts, bases, alphas, tensors[cfg["output_base_model"]], trace=tr, _mutation=mutation)
    out, rep, delta = engine.merge_layer(fts, bases, alphas, tensors[cfg["output_base_model"]], want_delta=True)
    return rep, tr, delta


@pytest.mark.parametrize("cid", ["layer_k3", "layer_k3_swap", "layer_k4"])
def test_k3_steps_and_unculled_spectrum_match_the_oracle(engine, cid):
    case = [c for c in gi.LAYER_CASES if c["id"] == cid][0]
    rep, tr, delta = _layer_run(engine, case)
    pc.check_layer_steps(rep, tr, delta.numel())
    outside, inside, flips = pc.masked_spectral_check(delta, tr)
    assert inside > 3 * outside            # the irreproducible part really is confined to the culled bins


@pytest.mark.parametrize("cid,mutation", [("layer_k3", "keep_cull_pct"), ("layer_k4", "keep_cull_pct"),
                                          ("layer_k3", "sum_weights"),     # (K = 4: both weights double, t is unchanged)
                                          ("layer_k3_swap", "swap_weights"), ("layer_k4", "swap_weights")])
def test_k3_checks_catch_a_wrong_round2(engine, cid, mutation):
    """Against an oracle with a deliberate slip (cull fraction not halved, merged weight not
    averaged, weights swapped with a/b - quirk Q4) the same checks must FAIL, on measured
    quantities alone (the library's self-reported t / cull fraction are not consulted):
    the old bound - 2.5 x the reference's chaos floor on the output - let all of these pass."""
    case = [c for c in gi.LAYER_CASES if c["id"] == cid][0]
    rep, tr_bad, delta = _layer_run(engine, case, mutation)
    caught = 0
    for check in (lambda: pc.check_layer_steps(rep, tr_bad, delta.numel(), reported_fields=False),
                  lambda: pc.masked_spectral_check(delta, tr_bad)):
        try:
            check()
        except AssertionError:
            caught += 1
    assert caught >= 1, f"{mutation} on {cid} went unnoticed"
    with pytest.raises(AssertionError):
        pc.check_layer_steps(rep, tr_bad, delta.numel())        # and the reported fields say so directly


def test_nan_inf_policy_in_the_inverse_row_pass(engine):
    """K = 2: the add-back happens inside the inverse row pass (its group-of-8 NaN/Inf screen
    and the packed bf16 conversion).  A NaN in output_base_model's tensor is zeroed and counted
    (fast_fourier.py:270-271), an Inf raises the reference's error (:273-274)."""
    g = torch.Generator().manual_seed(21)
    base = (torch.randn(64, 256, generator=g) * 0.02).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(64, 256, generator=g) * s_).to(torch.bfloat16) for s_ in (0.002, 0.003)]
    clean, rep0 = engine.merge_layer(fts, [base, base], [0.3, 0.5], base)
    bo = base.clone()
    bo[5, 17] = float("nan")
    bo[63, 255] = float("nan")
    out, rep = engine.merge_layer(fts, [base, base], [0.3, 0.5], bo)
    out, clean = out.cpu(), clean.cpu()
    assert rep.nan_final == 2 and rep.nan_ifft == 0
    assert out[5, 17] == 0 and out[63, 255] == 0 and not torch.isnan(out.float()).any()
    mask = torch.ones(64, 256, dtype=torch.bool)
    mask[5, 17] = mask[63, 255] = False
    assert torch.equal(out[mask], clean[mask])                     # nothing else moved
    bo = base.clone()
    bo[7, 3] = float("-inf")
    with pytest.raises(ValueError, match="Inf in merged tensor for model.layers.1.mlp"):
        engine.merge_layer(fts, [base, base], [0.3, 0.5], bo, layer_name="model.layers.1.mlp")
    out2, _ = engine.merge_layer(fts, [base, base], [0.3, 0.5], base)     # the context stays usable
    assert torch.equal(out2.cpu(), clean)


@pytest.mark.parametrize("cid", ["layer_k3", "layer_k4"])
def test_spectral_intermediates_are_used_and_agree_with_the_materialised_path(engine, golden, cid):
    """K >= 3: an intermediate stays in the spectral domain (no inverse + forward transform between
    rounds; the rounding noise the reference's round trip leaves in the culled bins is modelled).
    Both paths must sit within the reference's own reproducibility floor of each other, and the
    spectral one must really skip the transforms."""
    case = [c for c in gi.LAYER_CASES if c["id"] == cid][0]
    tensors, use, cfg, lname = pc.models_in_window(case)
    fts = [tensors[m["model"]] for m in use]
    bases = [tensors[m["base"]] for m in use]
    alphas = [m["alpha"] for m in use]
    bo = tensors[cfg["output_base_model"]]
    k = len(fts)

    def run(spectral):
        engine.ctx.debug_option("spectral_intermediates", 1 if spectral else 0)
        engine.ctx.profile(True)
        engine.ctx.profile_reset()
        try:
            out, rep, delta = engine.merge_layer(fts, bases, alphas, bo, want_delta=True)
            return out.cpu(), rep, delta.cpu(), engine.ctx.profile_table()
        finally:
            engine.ctx.profile(False)
            engine.ctx.debug_option("spectral_intermediates", 1)

    out_s, rep_s, delta_s, tab_s = run(True)
    out_m, rep_m, delta_m, tab_m = run(False)
    n_pairs = k - 1
    assert tab_m["i2_rows_inv"][0] == n_pairs and "spec_rescale" not in tab_m
    assert tab_s["i2_rows_inv"][0] == 1 and tab_s["i1_cols_inv"][0] == 1          # only the final inverse
    # its Parseval norm comes out of the cull selection pass and the role-a column pass: no pass of its own
    assert "spec_norm" not in tab_s and tab_s["spec_norm_sum"][0] == n_pairs - 1 and tab_s["spec_rescale"][0] == n_pairs - 1
    # every raw delta's rows are transformed once (all of them in ONE launch; the norms come with it: no delta_norms
    # pass), and its column pass runs when the pairing has placed it
    # (the two column passes of a pair of raw deltas share a launch: one pair at K = 3, two at K = 4)
    assert tab_s["f1_rows_fwd"][0] == 1 and tab_s["f2s_cols_fwd1"][0] == k - k // 2
    assert "f2_cols_fwd" not in tab_s and "delta_norms" not in tab_s
    assert rep_s.branches == rep_m.branches
    # the fused norm equals the separate Parseval pass (same sums, other association)
    engine.ctx.debug_option("fuse_spec_norm", 0)
    try:
        out_u, rep_u, delta_u = engine.merge_layer(fts, bases, alphas, bo, want_delta=True)
    finally:
        engine.ctx.debug_option("fuse_spec_norm", 1)
    assert float((delta_u.cpu() - delta_s).abs().max()) <= 2e-6 * float(delta_s.abs().max())
    floor = golden.manifest["layer_self_floor"][cid]
    from oracle import spectral_oracle as so
    assert so.rel_err(out_s.float(), out_m.float()) <= 2.0 * floor


@pytest.mark.parametrize("k", [2, 3])
def test_folded_column_pass(engine, k):
    """Long columns (14336 / 28672 rows of the Llama-3 MLP tensors) run with the first radix-4
    step of the COLUMN transform folded into the row pass (k_f1q) and R/4-point column
    transforms on 4 x Cb virtual columns; the spectrum planes are then permuted inside a bin
    column and the inverse column pass reads that order back.  Forced here at 8192 x 1024
    (the smallest shape the path accepts) and compared with the oracle as any other layer,
    K = 3 through the row-pair / single-signal variants."""
    from oracle import spectral_oracle as so
    rows, cols = 8192, 1024
    base, fts = so.synthetic_layer(rows, cols, k, seed=77)
    trx = so.LayerTrace()
    with so.exact_norms():
        refx = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=trx)
    engine.ctx.debug_option("fold_min_rows", 8192)
    engine.ctx.profile(True)
    engine.ctx.profile_reset()
    try:
        out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True, norm_mode="exact")
        engine.ctx.debug_option("fold_columns", 0)
        out0, rep0, delta0 = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True, norm_mode="exact")
    finally:
        engine.ctx.debug_option("fold_columns", 1)
        engine.ctx.debug_option("fold_min_rows", 0)
        engine.ctx.profile(False)
    pc.check_layer_steps(rep, trx, out.numel())
    if k == 2:
        d_total, d_resid = pc.spectral_residual(delta, trx.merged_delta)
        assert d_resid < 2e-5 and d_total < 8.0 / (rows * cols) ** 0.5
        assert pc.spectral_residual(delta, delta0)[1] < 2e-6            # folded vs plain: the same up to tie bins
    else:
        pc.masked_spectral_check(delta, trx, tol_outside=1e-3)
    assert so.rel_err(out.float(), refx.float()) < (1e-3 if k == 2 else 5e-3)


# ---- N3: AdditionMerge / TaskAdditionMerge (bit-exact: elementwise work in the tensors' dtype) ----
@pytest.mark.parametrize("case", gi.ADDITION_CASES, ids=lambda c: c["id"])
def test_addition_operators_bit_exact(engine, golden, case):
    base, fts = gi.addition_inputs(case)
    for tag, agree in (("addition", False), ("task_addition", True)):
        out = engine.addition_merge(fts, base, sign_agreement=agree).cpu()
        ref = golden.get("g9_addition.safetensors", f"{case['id']}.{tag}")
        assert out.dtype == ref.dtype and out.shape == ref.shape
        assert torch.equal(out, ref), f"{tag}: {(out != ref).sum().item()} of {out.numel()} elements differ"


def test_addition_known_answers_of_the_reference_tests(engine):
    """The reference's own known-answer cases (tests/merge/test_addition.py:92,139,186,
    tests/merge/test_taskaddition.py:45-93 and its disagreement case)."""
    one = torch.ones(4, 4)
    assert torch.allclose(engine.addition_merge([one * 2, one * 3], one).cpu(), one * 3.0)
    assert torch.allclose(engine.addition_merge([one, one], one).cpu(), torch.zeros(4, 4))
    assert torch.allclose(engine.addition_merge([one * 0, one * -1], one).cpu(), one * -3.0)
    assert torch.allclose(engine.addition_merge([one * 4], one).cpu(), one * 3.0)                      # single model, alpha unused
    assert torch.allclose(engine.addition_merge([one * 2, one * 3], one, sign_agreement=True).cpu(), one * 3.0)
    # signs disagree: sum of signs is 0 -> nothing matches sign 0 unless a delta is itself 0
    assert torch.allclose(engine.addition_merge([one * 2, one * 0], one, sign_agreement=True).cpu(), torch.zeros(4, 4))
    # two against one: the majority sign's deltas survive
    assert torch.allclose(engine.addition_merge([one * 2, one * 3, one * 0], one, sign_agreement=True).cpu(), one * 3.0)


def test_linear_blend_branch_with_a_larger_b(engine):
    """merge_options.b (N4): merge_tensors_fft2_slerp blends linearly, R = Fa + t Fb, when the norm
    ratio is below b (functions.py:196-202).  Unreachable at the reference's b = 0.1 (the layer
    routes ratios < 0.1 to Arithmetic-FFT first); with b = 0.5 a ratio of ~0.25 takes it."""
    from oracle import spectral_oracle as so
    base, fts = so.synthetic_layer(128, 256, 2, seed=61, sigmas=(0.004, 0.001))
    tr = so.LayerTrace()
    ref = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=tr, ratio_b=0.5)
    out, rep, delta = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True, b=0.5)
    assert rep.branches == tr.branches == ["linear"]
    assert so.rel_err(delta.cpu(), tr.merged_delta) < 5e-6
    assert (out.cpu().view(torch.int16) != ref.view(torch.int16)).float().mean().item() < 2e-3
    # and the default keeps the SLERP branch
    out2, rep2 = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base)
    assert rep2.branches == ["slerp"]


def test_norm_mode_reference_cpu_reproduces_torch_norm(engine):
    """merge_options.norm_mode = reference_cpu: every spatial norm is what torch.norm returns on CPU
    (8 fp32 lanes accumulated serially - biased for large tensors), so the layer follows the
    reference's device="cpu" decisions: target_norm and the delta norms equal the AS-IS oracle's,
    bit for bit, and the outputs agree beyond the tie bins."""
    from oracle import spectral_oracle as so
    for k, shape in ((2, (512, 1024)), (3, (256, 512))):
        base, fts = so.synthetic_layer(shape[0], shape[1], k, seed=9100 + k)
        tr = so.LayerTrace()
        ref = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=tr)        # as the reference is
        out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True, norm_mode="reference_cpu")
        torch_norms = [float((f.float() - base.float()).norm()) for f in fts]
        assert rep.delta_norms == torch_norms                       # exactly torch's values
        assert rep.target_norm == tr.target_norm
        exact = [float((f.float() - base.float()).double().norm()) for f in fts]
        assert rep.delta_norms != exact
        pc.check_layer_steps(rep, tr, out.numel())
        if k == 2:
            assert pc.spectral_residual(delta.cpu(), tr.merged_delta)[1] < 2e-5
    with pytest.raises(ValueError):
        engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, norm_mode="fast")


def test_reference_cpu_norm_is_torch_norm_bit_for_bit(engine):
    """smhip_reference_cpu_norm (what norm_mode = reference_cpu takes every spatial norm with) against torch.norm on
    the host, bit for bit: ATen's kernel is fma(x, x, acc) in 8 serial fp32 lanes, lanes added in order, the n % 8
    tail (first 4 * (T / 4) elements as product + add, the rest as fma), sqrt - reproduced by a PARALLEL algorithm
    (csrc/sm_aten_norm.hpp).  Sizes cover the one-chunk path, several chunks, the summaries' window; inputs cover
    bf16 / f16 deltas (lattice values: ties everywhere), general fp32 (where a rounded product would differ from
    the fma), every tail length, and data that defeats the binade predictions."""
    import math
    torch.manual_seed(0)

    def check(x, b=None, what=""):
        mine = engine.reference_cpu_norm(x, b)
        ref = torch.norm(x.float() - (b.float() if b is not None else 0)).item()
        assert mine == ref or (math.isnan(mine) and math.isnan(ref)), f"{what} n={x.numel()}: {mine!r} vs torch {ref!r}"

    for n in (1, 5, 8, 9, 12, 13, 15, 1000, 4099, 4100, 4102, 65536, 3 * 65536 + 7, (1 << 20) + 5, 1 << 22):
        check(torch.randn(n) * 0.003, what="gauss")
        check(torch.randn(n) * 300.0, what="large")
        for dt in (torch.bfloat16, torch.float16):
            b = (torch.randn(n) * 0.02).to(dt)
            check((b.float() + torch.randn(n) * 0.003).to(dt), b, what=str(dt))
    n = 1 << 21
    x = torch.randn(n) * 0.003
    x[n // 2:] *= 100
    check(x, what="step up")
    x = torch.randn(n) * 0.003
    x[::8] *= 50
    check(x, what="one lane 50x larger")
    x = torch.zeros(n)
    x[n // 2:] = torch.randn(n // 2) * 0.01
    check(x, what="zeros first")
    check(torch.randn(n) * 1e-22, what="squares underflow")
    check(torch.randn(n) * 1e19, what="overflow to inf")
    check(torch.full((n,), 0.0078125), what="all ties")
    check(torch.randn(n).abs() ** 8 * 1e-3, what="heavy tail")
    x = torch.randn(n) * 0.003
    x[777777] = float("nan")
    check(x, what="nan")
    # the walker composed most chunks from their summaries (this is not the serial chain in disguise)
    engine.reference_cpu_norm(torch.randn(1 << 22) * 0.003)
    assert engine.ctx.debug_query("aten_fast") > 400 and engine.ctx.debug_query("aten_slow") <= 16
    # and the old serial single-work-group chain (test hook) agrees
    engine.ctx.debug_option("aten_serial", 1)
    try:
        x = torch.randn(1 << 16) * 0.003
        assert engine.reference_cpu_norm(x) == torch.norm(x).item()
    finally:
        engine.ctx.debug_option("aten_serial", 0)


def test_reference_cpu_mode_follows_the_biased_norms_everywhere(engine):
    """norm_mode = reference_cpu at a size where torch.norm's bias is visible (2048^2: -1e-5 on the slerp cosine):
    the layer's cosine follows the AS-IS oracle's - with the class norms modelled from sampled statistics (default),
    with the ordered emulation over the planes (test hook class_norms = 2) - and not the exact-norm oracle's, which
    is what leaving the class norms exact (class_norms = 0) gives."""
    from oracle import spectral_oracle as so
    base, fts = so.synthetic_layer(2048, 2048, 2, seed=6048)
    tr, trx = so.LayerTrace(), so.LayerTrace()
    so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=tr)
    with so.exact_norms():
        so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=trx)
    d_ref, d_exact = tr.steps[0].dot, trx.steps[0].dot
    assert abs(d_ref - d_exact) > 5e-6                      # the effect under test exists at this size
    got = {}
    try:
        for mode in (1, 2, 0):
            engine.ctx.debug_option("class_norms", mode)
            out, rep, delta = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True, norm_mode="reference_cpu")
            got[mode] = rep.infos[0].dot
            assert rep.target_norm == tr.target_norm
            assert so.rel_err(delta.cpu(), tr.merged_delta) < 3e-4
    finally:
        engine.ctx.debug_option("class_norms", 1)
    assert abs(got[1] - d_ref) < 0.25 * abs(d_ref - d_exact), (got, d_ref, d_exact)
    assert abs(got[2] - d_ref) < 0.25 * abs(d_ref - d_exact), (got, d_ref, d_exact)
    assert abs(got[0] - d_exact) < 0.25 * abs(d_ref - d_exact), (got, d_ref, d_exact)


def test_reference_cpu_mode_keeps_k3_intermediates_spectral(engine):
    """K = 3 in norm_mode = reference_cpu: the intermediate stays in the spectral domain (its torch.norm is modelled
    from its exact Parseval norm: aten_gauss_norm_ratio) and the layer still follows the as-is oracle's steps; with
    the intermediates materialised (test hook) - whose norm is then emulated exactly - the result agrees."""
    from oracle import spectral_oracle as so
    base, fts = so.synthetic_layer(512, 1024, 3, seed=7103)
    tr = so.LayerTrace()
    so.merge_layer(fts, [base] * 3, so.ALPHAS[:3], base, trace=tr)
    out_s, rep_s, delta_s = engine.merge_layer(fts, [base] * 3, so.ALPHAS[:3], base, want_delta=True, norm_mode="reference_cpu")
    pc.check_layer_steps(rep_s, tr, out_s.numel())
    engine.ctx.debug_option("spectral_intermediates", 0)
    try:
        out_m, rep_m, delta_m = engine.merge_layer(fts, [base] * 3, so.ALPHAS[:3], base, want_delta=True, norm_mode="reference_cpu")
    finally:
        engine.ctx.debug_option("spectral_intermediates", 1)
    pc.check_layer_steps(rep_m, tr, out_m.numel())
    assert rep_s.target_norm == rep_m.target_norm == tr.target_norm
    assert so.rel_err(delta_s.cpu(), delta_m.cpu()) < 2e-2         # the K = 3 floor: both within it


@pytest.mark.parametrize("k", [2, 3])
def test_rank3_tensor_is_a_batch_of_transforms_with_global_statistics(engine, k):
    """The reference transforms the LAST TWO dims of an N-D tensor (fftn(dim=(-2,-1)),
    functions.py:55-58) while its norms, quantiles and slerp sums are flat over the whole tensor
    (fused MoE expert weights [E, R, C]).  merge_layer takes such a tensor as `batch` slices."""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(40 + k)
    shape = (3, 128, 256) if k == 2 else (2, 2, 128, 128)
    base = (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(shape, generator=g) * s_).to(torch.bfloat16) for s_ in (0.002, 0.003, 0.0025)[:k]]
    tr = so.LayerTrace()
    ref = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=tr)
    out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
    out, delta = out.cpu(), delta.cpu()
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    pc.check_layer_steps(rep, tr, out.numel())
    r, c = shape[-2], shape[-1]
    if k == 2:
        d_total, d_resid = pc.spectral_residual(delta.reshape(-1, c), tr.merged_delta.reshape(-1, c))
        assert d_resid < 2e-5 and so.rel_err(out.float(), ref.float()) < 1e-3
        # NOT the same as merging the slices one by one (per-slice thresholds and norms)
        one = engine.merge_layer([f[0] for f in fts], [base[0]] * k, so.ALPHAS[:k], base[0], want_delta=True)[2].cpu()
        assert so.rel_err(one, tr.merged_delta[0]) > 1e-3
    else:
        assert so.rel_err(out.float(), ref.float()) < 5e-3
        assert so.rel_err(delta, tr.merged_delta) < 3e-2          # K = 3: the reference's own chaos floor
    x = torch.randn(2, 3, 64, 96, generator=g)
    f = engine.fft_transform(x).cpu()
    assert so.rel_err(torch.view_as_real(f), torch.view_as_real(so.fft_transform(x))) < 3e-6
    assert so.rel_err(engine.ifft_transform(f).cpu(), x) < 3e-6
    with pytest.raises(NotImplementedError):
        engine.merge_tensors_fft2_slerp(x, x, 0.5)               # the function-level pair API is 1-D / 2-D


@pytest.mark.parametrize("case", gi.CORR_CASES, ids=lambda c: c["id"])
def test_correlate_pairs(engine, golden, case):
    """The legacy operator's pairing matrix (functions.py:304-314) against the reference's values."""
    want = torch.tensor(golden.manifest["corr"][case["id"]])
    got = engine.correlate_pairs(gi.corr_input(case))
    assert got.shape == want.shape and torch.equal(got, got.T) and float(got.diagonal().abs().max()) == 0.0
    assert float((got - want).abs().max()) < 2e-7
    from shardmerge_amd.tensor import functions as fn      # and the pairing that follows it
    pairs_ref = [(x, y) for x, y, _ in fn.correlated_pairs(want, "least")]
    assert [(x, y) for x, y, _ in fn.correlated_pairs(got, "least")] == pairs_ref


def _layer_inputs(shape, k, seed):
    g = torch.Generator().manual_seed(seed)
    base = (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(shape, generator=g) * s_).to(torch.bfloat16)
           for s_ in (0.002, 0.003, 0.0025, 0.0035)[:k]]
    return base, fts


@pytest.mark.parametrize("shape", [(34, 64), (136, 96), (76, 96), (172, 128), (272, 64), (668, 32)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("k", [2, 3])
def test_column_length_without_a_plan_is_split_into_row_blocks(engine, shape, k):
    """A column length with a prime factor > 13 (11008 = 43 * 256 of Llama-2-7B, 18944 = 37 * 512 of
    Qwen2-7B; here 34 = 17 * 2, 136 = 17 * 8, 76 = 19 * 4, 172 = 43 * 4, 272 = 17 * 16, 668 = 167 * 4: p > 126 takes the generic k_dftp) is merged as p row blocks of M
    rows that k_dftp combines (sm_kernels.hpp) - same bar as any other length."""
    from oracle import spectral_oracle as so
    assert engine.lib.shape_supported(*shape) and not engine.lib.length_supported(shape[0])
    base, fts = _layer_inputs(shape, k, 100 + k)
    tr = so.LayerTrace()
    ref = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=tr)
    out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
    pc.check_layer_steps(rep, tr, out.numel())
    if k == 2:
        assert pc.spectral_residual(delta, tr.merged_delta)[1] < 2e-5
        assert so.rel_err(out.float(), ref.float()) < 8.0 / (out.numel() ** 0.5)
    else:           # K = 3: the reference's own chaos floor, as for any other length
        assert so.rel_err(out.float(), ref.float()) < 5e-3 and so.rel_err(delta, tr.merged_delta) < 3e-2


@pytest.mark.parametrize("p", [2, 4, 8])
@pytest.mark.parametrize("k", [2, 3])
def test_forced_split_agrees_with_the_plain_column_pass(engine, p, k):
    """The same tensor through the plain column pass and through p row blocks + k_dftp (test hook
    force_split; what a length > 32768 takes with p = 2): same branches, thresholds and result."""
    from oracle import spectral_oracle as so
    base, fts = _layer_inputs((256, 128), k, 7 + p)
    plain = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
    engine.ctx.debug_option("force_split", p)
    try:
        split = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
    finally:
        engine.ctx.debug_option("force_split", 0)
    assert split[1].branches == plain[1].branches and split[1].steps == plain[1].steps
    a, b = split[1].infos[0], plain[1].infos[0]          # the first pair merges raw deltas: rounding-level agreement
    assert abs(a.cutoff_threshold - b.cutoff_threshold) <= 1e-5 * abs(b.cutoff_threshold)
    assert abs(a.cull_threshold - b.cull_threshold) <= 5e-5 * b.cull_threshold and abs(a.n_slerp - b.n_slerp) <= 4
    if k == 2:
        assert pc.spectral_residual(split[2], plain[2])[1] < 2e-6
    else:
        assert so.rel_err(split[2], plain[2]) < 3e-2           # K = 3: the chaos floor, as between any two FFTs


@pytest.mark.parametrize("shape", [(64, 34), (96, 76), (1, 68), (128, 2 * 43)], ids=lambda s: "x".join(map(str, s)))
def test_row_length_without_a_plan_is_merged_transposed(engine, shape):
    """When the ROW length is the one without a plan the operands are transposed on the device, merged
    ([C x R]: the rough length is the column length now) and the result is transposed back.  fft2
    commutes with the transpose; the reference's result, though, depends on the orientation in the
    bins that sit ON a threshold (its one-sided transform stores the self-conjugate column's twin bins
    separately, the other twins as exact copies: which pairs straddle a rank changes) - so the bar
    is the usual one against the oracle run in EITHER orientation."""
    from oracle import spectral_oracle as so
    assert engine.lib.shape_supported(*shape) and not engine.lib.length_supported(shape[1])
    base, fts = _layer_inputs(shape, 2, 300)
    out, rep, delta = engine.merge_layer(fts, [base] * 2, so.ALPHAS[:2], base, want_delta=True)
    assert out.shape == base.shape and delta.shape == base.shape
    resid = []
    for flip in (False, True):
        tr = so.LayerTrace()
        tt = (lambda x: x.T.contiguous()) if flip else (lambda x: x)
        ref = so.merge_layer([tt(f) for f in fts], [tt(base)] * 2, so.ALPHAS[:2], tt(base), trace=tr)
        pc.check_layer_steps(rep, tr, out.numel())
        resid.append((pc.spectral_residual(tt(delta), tr.merged_delta)[1], so.rel_err(tt(out).float(), ref.float())))
    assert min(r[0] for r in resid) < 2e-5 and max(r[0] for r in resid) < 2e-2
    assert min(r[1] for r in resid) < 8.0 / (out.numel() ** 0.5)
    # K = 3 through the same path
    base, fts = _layer_inputs(shape, 3, 301)
    tr = so.LayerTrace()
    ref = so.merge_layer(fts, [base] * 3, so.ALPHAS[:3], base, trace=tr)
    out, rep, delta = engine.merge_layer(fts, [base] * 3, so.ALPHAS[:3], base, want_delta=True)
    assert rep.branches == tr.branches and so.rel_err(out.float(), ref.float()) < 1e-2


@pytest.mark.parametrize("shape", [(64, 96), (32, 256), (128, 40)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("k", [2, 3])
def test_chirp_z_row_passes_agree_with_the_planned_ones(engine, shape, k):
    """k_f1b / k_i2b (sm_bluestein.hpp: the row passes of a row length without a plan) forced onto lengths that have
    one (test hook force_bluestein): same branches, thresholds and result as the planned row passes."""
    from oracle import spectral_oracle as so
    base, fts = _layer_inputs(shape, k, 17 + k)
    plain = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
    engine.ctx.debug_option("force_bluestein", 1)
    try:
        blue = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
    finally:
        engine.ctx.debug_option("force_bluestein", 0)
    assert blue[1].branches == plain[1].branches and blue[1].steps == plain[1].steps
    a, b = blue[1].infos[0], plain[1].infos[0]
    assert abs(a.cutoff_threshold - b.cutoff_threshold) <= 1e-5 * abs(b.cutoff_threshold)
    assert abs(a.cull_threshold - b.cull_threshold) <= 5e-5 * b.cull_threshold and abs(a.n_slerp - b.n_slerp) <= 4
    if k == 2:
        assert pc.spectral_residual(blue[2], plain[2])[1] < 2e-6
    else:
        assert so.rel_err(blue[2], plain[2]) < 3e-2           # K = 3: the chaos floor, as between any two FFTs


@pytest.mark.parametrize("shape", [(34, 38), (68, 76), (142, 142), (172, 86), (34, 17), (17, 32), (17, 128)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("k", [2, 3])
def test_both_lengths_without_a_plan(engine, shape, k):
    """Falcon-7B's tensors (4544 = 71 * 64, 4672 = 73 * 64) have no planned length at all: the column length is split
    into row blocks as before, the rows go through the chirp-z row passes.  An ODD rough length (17) can only be the
    row length - such a tensor is merged transposed when its other length can be the column length.  Same bar as
    any other shape; in the masked residual the bins that sit ON a threshold are left out (the reference's result
    depends on the orientation there, see the transposed test above).  (Not here: [64 x 17], [64 x 19], [64 x 23] -
    on those the REFERENCE's imaginary detour, functions.py:152-158, divides 0 by 0, its NaNs are zeroed and the
    merged delta loses 95 % of its norm; the transposed tensors are fine.)"""
    from oracle import spectral_oracle as so
    assert engine.lib.shape_supported(*shape)
    assert not engine.lib.length_supported(shape[0]) or not engine.lib.length_supported(shape[1])
    base, fts = _layer_inputs(shape, k, 400 + k)
    out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
    assert out.shape == base.shape and delta.shape == base.shape
    flip = not engine.lib.length_supported(shape[0]) and shape[0] % 2 == 1         # merged transposed
    tt = (lambda x: x.T.contiguous()) if flip else (lambda x: x)
    tr = so.LayerTrace()
    ref = so.merge_layer([tt(f) for f in fts], [tt(base)] * k, so.ALPHAS[:k], tt(base), trace=tr)
    if k == 2:
        pc.check_layer_steps(rep, tr, out.numel())
        assert pc.spectral_residual(tt(delta), tr.merged_delta)[1] < 2e-5
        assert so.rel_err(tt(out).float(), ref.float()) < max(2e-3, 8.0 / (out.numel() ** 0.5))
    else:           # K = 3 at these sizes: one bin flipped in round 1 moves round 2's thresholds by a per cent
        assert rep.branches == tr.branches and [(s_[0], s_[1]) for s_ in rep.steps] == tr.pairs
        assert so.rel_err(tt(out).float(), ref.float()) < 5e-3 and so.rel_err(tt(delta), tr.merged_delta) < 3e-2


@pytest.mark.parametrize("shape", [(64, 38), (32, 17), (1, 38), (128, 142)], ids=lambda s: "x".join(map(str, s)))
def test_function_level_transforms_on_a_row_length_without_a_plan(engine, shape):
    """fft_transform / ifft_transform (A4, A8) and the pair merge (A9) with a rough ROW length: the chirp-z row passes."""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(shape[0] * 1000 + shape[1])
    x = torch.randn(shape, generator=g).squeeze(0) if shape[0] == 1 else torch.randn(shape, generator=g)
    f = engine.fft_transform(x).cpu()
    assert so.rel_err(torch.view_as_real(f), torch.view_as_real(so.fft_transform(x))) < 3e-6
    assert so.rel_err(engine.ifft_transform(f).cpu(), x) < 3e-6
    if x.ndim == 2:
        y = x * 0.7 + 0.5 * torch.randn(shape, generator=g)
        tr = so.BlendTrace()
        ref, n0, n1 = so.merge_tensors_fft2_slerp(x, y, t=0.4, cutoff_pct=0.08, cull_pct=0.2, trace=tr)
        out = engine.merge_tensors_fft2_slerp(x, y, t=0.4, cutoff_pct=0.08, cull_pct=0.2)[0].cpu()
        assert pc.spectral_residual(out, ref)[1] < 2e-5


def test_shapes_no_orientation_can_take_are_refused(engine):
    assert not engine.lib.shape_supported(17, 19)            # two odd rough lengths: neither can be the column length
    assert not engine.lib.shape_supported(2 * 17, 2 * 16411) # rough and too long for the chirp-z row pass
    base, fts = _layer_inputs((17, 19), 2, 5)
    with pytest.raises(NotImplementedError):
        engine.merge_layer(fts, [base] * 2, [1.0, 1.0], base)
    base, fts = _layer_inputs((2, 34, 8), 2, 5)              # rank > 2: the slices' lengths need plans
    with pytest.raises(NotImplementedError):
        engine.merge_layer(fts, [base] * 2, [1.0, 1.0], base)


@pytest.mark.parametrize("shape", [(3, 64, 128), (172, 128), (64, 86)], ids=lambda s: "x".join(map(str, s)))
def test_arith_and_linear_branches_on_sliced_geometries(engine, shape):
    """The Arithmetic-FFT branch (ratio < 0.1) and the linear blend (ratio < b) run on the layer's own
    geometry: every slice of a rank > 2 tensor (the arithmetic branch used to transform the first slice
    only), the row blocks of a split column length, the transposed operands of a rough row length."""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(9)
    base = (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(shape, generator=g) * s_).to(torch.bfloat16) for s_ in (0.004, 0.0002)]
    tr = so.LayerTrace()
    ref = so.merge_layer(fts, [base] * 2, [1.0, 1.0], base, trace=tr)
    out, rep, delta = engine.merge_layer(fts, [base] * 2, [1.0, 1.0], base, want_delta=True)
    assert tr.branches == ["arith"] and rep.branches == ["arith"]
    assert so.rel_err(delta, tr.merged_delta) < 2e-6 and so.rel_err(out.float(), ref.float()) < 1e-4       # a bf16 rounding flip or two
    fts = [(base.float() + torch.randn(shape, generator=g) * s_).to(torch.bfloat16) for s_ in (0.004, 0.0015)]
    tr = so.LayerTrace()
    ref = so.merge_layer(fts, [base] * 2, [1.0, 1.0], base, ratio_b=0.6, trace=tr)
    out, rep, delta = engine.merge_layer(fts, [base] * 2, [1.0, 1.0], base, b=0.6, want_delta=True)
    assert tr.branches == ["linear"] and rep.branches == ["linear"]
    assert so.rel_err(delta, tr.merged_delta) < 2e-6 and so.rel_err(out.float(), ref.float()) < 1e-3


def test_generic_dft_kernel_agrees_with_the_paired_one(engine):
    """k_dftp_pairs (p <= 126) and the generic k_dftp (any p <= 256) on the same tensor, both directions
    and both T1 layouts (K = 2: two-signal row pass; K = 3: row pairs)."""
    from oracle import spectral_oracle as so
    for k, shape in ((2, (172, 128)), (3, (136, 96))):
        base, fts = _layer_inputs(shape, k, 55)
        fast = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
        engine.ctx.debug_option("dftp_pairs", 0)
        try:
            slow = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
        finally:
            engine.ctx.debug_option("dftp_pairs", 1)
        assert fast[1].branches == slow[1].branches
        if k == 2:
            assert pc.spectral_residual(fast[2], slow[2])[1] < 2e-6
        else:
            assert so.rel_err(fast[2], slow[2]) < 3e-2


@pytest.mark.parametrize("k,shape", [(2, (128, 256)), (3, (256, 128)), (2, (63, 40)), (2, (3, 64, 128))],
                         ids=["k2", "k3", "k2_unaligned", "k2_rank3"])
def test_speculative_cull_selection_never_changes_a_bit(engine, k, shape):
    """The SLERP blend guesses the level-1 bin of the cull threshold (what it was the last time this tournament
    round ran) and does the cull selection's level-2 pass in the same sweep; k_spec_check confirms or voids
    it.  Confirmed, voided and switched off must give the same bits - outputs, thresholds, norms."""
    from oracle import spectral_oracle as so
    base, fts = _layer_inputs(shape, k, 77)
    args = (fts, [base] * k, so.ALPHAS[:k], base)
    engine.ctx.debug_option("spec_cull", 0)
    try:
        off = engine.merge_layer(*args, want_delta=True)
    finally:
        engine.ctx.debug_option("spec_cull", 1)
    engine.ctx.debug_option("spec_min_bins", 0)          # (small spectra do not speculate by default)
    try:
        _speculation_checks(engine, args, off, k)
    finally:
        engine.ctx.debug_option("spec_min_bins", -1)


@pytest.mark.parametrize("k,shape", [(3, (64, 128)), (4, (128, 64)), (3, (66, 40))], ids=["k3", "k4", "k3_unaligned"])
def test_row_passes_of_all_deltas_in_one_launch_change_nothing(engine, k, shape):
    """rows_first puts the row passes of all K raw deltas into ONE launch (the work-groups of a row block's K signals
    on one XCD: their shared base rows come out of that L2).  Same bits as one launch per signal (test hook f1_multi)."""
    from oracle import spectral_oracle as so
    base, fts = _layer_inputs(shape, k, 91)
    args = (fts, [base] * k, so.ALPHAS[:k], base)
    one = engine.merge_layer(*args, want_delta=True)
    engine.ctx.debug_option("f1_multi", 0)
    try:
        each = engine.merge_layer(*args, want_delta=True)
    finally:
        engine.ctx.debug_option("f1_multi", 1)
    assert torch.equal(one[0].view(torch.int16), each[0].view(torch.int16)) and torch.equal(one[2], each[2])
    assert one[1].delta_norms == each[1].delta_norms and one[1].branches == each[1].branches


@pytest.mark.parametrize("k,shape", [(3, (64, 128)), (4, (128, 64)), (3, (66, 40)), (5, (32, 64))], ids=["k3", "k4", "k3_unaligned", "k5"])
def test_column_passes_of_a_raw_pair_in_one_launch_change_nothing(engine, k, shape):
    """K >= 3: the two single-signal column passes of a pair of raw deltas run as ONE launch (test hook f2s_pair).  Same
    branches and thresholds; the Parseval partial sums are reduced in another association, so the result may differ
    in the last bits of an fp32 value - not more."""
    from oracle import spectral_oracle as so
    base, fts = _layer_inputs(shape, k, 93)
    args = (fts, [base] * k, so.ALPHAS[:k], base)
    one = engine.merge_layer(*args, want_delta=True)
    engine.ctx.debug_option("f2s_pair", 0)
    try:
        two = engine.merge_layer(*args, want_delta=True)
    finally:
        engine.ctx.debug_option("f2s_pair", 1)
    assert one[1].branches == two[1].branches and one[1].delta_norms == two[1].delta_norms
    for a, b in zip(one[1].infos, two[1].infos):
        assert (a is None) == (b is None)
        if a is not None:
            assert (a.cutoff_threshold, a.n_slerp) == (b.cutoff_threshold, b.n_slerp)
    assert so.rel_err(one[2], two[2]) < 1e-6
    assert (one[0].view(torch.int16) != two[0].view(torch.int16)).float().mean() < 1e-3


def test_cull_speculation_on_alternating_layers_misses_and_changes_nothing(engine):
    """The guess is the threshold's level-1 bin of the previous layer.  The spectra of unit-norm deltas look alike
    whatever the deltas' scale or the tensor's size (|X| ~ 1 in every bin: the bench's and real models' hit rates are
    ~100 %), so the layers here differ in STRUCTURE - white noise, a low-rank delta (its energy in a few bins), a
    smooth one - which puts their cull thresholds octaves apart: every guess is wrong.  The counters the bench
    reports (`spec_checked`, `spec_hits`) must say so and every result must be the one of the non-speculating run,
    bit for bit; the same layer twice in a row then hits."""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(5)
    layers = []
    shape = (64, 128)
    for kind in ("white", "lowrank", "smooth"):
        base = (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
        fts = []
        for s_ in (0.002, 0.003):
            d = torch.randn(shape, generator=g) * s_
            if kind == "lowrank":
                d = 0.02 * d + 30 * s_ * torch.randn(shape[0], 1, generator=g) * torch.randn(1, shape[1], generator=g)
            elif kind == "smooth":
                d = 0.01 * d + torch.cumsum(torch.cumsum(d, 0), 1) * 0.02
            fts.append((base.float() + d).to(torch.bfloat16))
        layers.append((fts, [base] * 2, so.ALPHAS[:2], base))
    engine.ctx.debug_option("spec_cull", 0)
    try:
        off = [engine.merge_layer(*a, want_delta=True) for a in layers]
    finally:
        engine.ctx.debug_option("spec_cull", 1)
    bins = {round(float(torch.log2(torch.tensor(o[1].infos[0].cull_threshold))) * 8) for o in off}
    assert len(bins) == 3, "the three layers' cull thresholds must sit in different level-1 bins for this test"
    engine.ctx.debug_option("spec_min_bins", 0)
    try:
        engine.merge_layer(*layers[0])                                   # leaves a guess behind
        c0, h0 = engine.ctx.debug_query("spec_checked"), engine.ctx.debug_query("spec_hits")
        order = [1, 2, 0, 1, 2, 0]
        for i in order:
            got = engine.merge_layer(*layers[i], want_delta=True)
            assert torch.equal(got[0].view(torch.int16), off[i][0].view(torch.int16)) and torch.equal(got[2], off[i][2])
        assert engine.ctx.debug_query("spec_checked") - c0 == len(order)
        assert engine.ctx.debug_query("spec_hits") - h0 == 0
        engine.merge_layer(*layers[0])                                   # the same layer again: a hit
        assert engine.ctx.debug_query("spec_hits") - h0 == 1
    finally:
        engine.ctx.debug_option("spec_min_bins", -1)


def _speculation_checks(engine, args, off, k):
    # a different cull fraction moves the threshold into another bin: the next guess is wrong
    engine.merge_layer(*args, cull_start_pct=0.45)
    miss = engine.merge_layer(*args, want_delta=True)
    assert engine.ctx.debug_query("spec_hit") == 0 or k == 3      # (K = 3: the verdict read is the LAST round's)
    hit = engine.merge_layer(*args, want_delta=True)
    assert engine.ctx.debug_query("spec_hit") == 1
    for other in (miss, hit):
        assert torch.equal(other[0].view(torch.int16), off[0].view(torch.int16))
        assert torch.equal(other[2], off[2])
        assert other[1].branches == off[1].branches and other[1].delta_norms == off[1].delta_norms
        for a, b in zip(other[1].infos, off[1].infos):
            assert (a is None) == (b is None)
            if a is not None:
                assert (a.cutoff_threshold, a.cull_threshold, a.n_slerp, a.dot) == (b.cutoff_threshold, b.cull_threshold, b.n_slerp, b.dot)
    # ... and the candidate-list overflow on a confirmed guess still falls back to the full-pass selection
    engine.ctx.debug_option("cand_cap", 7)
    try:
        capped = engine.merge_layer(*args, want_delta=True)
    finally:
        engine.ctx.debug_option("cand_cap", 0)
    assert torch.equal(capped[0].view(torch.int16), off[0].view(torch.int16))


@pytest.mark.parametrize("shape", [(256,), (1000,), (4096,), (1, 512), (8192,), (77,), (6144,)], ids=lambda s: "x".join(map(str, s)))
def test_one_launch_pair_merge_of_1d_tensors(engine, shape):
    """1-D tensors (every norm weight of a model) take k_pair1d: the whole SLERP pair merge - transform,
    split, both order statistics, slerp constants, blend, cull, inverse, add-back - in one work-group
    instead of ~25 launches.  Same bar as the multi-kernel pipeline, and the same thresholds as it."""
    from oracle import spectral_oracle as so
    base, fts = _layer_inputs(shape, 2, 400)
    tr = so.LayerTrace()
    ref = so.merge_layer(fts, [base] * 2, so.ALPHAS[:2], base, trace=tr)
    eng_out = engine.merge_layer(fts, [base] * 2, so.ALPHAS[:2], base, want_delta=True)
    engine.ctx.debug_option("pair1d", 0)
    try:
        multi = engine.merge_layer(fts, [base] * 2, so.ALPHAS[:2], base, want_delta=True)
    finally:
        engine.ctx.debug_option("pair1d", 1)
    out, rep, delta = eng_out
    pc.check_layer_steps(rep, tr, out.numel())
    assert rep.branches == ["slerp"]
    assert pc.spectral_residual(delta, tr.merged_delta)[1] < 2e-6 and so.rel_err(out.float(), ref.float()) < 8.0 / out.numel() ** 0.5
    a, b = rep.infos[0], multi[1].infos[0]
    # the two paths round their spectra differently (Hermitian split before / after the scaling): ulps
    assert abs(a.cutoff_threshold - b.cutoff_threshold) <= 1e-5 * b.cutoff_threshold and abs(a.n_slerp - b.n_slerp) <= 2
    assert abs(a.cull_threshold - b.cull_threshold) <= 1e-5 * b.cull_threshold and abs(a.dot - b.dot) < 1e-5
    assert pc.spectral_residual(delta, multi[2])[1] < 1e-6
    # K = 3 / 4: the tournament around it (fp32 intermediates, the next round's norm from the kernel's own partial)
    for k in (3, 4):
        base, fts = _layer_inputs(shape, k, 401 + k)
        tr = so.LayerTrace()
        ref = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=tr)
        out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
        assert rep.branches == tr.branches and [(s[0], s[1]) for s in rep.steps] == tr.pairs
        # first-round pairs merge raw deltas: thresholds as tight as at K = 2 (later rounds: the chaos floor)
        first = [i for i, bt in enumerate(tr.steps) if bt is not None][0]
        assert abs(rep.infos[first].cutoff_threshold - tr.steps[first].cutoff_threshold) <= max(1e-5, 8.0 / out.numel()) * tr.steps[first].cutoff_threshold
        assert abs(rep.infos[first].n_slerp - tr.steps[first].n_slerp) <= 4
        assert abs(rep.target_norm - tr.target_norm) <= 2e-6 * tr.target_norm
        assert so.rel_err(out.float(), ref.float()) < (1e-2 if k == 3 else 8e-2)      # manifest layer_self_floor x 2.5


def test_1d_policies_through_the_one_launch_kernel(engine):
    """NaN -> 0 counts and the Inf error of the inverse transform / add-back, no cull, no cutoff, unaligned views"""
    from oracle import spectral_oracle as so
    g = torch.Generator().manual_seed(3)
    base = (torch.randn(1024 + 3, generator=g) * 0.02).to(torch.bfloat16)[3:]          # 6-byte offset: unaligned
    fts = [(base.float() + torch.randn(1024, generator=g) * s_).to(torch.bfloat16) for s_ in (0.002, 0.003)]
    for kw in ({}, {"cull_start_pct": 0.0}, {"cutoff_pct": 0.0}, {"cutoff_pct": 0.0, "cull_start_pct": 0.0}):
        tr = so.LayerTrace()
        okw = dict(kw)
        ref = so.merge_layer(fts, [base] * 2, so.ALPHAS[:2], base, trace=tr, **okw)
        out, rep, delta = engine.merge_layer(fts, [base] * 2, so.ALPHAS[:2], base, want_delta=True, **kw)
        assert pc.spectral_residual(delta, tr.merged_delta)[1] < 2e-6, kw
    big = base.clone()
    big[5] = float("inf")
    with pytest.raises(ValueError):
        engine.merge_layer(fts, [base] * 2, so.ALPHAS[:2], big)


# ---- K >= 3 against reference-held floor data (tests/golden/g11_floor.safetensors, oracle/gen_golden.py --only-floor) --
@pytest.mark.parametrize("case", gi.FLOOR_CASES, ids=lambda c: c["id"])
def test_k3_k4_within_the_reference_own_floor(engine, golden, case):
    """1024 x 1024, K = 3 and K = 4, the product's default mode: d(HIP, reference) <= 1.25 d(reference with fp64 FFTs,
    reference) on the bf16 output and on the merged delta (reference fast_fourier.py:171-254,
    functions.py:124-148)."""
    rec = pc.check_floor(engine, golden, case)
    print(rec)


def test_noise_model_seed_does_not_matter_beyond_the_floor(engine, golden):
    d01, d02, floor = pc.check_noise_seed_sensitivity(engine, golden, gi.FLOOR_CASES[0])
    print(f"two other seeds move the K = 3 output by {d01:.2e} / {d02:.2e}; the reference's own floor is {floor:.2e}")


@pytest.mark.parametrize("k,shape,fold", [(2, (64, 2048), False), (2, (48, 4096), False), (3, (24, 8192), False),
                                         (3, (16, 16384), False), (3, (8192, 2048), True)],
                         ids=lambda v: str(v).replace(" ", ""))
def test_row_pass_summarises_its_deltas_for_the_norm_emulation(engine, k, shape, fold):
    """norm_mode = reference_cpu, round 4: the forward row pass (k_f1 / k_f1q, AtenFuse) evaluates the torch.norm chunk
    summaries on the deltas it has just formed - two real fma chains per lane and candidate binade over the natural-order
    row in LDS, composed in order across the wave - instead of a second pass over every finetune and base
    (k_aten_part).  The norms stay torch.norm's BIT FOR BIT (reference functions.py:85, fast_fourier.py:152,209-210),
    equal to the separate pass's, and the walker composes the chunks from those summaries (it is not walking the data)."""
    from oracle import spectral_oracle as so
    rows, cols = shape
    base, fts = so.synthetic_layer(rows, cols, k, seed=4400 + rows + k)
    torch_norms = [float((f.float() - base.float()).norm()) for f in fts]
    if fold:
        engine.ctx.debug_option("fold_min_rows", 8192)
    engine.ctx.profile(True)
    try:
        got = {}
        for fuse in (1, 0):
            engine.ctx.debug_option("fuse_norms", fuse)
            engine.ctx.profile_reset()
            out, rep = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, norm_mode="reference_cpu")
            names = set(engine.ctx.profile_table())
            fused = bool(fuse) and not fold          # (the folded row pass k_f1q keeps the separate summary pass: see there)
            assert ("aten_norm_rec" in names) == fused and ("aten_norm_part" in names) == (not fused), names
            assert rep.delta_norms == torch_norms, (fuse, rep.delta_norms, torch_norms)
            stats = [engine.ctx.debug_query(q) for q in ("aten_fast", "aten_group", "aten_slow")]
            assert stats[0] + stats[1] > 0, stats            # chunks composed from summaries / crossed by group summaries
            got[fuse] = (out.cpu(), stats)
        assert torch.equal(got[0][0], got[1][0])
        assert got[0][1] == got[1][1]                        # the walker finds the same summaries usable either way
    finally:
        engine.ctx.debug_option("fuse_norms", 1)
        engine.ctx.debug_option("fold_min_rows", 0)
        engine.ctx.profile(False)


@pytest.mark.parametrize("case", gi.LEGACY_CASES, ids=lambda c: c["id"])
def test_legacy_fourier_operator(engine, golden, case):
    """the reference's older FourierMerge class (shard/merge/fourier.py:35-205) behind the same boundary"""
    print(pc.check_legacy(engine, golden, case))
