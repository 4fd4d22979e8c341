"""GPU tier (-m gpu): parity of the HIP path (libshardmerge_hip.so through the C
ABI) with the reference's golden vectors and the CPU oracle, plus size-independent
properties at BASELINE's full 8192 x 8192 size."""
import math

import pytest
import torch

from tests import parity_checks as pc
from tests.golden import inputs as gi
from oracle import spectral_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from shardmerge_amd.engine import get_engine
    return get_engine("cuda")


def test_native_library_is_loaded(engine):
    import os
    assert "gfx950" in engine.lib.version()
    with open(f"/proc/{os.getpid()}/maps") as f:
        assert "libshardmerge_hip.so" in f.read()


@pytest.mark.parametrize("case", gi.FFT_CASES, ids=lambda c: c["id"])
def test_fft(engine, golden, case):
    pc.check_fft(engine, golden, case)


@pytest.mark.parametrize("case", gi.INTERP_CASES, ids=lambda c: c["id"])
def test_interpolate(engine, golden, case):
    pc.check_interp(engine, golden, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", gi.SLERP_CASES, ids=lambda c: c["id"])
def test_function_level_slerp(engine, golden, case):
    pc.check_fn_slerp(engine, golden, case)


@pytest.mark.gpu
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


# ---- mid sizes against the oracle (seconds of CPU work) -----------------------------
# From ~1M elements on, torch's CPU norm kernel (8-lane serial fp32 accumulation)
# is visibly biased (oracle/norm_bias_probe.py); the reference's output depends on
# the a/b norm RATIO through hard |r0| > |r1| decisions, so its bias moves the
# reference itself by ~3e-3 (merged delta) at 1024^2.  The HIP path takes exact L2
# norms (as the reference's device="cuda" mode does): we therefore check tightly
# against the oracle evaluated with exact norms, and loosely against the oracle as
# the reference is (torch norms).
@pytest.mark.parametrize("shape", [(1024, 1024), (2048, 512), (448, 1024), (1024, 14336 // 8), (1, 8192)],
                         ids=lambda s: f"{s[0]}x{s[1]}")
def test_layer_k2_vs_oracle(engine, shape):
    rows, cols = shape
    base, fts = so.synthetic_layer(rows, cols, 2, seed=4000 + rows)
    tr, trx = so.LayerTrace(), so.LayerTrace()
    ref = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=tr)
    with so.exact_norms():
        refx = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=trx)
    out, rep, delta = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True, norm_mode="exact")
    assert rep.branches == tr.branches == ["slerp"]
    assert abs(rep.target_norm - trx.target_norm) <= 2e-6 * trx.target_norm
    d_total, d_resid = pc.spectral_residual(delta.cpu(), trx.merged_delta)
    assert d_resid < 2e-5, f"vs exact-norm oracle beyond the tie bins: {d_resid:.2e}"
    assert d_total < 8.0 / math.sqrt(rows * cols) + 1e-6
    # BASELINE tolerance: 1e-3 relative on the bf16 output
    assert so.rel_err(out.cpu().float(), refx.float()) < 1e-3
    if rows * cols >= 1 << 20:
        # the stricter delta-level bar of SURVEY 8(d); the tie-bin floor is ~1/sqrt(n)
        assert d_total < (1e-3 if rows * cols >= 1 << 22 else 3e-3)
    # against the reference as it is: bounded by what its own norm artefact does to it
    shift = so.rel_err(ref.float(), refx.float())
    assert so.rel_err(out.cpu().float(), ref.float()) < 1e-3 + 1.5 * shift


@pytest.mark.parametrize("shape", [(28672, 16), (16, 28672), (14336, 32), (8, 4096), (16384, 8), (3072, 40),
                                   (5120, 24), (24, 5120), (6144, 16), (16, 7168), (12288, 8), (16, 3584), (2560, 48), (13824, 8), (8, 13824), (27648, 8), (8, 27648), (2304, 32)],
                         ids=lambda s: f"{s[0]}x{s[1]}")
def test_long_and_odd_lengths_on_device(engine, shape):
    """Llama-3-70B / Mixtral lengths (7 * 2^12, 7 * 2^11, the 8 x 4096 router) and run-time
    planned ones, as transforms and as a K=2 merge against the exact-norm oracle."""
    rows, cols = shape
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g)
    f = engine.fft_transform(x).cpu()
    assert so.rel_err(torch.view_as_real(f), torch.view_as_real(so.fft_transform(x))) < 3e-6
    assert so.rel_err(engine.ifft_transform(f).cpu(), x) < 3e-6
    base, fts = so.synthetic_layer(rows, cols, 2, seed=77 + rows)
    trx = so.LayerTrace()
    with so.exact_norms():
        so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=trx)
    out, rep, delta = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True, norm_mode="exact")
    assert rep.branches == trx.branches
    d_total, d_resid = pc.spectral_residual(delta.cpu(), trx.merged_delta)
    assert d_resid < 2e-5 and d_total < 8.0 / math.sqrt(rows * cols)


def test_random_shapes_and_noise_levels_vs_oracle(engine):
    """A seeded sweep over random supported shapes (run-time planned lengths with factors up to
    13, 1-D tensors, ragged sizes), noise levels and alphas against the exact-norm oracle
    (tools/fuzz_gpu.py runs more of the same)."""
    import random
    rng = random.Random(7)
    lens = [n for n in range(2, 2049) if all(_smooth(n))]
    done = 0
    while done < 14:
        rows = rng.choice(lens + [1] * 200)
        cols = rng.choice(lens)
        if not (256 <= rows * cols <= 1 << 19):
            continue
        done += 1
        g = torch.Generator().manual_seed(rng.randrange(1 << 30))
        shape = (cols,) if rows == 1 else (rows, cols)
        base = (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
        sig = [10 ** rng.uniform(-3.2, -2.0) for _ in range(2)]
        fts = [(base.float() + torch.randn(shape, generator=g) * s_).to(torch.bfloat16) for s_ in sig]
        alphas = [rng.uniform(0.05, 1.0) for _ in range(2)]
        trx = so.LayerTrace()
        with so.exact_norms():
            refx = so.merge_layer(fts, [base, base], alphas, base, trace=trx)
        out, rep, delta = engine.merge_layer([t.cuda() for t in fts], [base.cuda()] * 2, alphas, base.cuda(), want_delta=True, norm_mode="exact")
        assert rep.branches == trx.branches, (rows, cols)
        d_total, d_resid = pc.spectral_residual(delta.cpu().reshape(max(rows, 1), cols), trx.merged_delta.reshape(max(rows, 1), cols))
        assert d_resid < 5e-5, (rows, cols, d_resid)
        assert d_total < 10.0 / math.sqrt(rows * cols) + 1e-5, (rows, cols, d_total)


def _smooth(n):
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            n //= p
    return [n == 1]


@pytest.mark.parametrize("k", [3, 4])
def test_layer_k3_k4_vs_oracle(engine, k):
    """1024 x 1024, K = 3 / 4: every step's thresholds, class counts, cosine, t and cull fraction
    against the oracle trace, and the merged delta's spectrum outside the bins that earlier rounds
    culled (tests/parity_checks.py: what is and is not reproducible from round 2 on)."""
    base, fts = so.synthetic_layer(1024, 1024, k, seed=5000)
    tr = so.LayerTrace()
    with so.exact_norms():
        ref = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=tr)
    out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True, norm_mode="exact")
    pc.check_layer_steps(rep, tr, out.numel())
    outside, inside, flips = pc.masked_spectral_check(delta.cpu(), tr, tol_outside=5e-4 if k == 3 else 1.5e-3)   # 1M elements: tighter than the 64K goldens
    print(f"K={k}: outside the culled bins {outside:.2e}, inside {inside:.2e}, final-cull flips {flips}")
    # the irreproducible part (the reference's own chaos floor on the bf16 output: 1.6e-3 at
    # K = 3, 3.1e-2 at K = 4, every size) bounds the total
    assert so.rel_err(out.cpu().float(), ref.float()) < (2e-3 if k == 3 else 4e-2)


# ---- full-size properties (8192 x 8192: BASELINE's shape) ---------------------------
N_FULL = 8192


@pytest.fixture(scope="module")
def full_inputs():
    g = torch.Generator(device="cpu").manual_seed(77)
    base = (torch.randn(N_FULL, N_FULL, generator=g) * 0.02).to(torch.bfloat16).cuda()
    d0 = torch.randn(N_FULL, N_FULL, generator=g).cuda() * 0.002
    d1 = torch.randn(N_FULL, N_FULL, generator=g).cuda() * 0.003
    return base, d0, d1


def test_full_size_transform_roundtrip_and_parseval(engine, full_inputs):
    _, d0, _ = full_inputs
    f = engine.fft_transform(d0)
    # Parseval: sum |F|^2 = n sum x^2
    lhs = (torch.view_as_real(f).double() ** 2).sum().item()
    rhs = (d0.double() ** 2).sum().item() * d0.numel()
    assert abs(lhs / rhs - 1) < 1e-5
    # Hermitian symmetry of the expanded spectrum at a few thousand random bins
    idx = torch.randint(0, N_FULL, (4096, 2), device="cuda")
    a = f[idx[:, 0], idx[:, 1]]
    b = f[(-idx[:, 0]) % N_FULL, (-idx[:, 1]) % N_FULL]
    assert torch.allclose(a, b.conj(), rtol=0, atol=1e-3 * a.abs().max().item())
    back = engine.ifft_transform(f)
    assert torch.allclose(back, d0, atol=1e-4)                     # reference test_functions.py:114-121
    assert so.rel_err(back.cpu(), d0.cpu()) < 2e-6


def test_full_size_k1_identity(engine, full_inputs):
    base, d0, _ = full_inputs
    ft = (base.float() + d0).to(torch.bfloat16)
    out, rep = engine.merge_layer([ft], [base], [0.7], base)
    # K = 1 returns base + (ft - base), never scaled by alpha (quirk Q6)
    assert torch.equal(out, (base.float() + (ft.float() - base.float())).to(torch.bfloat16))


def test_full_size_arith_closed_form(engine, full_inputs):
    """ft_b == base: the second delta is exactly zero, the Arithmetic-FFT branch runs and
    (quirk Q3: disagreeing bins take Re F_b = 0) leaves only i*Im F_a, i.e. the odd part
    of a, scaled by target_norm/||a|| = 1/2.  A closed form that exercises both forward
    passes, the blend and both inverse passes at full size."""
    base, d0, _ = full_inputs
    z = torch.zeros_like(d0)
    out, rep, delta = engine.merge_layer([d0, z], [z, z], [0.3, 0.5], z, want_delta=True)
    assert rep.branches == ["arith"]
    flipped = torch.roll(torch.flip(d0, dims=(0, 1)), shifts=(1, 1), dims=(0, 1))   # a[-r, -c]
    expect = (d0 - flipped) * (0.5 * rep.target_norm / (0.5 * rep.delta_norms[0]) * 0.5)
    assert so.rel_err(delta.cpu(), expect.cpu()) < 5e-6


def test_full_size_scale_equivariance_and_determinism(engine, full_inputs):
    """Scaling both deltas by a power of two scales the merged delta by exactly that
    factor (every stage is homogeneous of degree one or scale-free); and two runs are
    bit-identical."""
    _, d0, d1 = full_inputs
    z = torch.zeros_like(d0)
    out1, rep1, delta1 = engine.merge_layer([d0, d1], [z, z], [0.3, 0.5], z, want_delta=True)
    out2, rep2, delta2 = engine.merge_layer([d0, d1], [z, z], [0.3, 0.5], z, want_delta=True)
    assert rep1.branches == ["slerp"]
    assert torch.equal(delta1, delta2) and torch.equal(out1, out2)
    out4, rep4, delta4 = engine.merge_layer([d0 * 4, d1 * 4], [z, z], [0.3, 0.5], z, want_delta=True)
    assert so.rel_err(delta4.cpu(), (delta1 * 4).cpu()) < 1e-6
    # class bookkeeping: every bin is in exactly one class, 8% / 20% order statistics
    info = rep1.infos[0]
    assert info.cutoff_threshold > 0 and info.cull_threshold > info.cutoff_threshold
    assert 0 < info.n_slerp < N_FULL * N_FULL
    assert abs(rep1.merged_delta_norm / rep1.target_norm - 1) < 0.2


@pytest.mark.parametrize("shape", [(8192, 8192), (28672, 4096)], ids=["8192sq", "70B_mlp_half"])
def test_large_tensors_take_the_fast_selection_path(engine, shape):
    """The candidate lists of the selection passes must not overflow on ordinary data at
    BASELINE's sizes (an overflow redoes the whole layer with full histogram passes: correct,
    but twice the time - it happened silently at the Llama-3-70B shapes)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    base = (torch.randn(shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(shape, generator=g, device="cuda") * s).to(torch.bfloat16) for s in (0.002, 0.003)]
    engine.ctx.profile(True)
    engine.ctx.profile_reset()
    try:
        out, rep = engine.merge_layer(fts, [base, base], [0.3, 0.5], base)
        table = engine.ctx.profile_table()
    finally:
        engine.ctx.profile(False)
    assert rep.branches == ["slerp"]
    assert table["f1_rows_fwd"][0] == 1 and table["i2_rows_inv"][0] == 1        # one pass, no retry
    assert "select_hist" not in table and "slerp_reduce" not in table          # no safe-mode kernels
    # exact order statistics at full size, checked against torch on the device: the cull
    # threshold is the 20 % quantile position of |Re R| over the full spectrum - here only its
    # sanity (the parity tests check values at sizes the oracle can do)
    info = rep.infos[0]
    assert info.cull_threshold > info.cutoff_threshold > 0
    del out, fts, base
    torch.cuda.empty_cache()


def test_midstream_candidate_flush_on_device(engine):
    """Same layer with the selection pass forced into many rounds and a flush after each
    (what a 235 M-element Llama-3-70B tensor does by itself): identical thresholds, class
    counts and output."""
    g = torch.Generator(device="cuda").manual_seed(9)
    shape = (4096, 2048)
    base = (torch.randn(shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(shape, generator=g, device="cuda") * s).to(torch.bfloat16) for s in (0.002, 0.003)]
    out0, rep0 = engine.merge_layer(fts, [base, base], [0.3, 0.5], base)
    engine.ctx.debug_option("sel_chunks", 96)
    engine.ctx.debug_option("sel_flush_always", 1)
    try:
        out1, rep1 = engine.merge_layer(fts, [base, base], [0.3, 0.5], base)
    finally:
        engine.ctx.debug_option("sel_chunks", 0)
        engine.ctx.debug_option("sel_flush_always", 0)
    i0, i1 = rep0.infos[0], rep1.infos[0]
    assert i0.cutoff_threshold == i1.cutoff_threshold and i0.cull_threshold == i1.cull_threshold
    assert i0.n_slerp == i1.n_slerp
    assert (out0.float() - out1.float()).abs().max().item() <= 2 ** -7 * out0.float().abs().max().item()   # at most a bf16 ulp
    assert (out0 != out1).float().mean().item() < 1e-3


def test_cli_end_to_end_on_device(tmp_path, golden):
    """`python -m shard merge CONFIG` (device: cuda) on the tiny on-disk model of G8:
    files, index, README and tensors against the reference CLI's output."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    from safetensors import safe_open
    repo = Path(__file__).resolve().parents[1]
    cfg = gi.write_cli_model(tmp_path, device="cuda")
    res = subprocess.run([sys.executable, "-m", "shard", "merge", str(cfg), "--device", "cuda"], cwd=str(repo),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    out_dir = tmp_path / "merged"
    want = golden.manifest["cli"]
    assert sorted(p.name for p in out_dir.iterdir()) == want["files"]
    assert json.load(open(out_dir / "model.safetensors.index.json")) == want["index"]
    assert (out_dir / "README.md").read_text() == want["readme"]
    for shard in gi.CLI_SHARDS:
        with safe_open(str(out_dir / shard), framework="pt") as f:
            for k in f.keys():
                ref = golden.get("g8_cli.safetensors", f"{shard}::{k}")
                got = f.get_tensor(k)
                assert got.dtype == ref.dtype and got.shape == ref.shape
                if "layers" in k:
                    assert so.rel_err(got.float(), ref.float()) < 2e-3
                else:
                    assert torch.equal(got, ref)


# ---- policy and edge paths ON THE DEVICE (the emulator tier runs the same checks with g++;
# here they go through the gfx950 code: real atomics for the sticky overflow word, the
# and+max NaN screen with v_cvt_pk_bf16_f32, the run-time planned / element-wise kernels) -----
from tests import test_emul_parity as emul_tier      # noqa: E402  (functions take the engine as an argument)

_SHARED = [emul_tier.test_inf_raises_like_reference, emul_tier.test_nan_is_zeroed, emul_tier.test_heavy_ties,
           emul_tier.test_unsupported_length_is_loud, emul_tier.test_non_finite_delta_norm_is_an_error_not_a_hang,
           emul_tier.test_mixed_input_dtypes_are_promoted_not_demoted, emul_tier.test_nan_inf_policy_in_the_inverse_row_pass,
           emul_tier.test_norm_mode_reference_cpu_reproduces_torch_norm,
           emul_tier.test_reference_cpu_norm_is_torch_norm_bit_for_bit,
           emul_tier.test_reference_cpu_mode_follows_the_biased_norms_everywhere,
           emul_tier.test_reference_cpu_mode_keeps_k3_intermediates_spectral]


@pytest.mark.parametrize("check", _SHARED, ids=lambda f: f.__name__[5:])
def test_policy_on_device(engine, check):
    check(engine)


@pytest.mark.parametrize("k,shape,fold", [(2, (64, 2048), False), (3, (24, 8192), False), (3, (16, 16384), False), (2, (16, 28672), False),
                                         (3, (8192, 2048), True), (3, (14336, 4096), True), (2, (28672, 8192), True)],
                         ids=lambda v: str(v).replace(" ", ""))
def test_row_pass_summarises_its_deltas_for_the_norm_emulation_on_device(engine, k, shape, fold):
    """the fused torch.norm summaries of k_f1 / k_f1q against torch.norm and the separate pass, on the device - incl. the
    metric's shapes (KF1Q<8192> at 28672 x 8192, KF1<28672>, KF1Q<4096>)"""
    emul_tier.test_row_pass_summarises_its_deltas_for_the_norm_emulation(engine, k, shape, fold)


def test_candidate_list_overflow_falls_back_on_device(engine, golden):
    emul_tier.test_candidate_list_overflow_falls_back(engine, golden)


@pytest.mark.parametrize("cid", ["layer_k3", "layer_k3_swap", "layer_k4"])
def test_k3_steps_and_unculled_spectrum_on_device(engine, cid):
    emul_tier.test_k3_steps_and_unculled_spectrum_match_the_oracle(engine, cid)


@pytest.mark.parametrize("cid,mutation", [("layer_k3", "keep_cull_pct"), ("layer_k4", "keep_cull_pct"),
                                          ("layer_k3", "sum_weights"), ("layer_k3_swap", "swap_weights")])
def test_k3_checks_catch_a_wrong_round2_on_device(engine, cid, mutation):
    emul_tier.test_k3_checks_catch_a_wrong_round2(engine, cid, mutation)


def test_misaligned_device_inputs_take_the_elementwise_path(engine):
    """A device view that is not 16-byte aligned must run the run-time planned, element-wise
    kernels (`vec = 0`) and give the same bits as the aligned copy."""
    g = torch.Generator().manual_seed(9)
    big = (torch.randn(3 + 64 * 1024, generator=g) * 0.01).to(torch.bfloat16).cuda()
    base = big[3:3 + 32 * 1024].view(32, 1024)              # 6-byte offset into the allocation
    assert base.data_ptr() % 16 != 0 and base.is_contiguous()
    ft0 = (base.float() + torch.randn(32, 1024, generator=g).cuda() * 0.002).to(torch.bfloat16)
    ft1 = (base.float() + torch.randn(32, 1024, generator=g).cuda() * 0.003).to(torch.bfloat16)
    for mode, bar in (("exact", 1e-5), ("reference_cpu", 5e-5)):
        out_u, rep_u = engine.merge_layer([ft0, ft1], [base, base], [0.3, 0.5], base, norm_mode=mode)
        out_a, rep_a = engine.merge_layer([ft0, ft1], [base.clone(), base.clone()], [0.3, 0.5], base.clone(), norm_mode=mode)
        assert rep_u.branches == rep_a.branches == ["slerp"]
        # the torch.norm emulation takes unaligned inputs element by element: the same bits, never other numerics
        assert rep_u.delta_norms == rep_a.delta_norms
        if mode == "reference_cpu":
            assert rep_u.delta_norms == [float((f.float().cpu() - base.float().cpu()).norm()) for f in (ft0, ft1)]
        assert (out_u != out_a).float().mean().item() < 1e-3     # two plans (static / run-time): ulps apart
        assert so.rel_err(out_u.float().cpu(), out_a.float().cpu()) < bar, mode


# ---- the BASELINE configs' REAL tensor shapes against the oracle (slow: the oracle's two sorts) ----
def _oracle_threads():
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8
    torch.set_num_threads(max(1, min(16, n)))       # the sort-bound oracle is slower on hundreds of threads


@pytest.mark.parametrize("shape", [(4096, 4096), (1024, 4096), (14336, 4096), (4096, 14336), (1024, 8192)],
                         ids=lambda s: f"{s[0]}x{s[1]}")
def test_real_model_shapes_k2_vs_exact_norm_oracle(engine, shape):
    """Every 2-D block-tensor shape of Llama-3-8B / Mixtral (BASELINE configs 2, 3, 5) and the
    1024 x 8192 of Llama-3-70B at FULL size: K = 2 merge against the oracle with exact norms."""
    _oracle_threads()
    rows, cols = shape
    base, fts = so.synthetic_layer(rows, cols, 2, seed=900 + rows + cols)
    trx = so.LayerTrace()
    with so.exact_norms(), so.fast_select():
        refx = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=trx)
    out, rep, delta = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True, norm_mode="exact")
    pc.check_layer_steps(rep, trx, out.numel())
    # the number of bins that sit ON a threshold (within the ~1e-7 by which two correct FFTs differ)
    # grows with the tensor: 58 M elements have ~10-30 of them (tools/fold_check.py) - beyond those
    # the agreement is 4e-7
    d_total, d_resid = pc.spectral_residual(delta.cpu(), trx.merged_delta, drop=64)
    assert d_resid < 5e-6, f"beyond the tie bins: {d_resid:.2e}"
    assert d_total < 1e-3, f"merged delta (SURVEY 8d's stricter bar): {d_total:.2e}"
    assert so.rel_err(out.cpu().float(), refx.float()) < 1e-3          # BASELINE: 1e-3 on the bf16 output
    # a tie bin that flips moves every output by ~thr/n: values within that of a bf16 rounding
    # boundary (ulp 2^-8 relative) change by one ulp - a few per cent of them at these sizes
    mism = (out.cpu().view(torch.int16) != refx.view(torch.int16)).float().mean().item()
    assert mism < 0.08, f"{mism:.3%} of the bf16 outputs differ"


def _layer_ms(engine, fts, bases, alphas, base, norm_mode, reps=5):
    """wall time per merge_layer call, single stream, workspace warm"""
    import time
    engine.merge_layer(fts, bases, alphas, base, norm_mode=norm_mode)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        engine.merge_layer(fts, bases, alphas, base, norm_mode=norm_mode)
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3


def test_fullsize_8192sq_vs_reference_as_is(engine):
    """One [8192 x 8192] K = 2 merge (the north-star micro-benchmark shape) against the oracle
    EXACTLY as the reference is - torch's CPU norm kernel included, which is biased by -5e-3 at
    67 M elements (oracle/norm_bias_probe.py) - in norm_mode = reference_cpu (the mode bench.py
    measures): BASELINE's 1e-3 on the bf16 output AND SURVEY 8(d)'s 1e-3 on the merged delta.
    Also recorded: the accurate-norm mode against both oracles, and what the mode costs.
    (gpurun_out/parity_fullsize.json -> profiles/)."""
    import json
    import os
    import time
    _oracle_threads()
    base, fts = so.synthetic_layer(8192, 8192, 2, seed=1000)
    t0 = time.time()
    tr, trx = so.LayerTrace(), so.LayerTrace()
    with so.fast_select():
        ref = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=tr)
        with so.exact_norms():
            refx = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=trx)
    t_oracle = time.time() - t0
    out, rep, delta = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True, norm_mode="exact")
    out, delta = out.cpu(), delta.cpu()
    out_r, rep_r, delta_r = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True, norm_mode="reference_cpu")
    out_r, delta_r = out_r.cpu(), delta_r.cpu()
    dev = [t.cuda() for t in fts], base.cuda()
    ms_exact = _layer_ms(engine, dev[0], [dev[1]] * 2, so.ALPHAS[:2], dev[1], "exact")
    ms_ref = _layer_ms(engine, dev[0], [dev[1]] * 2, so.ALPHAS[:2], dev[1], "reference_cpu")
    rec = {
        "shape": [8192, 8192], "k": 2, "seed": 1000, "oracle_seconds_both_modes": round(t_oracle, 1),
        "torch_threads": torch.get_num_threads(),
        "as_is_with_norm_mode_reference_cpu": {
            "out_rel_err": so.rel_err(out_r.float(), ref.float()), "delta_rel_err": so.rel_err(delta_r, tr.merged_delta),
            "delta_beyond_64_tie_bins": pc.spectral_residual(delta_r, tr.merged_delta, drop=64)[1],
            "target_norm_hip": rep_r.target_norm, "target_norm_ref": tr.target_norm, "delta_norms_hip": rep_r.delta_norms,
            "cosine_hip": rep_r.infos[0].dot, "cosine_ref": tr.steps[0].dot, "cosine_exact_norm_oracle": trx.steps[0].dot},
        "as_is_with_exact_norms": {"out_rel_err": so.rel_err(out.float(), ref.float()), "delta_rel_err": so.rel_err(delta, tr.merged_delta),
                                   "target_norm_hip": rep.target_norm},
        "exact_norm_oracle_vs_exact_norms": {"out_rel_err": so.rel_err(out.float(), refx.float()),
                                             "delta_rel_err": so.rel_err(delta, trx.merged_delta), "target_norm_ref": trx.target_norm},
        "reference_self_shift": {"out": so.rel_err(ref.float(), refx.float()),
                                 "delta": so.rel_err(tr.merged_delta, trx.merged_delta)},
        "layer_ms_single_stream": {"exact": round(ms_exact, 3), "reference_cpu": round(ms_ref, 3), "ratio": round(ms_ref / ms_exact, 3)},
        "steps_hip_reference_cpu": [vars(i) for i in rep_r.infos],
        "steps_as_is": [{k: v for k, v in vars(b).items() if k != "culled_mask"} for b in tr.steps if b is not None],
        "steps_exact": [{k: v for k, v in vars(b).items() if k != "culled_mask"} for b in trx.steps if b is not None],
    }
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_fullsize.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({k: rec[k] for k in ("as_is_with_norm_mode_reference_cpu", "as_is_with_exact_norms",
                                          "exact_norm_oracle_vs_exact_norms", "reference_self_shift", "layer_ms_single_stream")}))
    # accurate norms against the accurate-norm oracle
    pc.check_layer_steps(rep, trx, out.numel())
    assert rec["exact_norm_oracle_vs_exact_norms"]["delta_rel_err"] < 1e-3 and rec["exact_norm_oracle_vs_exact_norms"]["out_rel_err"] < 1e-3
    # norm_mode = reference_cpu against the reference's device="cpu" output AS IT IS
    pc.check_layer_steps(rep_r, tr, out.numel())
    assert rep_r.target_norm == tr.target_norm
    assert rep_r.delta_norms == [float((f.float() - base.float()).norm()) for f in fts]       # torch.norm's values, bit for bit
    assert abs(rep_r.infos[0].dot - tr.steps[0].dot) < 0.05 * abs(tr.steps[0].dot - trx.steps[0].dot)
    assert rec["as_is_with_norm_mode_reference_cpu"]["out_rel_err"] < 1e-3
    assert rec["as_is_with_norm_mode_reference_cpu"]["delta_rel_err"] < 1e-3
    assert rec["as_is_with_norm_mode_reference_cpu"]["delta_beyond_64_tie_bins"] < 2e-5
    # what the conforming mode costs (single stream; bench.py's 8 streams hide the walker's latency)
    assert ms_ref <= 1.4 * ms_exact, f"reference_cpu {ms_ref:.2f} ms vs exact {ms_exact:.2f} ms per layer"


def _fullsize_as_is(engine, rows, cols, k, seed, tag, masked=True):
    """One layer of a BASELINE shape at FULL size in norm_mode = reference_cpu (the mode bench.py measures) against
    the oracle AS THE REFERENCE IS (torch's CPU norms included; order statistics by selection instead of a full
    sort: so.fast_select(), pinned equal by tests/test_oracle_golden.py).  Records gpurun_out/<tag>.json
    (-> profiles/) and returns the record."""
    import json
    import os
    import time
    _oracle_threads()
    base, fts = so.synthetic_layer(rows, cols, k, seed=seed)
    t0 = time.time()
    tr = so.LayerTrace()
    with so.fast_select():
        ref = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=tr)
    t_oracle = time.time() - t0
    out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True, norm_mode="reference_cpu")
    out, delta = out.cpu(), delta.cpu()
    assert rep.delta_norms == [float((f.float() - base.float()).norm()) for f in fts]        # torch.norm's values, bit for bit
    assert rep.target_norm == tr.target_norm
    pc.check_layer_steps(rep, tr, out.numel())
    rec = {"shape": [rows, cols], "k": k, "seed": seed, "norm_mode": "reference_cpu", "oracle": "as the reference is (torch CPU norms)",
           "oracle_seconds": round(t_oracle, 1), "torch_threads": torch.get_num_threads(),
           "out_rel_err": so.rel_err(out.float(), ref.float()), "delta_rel_err": so.rel_err(delta, tr.merged_delta),
           "bf16_outputs_that_differ": (out.view(torch.int16) != ref.view(torch.int16)).float().mean().item(),
           "target_norm": rep.target_norm, "delta_norms": rep.delta_norms, "branches": rep.branches,
           "steps_hip": [vars(i) for i in rep.infos],
           "steps_as_is": [({kk: v for kk, v in vars(b).items() if kk != "culled_mask"} if b is not None else None) for b in tr.steps]}
    if k == 2:
        for drop in (64, 256):
            rec[f"delta_beyond_{drop}_tie_bins"] = pc.spectral_residual(delta, tr.merged_delta, drop=drop)[1]
    elif masked:
        outside, inside, flips = pc.masked_spectral_check(delta, tr, device="cuda" if masked == "device" else "cpu")
        rec.update({"delta_outside_culled_bins": outside, "delta_inside_culled_bins": inside, "final_cull_flips": flips})
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, tag + ".json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({kk: rec[kk] for kk in rec if not kk.startswith("steps")}))
    return rec


def test_llama3_70b_mlp_shape_fullsize_k2_as_is(engine):
    """[28672 x 8192] - the tensor shape that holds 82 % of a Llama-3-70B block's parameters (folded column pass,
    speculative cull selection), K = 2, against the reference as it is: BASELINE's 1e-3 on the bf16 output and
    SURVEY 8(d)'s 1e-3 on the merged delta.  235 M elements have 64 ... 256 bins within rounding of a threshold
    (DESIGN 6.1): the residual is taken beyond the 256 largest bins of the difference's spectrum."""
    rec = _fullsize_as_is(engine, 28672, 8192, 2, 4242, "parity_fullsize_70b_mlp")
    assert rec["delta_beyond_256_tie_bins"] < 2e-5, rec
    assert rec["delta_rel_err"] < 1e-3 and rec["out_rel_err"] < 1e-3 and rec["bf16_outputs_that_differ"] < 0.08, rec


def test_llama3_70b_mlp_down_shape_fullsize_k2_as_is(engine):
    """[8192 x 28672] (down_proj: the 28672-point ROW plans), K = 2, against the reference as it is."""
    rec = _fullsize_as_is(engine, 8192, 28672, 2, 4343, "parity_fullsize_70b_mlp_down")
    assert rec["delta_beyond_256_tie_bins"] < 2e-5, rec
    assert rec["delta_rel_err"] < 1e-3 and rec["out_rel_err"] < 1e-3 and rec["bf16_outputs_that_differ"] < 0.08, rec


def test_8192sq_fullsize_k3_as_is(engine):
    """[8192 x 8192], K = 3 (the metric's K): spectral intermediate, modelled norm of the intermediate, two rounds.
    Round 2 decides on rounding noise in the reference itself (it moves by 1.6e-3 on the bf16 output
    when its own FFT is evaluated in fp64): everything reproducible is compared step by step and outside the bins
    round 1 culled; the total is held to 3x that floor."""
    rec = _fullsize_as_is(engine, 8192, 8192, 3, 5151, "parity_fullsize_8192sq_k3")
    assert rec["branches"].count("slerp") == 2
    assert rec["out_rel_err"] < 2e-3, rec          # (the reference-held floor at 1024^2: 1.47e-3, g11_floor)
    assert rec["delta_outside_culled_bins"] < 1e-3, rec


def test_llama3_8b_mlp_shape_fullsize_k4_as_is(engine):
    """[14336 x 4096] (Llama-3-8B / Mixtral MLP), K = 4 - BASELINE config 3: three pair merges, the last one of two
    intermediates.  The reference's own K = 4 floor is 3.1e-2 on the bf16 output."""
    rec = _fullsize_as_is(engine, 14336, 4096, 4, 6161, "parity_fullsize_8b_mlp_k4")
    assert rec["branches"].count("slerp") == 3
    assert rec["out_rel_err"] < 4e-2, rec          # (the reference-held K = 4 floor: 3.06e-2, g11_floor)
    assert rec["delta_outside_culled_bins"] < pc.OUTSIDE_CULLED_TOL, rec     # (measured 1.7e-3: two rounds of culled bins feed the last merge)


def test_llama3_70b_mlp_shape_fullsize_k3_as_is(engine):
    """[28672 x 8192], K = 3 - the metric's configuration on its dominant tensor: steps checked, totals recorded."""
    rec = _fullsize_as_is(engine, 28672, 8192, 3, 7171, "parity_fullsize_70b_mlp_k3", masked="device")
    assert rec["branches"].count("slerp") == 2
    assert rec["out_rel_err"] < 2e-3, rec
    assert rec["delta_outside_culled_bins"] < pc.OUTSIDE_CULLED_TOL, rec        # (measured 1.5e-3, see the down_proj shape)


# ---- K >= 3 against reference-held floor data, on the device ---------------------------------------
@pytest.mark.parametrize("norm_mode", ["reference_cpu", "exact"])
@pytest.mark.parametrize("case", gi.FLOOR_CASES, ids=lambda c: c["id"])
def test_k3_k4_within_the_reference_own_floor_on_device(engine, golden, case, norm_mode):
    rec = pc.check_floor(engine, golden, case, norm_mode=norm_mode)
    print(rec)


def test_noise_model_seed_does_not_matter_beyond_the_floor_on_device(engine, golden):
    emul_tier.test_noise_model_seed_does_not_matter_beyond_the_floor(engine, golden)


def _k3_bar(n):
    """the K = 3 bar on the bf16 output: 2e-3 (the reference-held floor is 1.47e-3, tests/golden/manifest.json) once the
    threshold-tie floor 8 / sqrt(n) is below it"""
    return max(2e-3, 8.0 / math.sqrt(n))


def test_llama3_70b_mlp_down_shape_fullsize_k3_as_is(engine):
    """[8192 x 28672], K = 3: the 28672-point ROW plans (KF1<28672>, KI2<28672>) in the metric's configuration."""
    rec = _fullsize_as_is(engine, 8192, 28672, 3, 7272, "parity_fullsize_70b_mlp_down_k3", masked="device")
    assert rec["branches"].count("slerp") == 2
    assert rec["out_rel_err"] < 2e-3, rec
    # (measured 2.2e-3: at 235 M elements torch's norms are biased by -2e-2 and the spectral intermediate's is MODELLED to
    #  ~1e-5, which moves the second round's larger-of decisions by ~sqrt of that; 8192^2: 3.3e-4)
    assert rec["delta_outside_culled_bins"] < pc.OUTSIDE_CULLED_TOL, rec


def test_llama3_70b_kv_proj_shape_k3_as_is(engine):
    """[1024 x 8192] (k_proj / v_proj of Llama-3-70B), K = 3."""
    rec = _fullsize_as_is(engine, 1024, 8192, 3, 7373, "parity_fullsize_70b_kv_k3")
    assert rec["branches"].count("slerp") == 2
    assert rec["out_rel_err"] < _k3_bar(1024 * 8192), rec
    assert rec["delta_outside_culled_bins"] < 1e-3, rec


def test_llama3_70b_norm_weight_1d_k3_as_is(engine):
    """[8192] (input_layernorm / post_attention_layernorm weights), K = 3 through the one-launch 1-D pair merge."""
    rec = _fullsize_as_is(engine, 0, 8192, 3, 7474, "parity_fullsize_70b_norm1d_k3", masked=False)
    assert rec["branches"].count("slerp") == 2
    assert rec["out_rel_err"] < _k3_bar(8192), rec


# ---- N3 / N4 on the device ------------------------------------------------------------------------
@pytest.mark.parametrize("case", gi.ADDITION_CASES, ids=lambda c: c["id"])
def test_addition_operators_bit_exact_on_device(engine, golden, case):
    emul_tier.test_addition_operators_bit_exact(engine, golden, case)


def test_addition_known_answers_and_linear_branch_on_device(engine):
    emul_tier.test_addition_known_answers_of_the_reference_tests(engine)
    emul_tier.test_linear_blend_branch_with_a_larger_b(engine)


def test_addition_full_size_against_torch_on_device(engine):
    """8192 x 8192 bf16, K = 3: the kernel against the reference's formula evaluated by torch on
    the same device tensors (elementwise bf16 ops round like the CPU's: bit-exact expected)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    base = (torch.randn(8192, 8192, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(8192, 8192, generator=g, device="cuda") * 0.01).to(torch.bfloat16) for _ in range(3)]
    out = engine.addition_merge(fts, base)
    ref = torch.zeros_like(base)
    for ft in fts:
        ref += ft - base
    assert torch.equal(out, ref)
    out_t = engine.addition_merge(fts, base, sign_agreement=True)
    stack = torch.stack([ft - base for ft in fts], dim=0)
    signs = torch.sign(stack)
    keep = signs == torch.sum(signs, dim=0).sign().unsqueeze(0)
    assert torch.equal(out_t, torch.sum(stack * keep, dim=0))


# ---- the folded column pass on the device ------------------------------------------------------------
@pytest.mark.parametrize("k", [2, 3])
def test_folded_column_pass_on_device(engine, k):
    emul_tier.test_folded_column_pass(engine, k)


def test_folded_column_pass_is_taken_for_the_mlp_shapes(engine):
    """28672 x 8192 / 14336 x 4096 (Llama-3 MLP tensors) run the folded kernels by default; the
    result must equal the plain path's up to threshold-tie bins (same data, same thresholds)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    shape = (14336, 4096)
    base = (torch.randn(shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(shape, generator=g, device="cuda") * s).to(torch.bfloat16) for s in (0.002, 0.003)]
    # exact norms: same data, same thresholds.  reference_cpu: the norms of the gathered slerp class are MODELLED from a
    # sample of the planes (k_class_emf), and the folded planes hold each bin column in another order - another sample:
    # the cosine moves by ~3e-5, the cull threshold (10 % quantile of the blended values) by ~2e-6 with it
    for mode, thr_bar, res_bar in (("exact", 2e-6, 2e-6), ("reference_cpu", 1e-5, 1e-4)):
        out1, rep1, d1 = engine.merge_layer(fts, [base, base], [0.3, 0.5], base, want_delta=True, norm_mode=mode)
        engine.ctx.debug_option("fold_columns", 0)
        try:
            out0, rep0, d0 = engine.merge_layer(fts, [base, base], [0.3, 0.5], base, want_delta=True, norm_mode=mode)
        finally:
            engine.ctx.debug_option("fold_columns", 1)
        i1, i0 = rep1.infos[0], rep0.infos[0]
        assert rep1.delta_norms == rep0.delta_norms
        assert abs(i1.cutoff_threshold - i0.cutoff_threshold) <= 2e-6 * i0.cutoff_threshold, mode
        assert abs(i1.cull_threshold - i0.cull_threshold) <= thr_bar * i0.cull_threshold, mode
        assert abs(i1.dot - i0.dot) <= (1e-6 if mode == "exact" else 6e-5), mode
        assert abs(i1.n_slerp - i0.n_slerp) <= 64
        assert not torch.equal(d1, d0)                    # two different kernel paths did run
        assert pc.spectral_residual(d1.cpu(), d0.cpu(), drop=64)[1] < res_bar, mode


def test_bench_product_path_rehearsal_two_ranks_on_one_card(tmp_path):
    """bench.py --gpus 2 --product-path: the multi-GPU PRODUCT loop (plan, base-shard broadcasts a window ahead,
    prefetching loader, in-place output shards) measured end to end - rehearsed here with 2 gloo ranks sharing the
    box's one GPU, so that the first 8-GPU run over RCCL measures distributed.py and not a list of resident tensors."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parents[1]
    env = dict(os.environ, SHARDMERGE_BENCH_BACKEND="gloo", SHARDMERGE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--product-path", "--workload", "llama3-8b", "--blocks", "2",
                          "--k", "2", "--steps", "1", "--warmup", "1", "--product-root", str(tmp_path / "model")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert len(rec["per_rank"]) == 2 and all(r["tensors"] > 0 for r in rec["per_rank"])
    assert sum(r["tensors"] for r in rec["per_rank"]) == 2 * 9 + 3
    print(line)


def test_partitioned_merge_through_rccl_world_size_1(tmp_path, golden):
    """The multi-GPU product path (shardmerge_amd/distributed.py) with its collectives REAL: one
    rank, backend nccl (= RCCL), process group initialised on the device, the base-shard
    broadcast and the barriers executed.  (A one-GPU box cannot host two RCCL ranks; the N > 1
    logic is covered by the 2-rank gloo test on CPU.)  Output must equal the single-process CLI's."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path
    from safetensors import safe_open
    repo = Path(__file__).resolve().parents[1]
    cfg = gi.write_cli_model(tmp_path, device="cuda")
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SHARDMERGE_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="VERSION")
    res = subprocess.run([sys.executable, "-m", "shard", "merge", str(cfg), "--device", "cuda"], cwd=str(repo), env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "rank 0/1" in res.stderr + res.stdout                     # the partitioned path did run
    out_dir = tmp_path / "merged"
    want = golden.manifest["cli"]
    assert sorted(p.name for p in out_dir.iterdir()) == want["files"]
    assert json.load(open(out_dir / "model.safetensors.index.json")) == want["index"]
    for shard in gi.CLI_SHARDS:
        with safe_open(str(out_dir / shard), framework="pt") as f:
            for k in f.keys():
                ref = golden.get("g8_cli.safetensors", f"{shard}::{k}")
                got = f.get_tensor(k)
                assert got.dtype == ref.dtype and got.shape == ref.shape
                if "layers" in k:
                    assert so.rel_err(got.float(), ref.float()) < 2e-3
                else:
                    assert torch.equal(got, ref)


@pytest.mark.parametrize("k", [2, 3])
def test_rank3_tensor_on_device(engine, k):
    emul_tier.test_rank3_tensor_is_a_batch_of_transforms_with_global_statistics(engine, k)


@pytest.mark.parametrize("case", gi.CORR_CASES, ids=lambda c: c["id"])
def test_correlate_pairs_on_device(engine, golden, case):
    emul_tier.test_correlate_pairs(engine, golden, case)


# ---- column / row lengths without a work-group plan (k_dftp, transposed merge) ------------------------
@pytest.mark.parametrize("shape", [(34, 64), (136, 96), (76, 96), (172, 128), (272, 64), (668, 32)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("k", [2, 3])
def test_split_column_length_on_device(engine, shape, k):
    emul_tier.test_column_length_without_a_plan_is_split_into_row_blocks(engine, shape, k)


@pytest.mark.parametrize("p", [2, 4, 8])
@pytest.mark.parametrize("k", [2, 3])
def test_forced_split_on_device(engine, p, k):
    emul_tier.test_forced_split_agrees_with_the_plain_column_pass(engine, p, k)


@pytest.mark.parametrize("shape", [(64, 34), (96, 76), (1, 68), (128, 2 * 43)], ids=lambda s: "x".join(map(str, s)))
def test_transposed_merge_on_device(engine, shape):
    emul_tier.test_row_length_without_a_plan_is_merged_transposed(engine, shape)


@pytest.mark.parametrize("shape", [(3, 64, 128), (172, 128), (64, 86)], ids=lambda s: "x".join(map(str, s)))
def test_arith_and_linear_branches_on_sliced_geometries_on_device(engine, shape):
    emul_tier.test_arith_and_linear_branches_on_sliced_geometries(engine, shape)


def test_unsupported_shapes_are_refused_on_device(engine):
    emul_tier.test_shapes_no_orientation_can_take_are_refused(engine)


@pytest.mark.parametrize("shape", [(11008, 4096), (4096, 11008), (18944, 3584)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_llama2_and_qwen2_mlp_shapes_k2_vs_exact_norm_oracle(engine, shape):
    """The MLP tensors of Llama-2-7B (11008 = 43 * 256) and Qwen2-7B (18944 = 37 * 512) at FULL size,
    the 4096 x 11008 down-projection through the transposed path: K = 2 against the exact-norm oracle
    (run in both orientations for the transposed shape, see the emulator tier's test)."""
    _oracle_threads()
    rows, cols = shape
    base, fts = so.synthetic_layer(rows, cols, 2, seed=77 + rows + cols)
    out, rep, delta = engine.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True, norm_mode="exact")
    out, delta = out.cpu(), delta.cpu()
    flips = (False, True) if not engine.lib.length_supported(cols) else (False,)
    best = None
    for flip in flips:
        tt = (lambda x: x.T.contiguous()) if flip else (lambda x: x)
        trx = so.LayerTrace()
        with so.exact_norms():
            refx = so.merge_layer([tt(f) for f in fts], [tt(base)] * 2, so.ALPHAS[:2], tt(base), trace=trx)
        pc.check_layer_steps(rep, trx, out.numel())
        d_total, d_resid = pc.spectral_residual(tt(delta), trx.merged_delta, drop=64)
        o_err = so.rel_err(tt(out).float(), refx.float())
        mism = (tt(out).view(torch.int16) != refx.view(torch.int16)).float().mean().item()
        assert d_total < 1e-3 and o_err < 1e-3 and mism < 0.08
        best = d_resid if best is None else min(best, d_resid)
    assert best < 5e-6, f"beyond the tie bins: {best:.2e}"


@pytest.mark.parametrize("shape", [(64, 96), (32, 256), (128, 40)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("k", [2, 3])
def test_chirp_z_row_passes_on_device(engine, shape, k):
    emul_tier.test_chirp_z_row_passes_agree_with_the_planned_ones(engine, shape, k)


@pytest.mark.parametrize("shape", [(34, 38), (68, 76), (142, 142), (172, 86), (34, 17), (17, 32), (17, 128)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("k", [2, 3])
def test_both_lengths_without_a_plan_on_device(engine, shape, k):
    emul_tier.test_both_lengths_without_a_plan(engine, shape, k)


@pytest.mark.parametrize("shape", [(64, 38), (32, 17), (1, 38), (128, 142)], ids=lambda s: "x".join(map(str, s)))
def test_function_level_transforms_on_a_rough_row_length_on_device(engine, shape):
    emul_tier.test_function_level_transforms_on_a_row_length_without_a_plan(engine, shape)


@pytest.mark.parametrize("shape,k", [((4544, 4544), 2), ((4672, 4544), 2), ((4544, 4544), 3)], ids=["4544sq_k2", "4672x4544_k2", "4544sq_k3"])
def test_falcon_7b_shapes_vs_exact_norm_oracle(engine, shape, k):
    """Falcon-7B's attention tensors at FULL size: 4544 = 71 * 64 and 4672 = 73 * 64, no planned length on either
    axis - the column length split into row blocks (k_dftp), the rows through the chirp-z row passes (k_f1b / k_i2b).
    Same bar as the other full-size shapes."""
    _oracle_threads()
    rows, cols = shape
    assert engine.lib.shape_supported(rows, cols) and not engine.lib.length_supported(rows) and not engine.lib.length_supported(cols)
    base, fts = so.synthetic_layer(rows, cols, k, seed=91 + rows + cols)
    out, rep, delta = engine.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True, norm_mode="exact")
    out, delta = out.cpu(), delta.cpu()
    tr = so.LayerTrace()
    with so.exact_norms(), so.fast_select():
        ref = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=tr)
    pc.check_layer_steps(rep, tr, out.numel())
    if k == 2:
        d_total, d_resid = pc.spectral_residual(delta, tr.merged_delta, drop=64)
        assert d_total < 1e-3 and so.rel_err(out.float(), ref.float()) < 1e-3
        assert d_resid < 5e-6, f"beyond the tie bins: {d_resid:.2e}"
    else:
        assert so.rel_err(out.float(), ref.float()) < 5e-3 and so.rel_err(delta, tr.merged_delta) < 8e-2


def test_generic_dft_kernel_on_device(engine):
    emul_tier.test_generic_dft_kernel_agrees_with_the_paired_one(engine)


@pytest.mark.parametrize("k,shape", [(2, (128, 256)), (3, (256, 128)), (2, (63, 40)), (2, (3, 64, 128))],
                         ids=["k2", "k3", "k2_unaligned", "k2_rank3"])
def test_speculative_cull_selection_on_device(engine, k, shape):
    emul_tier.test_speculative_cull_selection_never_changes_a_bit(engine, k, shape)


@pytest.mark.parametrize("k,shape", [(3, (64, 128)), (4, (128, 64)), (3, (66, 40)), (3, (14336, 1024))], ids=["k3", "k4", "k3_unaligned", "k3_folded"])
def test_row_passes_in_one_launch_on_device(engine, k, shape):
    emul_tier.test_row_passes_of_all_deltas_in_one_launch_change_nothing(engine, k, shape)


@pytest.mark.parametrize("k,shape", [(3, (64, 128)), (4, (128, 64)), (3, (66, 40)), (3, (4096, 1024))], ids=["k3", "k4", "k3_unaligned", "k3_4096x1024"])
def test_column_passes_of_a_raw_pair_in_one_launch_on_device(engine, k, shape):
    emul_tier.test_column_passes_of_a_raw_pair_in_one_launch_change_nothing(engine, k, shape)


def test_cull_speculation_on_alternating_layers_on_device(engine):
    emul_tier.test_cull_speculation_on_alternating_layers_misses_and_changes_nothing(engine)


@pytest.mark.parametrize("shape", [(256,), (1000,), (4096,), (1, 512), (8192,), (77,), (6144,)], ids=lambda s: "x".join(map(str, s)))
def test_one_launch_pair_merge_of_1d_tensors_on_device(engine, shape):
    emul_tier.test_one_launch_pair_merge_of_1d_tensors(engine, shape)


def test_1d_policies_through_the_one_launch_kernel_on_device(engine):
    emul_tier.test_1d_policies_through_the_one_launch_kernel(engine)


@pytest.mark.parametrize("case", gi.LEGACY_CASES, ids=lambda c: c["id"])
def test_legacy_fourier_operator_on_device(engine, golden, case):
    """the reference's older FourierMerge class (shard/merge/fourier.py:35-205) behind the same boundary"""
    print(pc.check_legacy(engine, golden, case))
