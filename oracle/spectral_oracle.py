"""CPU oracle for the shardmerge per-layer spectral-merge hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product path (``shardmerge_amd``) never does and fails loudly when the
HIP library is missing.

What it is: a from-scratch restatement, on torch CPU ops, of the arithmetic the
reference performs in ``shard/tensor/functions.py`` and
``shard/merge/fast_fourier.py`` (reference @ /root/reference, file:line cited on
every function).  It is floating point (fp32 / complex64) work, so it is stated
with the same ATen primitives the reference calls (``torch.fft.fftn``,
``torch.sort``, ``torch.norm``) rather than in C: those are the third-party
arithmetic of the path (PyTorch >= 2.9.1 per the reference's pyproject.toml:11;
2.10.0 installed in this image).

Pinning: the reference's own tests hold no numeric golden vectors for this
path (shape/dtype/no-NaN only), so the oracle is pinned against outputs of the
reference itself, generated in the build container by ``oracle/gen_golden.py``
(which imports the reference from /root/reference) and committed under
``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every function here
against those vectors.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import Iterator, List, Optional, Sequence, Tuple

import torch

CPU = "cpu"

# ---------------------------------------------------------------------------
# L2 norms.  The reference takes every norm with torch.norm / Tensor.norm /
# F.normalize on CPU fp32 tensors.  ATen's CPU kernel for that accumulates x*x
# serially in 8 fp32 vector lanes (AVX2), which loses low-order bits once the
# running sum is large: -1e-5 relative at 1M elements, -7e-4 at 16M, -5e-3 at
# 67M (oracle/norm_bias_probe.py).  NORM_MODE = "torch" restates the reference
# as it is (and is what the goldens pin); "exact" takes the same norms with an
# accurate sum - what the reference's own device="cuda" mode and the HIP path
# compute - so that parity tests at >= 1M elements can separate that artefact
# from everything else.
# ---------------------------------------------------------------------------
NORM_MODE = "torch"


class exact_norms:
    """Context manager: evaluate the oracle with accurate L2 norms."""

    def __enter__(self):
        global NORM_MODE
        self._old = NORM_MODE
        NORM_MODE = "exact"

    def __exit__(self, *exc):
        global NORM_MODE
        NORM_MODE = self._old


def l2norm(x: torch.Tensor) -> torch.Tensor:
    if NORM_MODE == "exact":
        return x.double().pow(2).sum().sqrt().to(torch.float32)
    return x.norm()


# --------------------------------------------------------------------------
# A6 - slerp on the gathered (1-D, masked) real parts
# --------------------------------------------------------------------------
def slerp(v0: torch.Tensor, v1: torch.Tensor, t: float, info: Optional[dict] = None) -> torch.Tensor:
    """reference shard/tensor/functions.py:24-43.

    Note the reference's quirk: the cosine ``c`` is taken between
    the *un-normalised* vectors, ``v1 - c*v0`` is normalised to unit length and
    the result is ``v0*cos(theta) + unit*sin(theta)``.
    """
    c = torch.sum(v0 * v1) / (l2norm(v0) * l2norm(v1))
    c = torch.clamp(c, -1.0, 1.0)
    if info is not None:
        info["dot"] = float(c)
    theta = torch.acos(c) * t
    rel = v1 - v0 * c
    if NORM_MODE == "exact":
        rel = rel / torch.clamp(l2norm(rel), min=1e-12)
    else:
        rel = torch.nn.functional.normalize(rel, dim=-1)
    return v0 * torch.cos(theta) + rel * torch.sin(theta)


# --------------------------------------------------------------------------
# A4 / A8 - transforms and normalisation
# --------------------------------------------------------------------------
def fft_transform(x: torch.Tensor) -> torch.Tensor:
    """reference shard/tensor/functions.py:45-58 (device fixed to cpu)."""
    x = x.to(torch.float32)
    if x.ndim == 1:
        return torch.fft.fft(x)
    return torch.fft.fftn(x, dim=(-2, -1))


def ifft_transform(spec: torch.Tensor) -> torch.Tensor:
    """reference shard/tensor/functions.py:60-73: real part of the inverse."""
    if spec.ndim == 1:
        return torch.fft.ifft(spec).real
    return torch.fft.ifftn(spec, dim=(-2, -1)).real


def normalize_tensor(x: torch.Tensor) -> Tuple[torch.Tensor, float]:
    """reference shard/tensor/functions.py:75-88: x/||x||, untouched if 0."""
    nrm = l2norm(x).item()
    if nrm == 0:
        return x, nrm
    return x / nrm, nrm


# --------------------------------------------------------------------------
# A5 / A7 - real-part class logic, thresholds, imaginary detour
# --------------------------------------------------------------------------
@dataclass
class BlendTrace:
    """Side information the GPU parity tests compare against."""
    cutoff_threshold: float = 0.0
    cull_threshold: float = 0.0
    n_slerp: int = 0
    n_sum: int = 0
    n_rest: int = 0
    n_culled: int = 0
    dot: float = 0.0                      # clamped cosine of the slerp class (functions.py:36-37)
    t: float = 0.0                        # slerp fraction the caller passed
    cull_pct: float = 0.0                 # cull fraction the caller passed
    culled_mask: Optional[torch.Tensor] = None   # bool, full spectrum: bins zeroed by the cull


# The reference takes its two order statistics with a full torch.sort (85 % of its run time).  SELECT_MODE =
# "kthvalue" takes the SAME element with torch.kthvalue (a selection, 7x faster at 134 M values): test-side
# shortcut for the full-size GPU parity tests, pinned equal to the sort on every golden case by
# tests/test_oracle_golden.py.  The default restates the reference as it is.
SELECT_MODE = "sort"


class fast_select:
    """Context manager: order statistics by selection instead of a full sort (same values)."""

    def __enter__(self):
        global SELECT_MODE
        self._old = SELECT_MODE
        SELECT_MODE = "kthvalue"

    def __exit__(self, *exc):
        global SELECT_MODE
        SELECT_MODE = self._old


def kth_smallest(values: torch.Tensor, fraction: float) -> float:
    """Order statistic the reference takes with a full sort
    (functions.py:114-119 and :139-141): element ``int(len*fraction)`` of the
    ascending sort, clamped to the last element."""
    flat = values.ravel()
    idx = int(len(flat) * fraction)
    if idx >= len(flat):
        idx = len(flat) - 1
    if SELECT_MODE == "kthvalue":
        return torch.kthvalue(flat, idx + 1).values.item()
    flat, _ = torch.sort(flat, descending=False)
    return flat[idx].item()


def interpolate_fft_components(
    f0: torch.Tensor,
    f1: torch.Tensor,
    t: float,
    t_sum: float = 1.0,
    cutoff_pct: float = 0.0,
    cull_pct: float = 0.0,
    interp_imag: bool = True,
    trace: Optional[BlendTrace] = None,
) -> torch.Tensor:
    """reference shard/tensor/functions.py:90-162.

    Quirks kept: both "small" masks test |Re f1| (Q2, :125-126); sign(0)=0 is
    its own sign class (:124); the imaginary part goes through a second FFT,
    an interpolation with interp_imag=False and an inverse (A7, :152-158).
    """
    out = torch.zeros_like(f0)
    r0, r1 = f0.real, f1.real
    m0, m1 = r0.abs(), r1.abs()

    thr = 0
    if cutoff_pct > 0:
        thr = kth_smallest(torch.cat([m0, m1]), cutoff_pct)

    agree = r0.sign() == r1.sign()
    small = m1 < thr
    sel_slerp = agree & ~small & ~small
    sel_sum = agree & ~sel_slerp
    sel_rest = ~sel_slerp & ~sel_sum
    v0_bigger = m0 > m1

    sl_info: dict = {}
    out.real[sel_slerp] = slerp(r0[sel_slerp], r1[sel_slerp], t, sl_info)
    out.real[sel_sum] = r0[sel_sum] + t_sum * r1[sel_sum]
    out.real[sel_rest] = torch.where(v0_bigger[sel_rest], r0[sel_rest], r1[sel_rest])

    cull_thr = 0.0
    n_culled = 0
    kill = None
    if cull_pct > 0:
        if SELECT_MODE == "kthvalue":
            mag = out.real.abs().ravel()
            cull_thr = torch.kthvalue(mag, int(len(mag) * cull_pct) + 1).values.item()
        else:
            mag, _ = torch.sort(out.real.abs().ravel(), descending=False)
            cull_thr = mag[int(len(mag) * cull_pct)].item()
        below = int((mag < cull_thr).sum())
        # functions.py:143 overflow guard (cannot fire on sorted data; kept)
        if not below > len(mag) * (cull_pct * 2):
            kill = torch.abs(out.real) < cull_thr
            n_culled = int(kill.sum())
            out.real[kill] = 0

    if trace is not None:
        trace.cutoff_threshold = float(thr)
        trace.cull_threshold = float(cull_thr)
        trace.n_slerp = int(sel_slerp.sum())
        trace.n_sum = int(sel_sum.sum())
        trace.n_rest = int(sel_rest.sum())
        trace.n_culled = n_culled
        trace.dot = sl_info.get("dot", 0.0)
        trace.t = float(t)
        trace.cull_pct = float(cull_pct)
        trace.culled_mask = kill.clone() if kill is not None else None

    if interp_imag:
        g0 = fft_transform(f0.imag)
        g1 = fft_transform(f1.imag)
        g = interpolate_fft_components(g0, g1, t=t, cutoff_pct=0, cull_pct=0, interp_imag=False)
        out.imag = ifft_transform(g)
    else:
        out.imag = f0.imag
    return out


# --------------------------------------------------------------------------
# A9 - the SLERP-FFT pair merge
# --------------------------------------------------------------------------
def merge_tensors_fft2_slerp(
    v0: torch.Tensor,
    v1: torch.Tensor,
    t: float,
    b: float = 0.1,
    t_sum: float = 1.0,
    cutoff_pct: float = 0.0,
    cull_pct: float = 0.0,
    trace: Optional[BlendTrace] = None,
):
    """reference shard/tensor/functions.py:164-221.

    Early-outs (Q7) return the *normalised* v0 alone, not a 3-tuple-of-merged:
    the reference returns ``(v0, n0, n1)`` there too, so the shape of the
    return value is the same in all branches.
    """
    v0, n0 = normalize_tensor(v0)
    v1, n1 = normalize_tensor(v1)
    if n1 < 1e-4:
        return v0, n0, n1
    if n0 < 1e-4:
        return v0, n0, n1

    f0 = fft_transform(v0)
    f1 = fft_transform(v1)
    if n1 / (n0 + 1e-10) < b:
        spec = f0 + f1 * t
    else:
        spec = interpolate_fft_components(
            f0, f1, t=t, t_sum=t_sum, cutoff_pct=cutoff_pct, cull_pct=cull_pct, trace=trace
        )
    merged = ifft_transform(spec)
    if torch.any(torch.isnan(merged)):
        merged = torch.where(torch.isnan(merged), torch.zeros_like(merged), merged)
    if torch.any(torch.isinf(merged)):
        raise ValueError("Inf in ifft output")
    return merged, n0, n1


# --------------------------------------------------------------------------
# A10 - the Arithmetic-FFT pair merge
# --------------------------------------------------------------------------
def arithmetic_fft_components(
    f0: torch.Tensor, f1: torch.Tensor, t: float, agreement: bool, do_imag: bool = True
) -> torch.Tensor:
    """reference shard/tensor/functions.py:256-302.

    Quirk Q3 (:282-284): ``larger = |Re f0| > |Re f0|`` is identically False,
    so sign-disagreeing bins always take Re f1.
    """
    out = torch.zeros_like(f0)
    if agreement:
        agree = f0.real.sign() == f1.real.sign()
    else:
        agree = torch.ones_like(f0.real, dtype=torch.bool)
    out.real[agree] = f0.real[agree] + t * f1.real[agree]
    never = f0.real.abs() > f0.real.abs()
    out.real[~agree] = torch.where(never[~agree], f0.real[~agree], f1.real[~agree])
    if do_imag:
        g0 = fft_transform(f0.imag)
        g1 = fft_transform(f1.imag)
        g = arithmetic_fft_components(g0, g1, t=t, agreement=agreement, do_imag=False)
        out.imag = ifft_transform(g)
    else:
        out.imag = f0.imag
    return out


def task_arithmetic_fft2(v0: torch.Tensor, v1: torch.Tensor, t: float, agreement: bool = True) -> torch.Tensor:
    """reference shard/tensor/functions.py:224-254."""
    spec = arithmetic_fft_components(fft_transform(v0), fft_transform(v1), t=t, agreement=agreement)
    return ifft_transform(spec)


# --------------------------------------------------------------------------
# A3 - greedy pairing
# --------------------------------------------------------------------------
def correlated_pairs(corr: torch.Tensor, way: str = "least") -> Iterator[Tuple[int, int, float]]:
    """reference shard/tensor/functions.py:316-365.

    Greedy: among unused (i<j) cells take the first (row-major) one whose
    |corr| is extremal; unpaired indices come out last as (i, -1, corr[i,i]).
    Stated with plain Python loops over the K x K matrix (K <= ~8).
    """
    if way not in ("least", "most"):
        raise ValueError("Invalid way. Choose 'least' or 'most'.")
    k = corr.size(0)
    free = list(range(k))
    open_cell = [[j > i for j in range(k)] for i in range(k)]
    while any(any(row) for row in open_cell):
        cells = [(i, j) for i in range(k) for j in range(k) if open_cell[i][j]]
        mags = [abs(corr[i, j].item()) for (i, j) in cells]
        target = min(mags) if way == "least" else max(mags)
        # torch.nonzero(|valid| == target) in the reference scans the whole
        # matrix row-major with closed cells set to +inf: only open cells match
        # (unless target itself is inf, which cannot happen for finite norms).
        hit = next(((i, j) for (i, j), m in zip(cells, mags) if m == target), None)
        if hit is None:
            break
        x, y = hit
        yield (x, y, corr[x, y].item())
        for q in range(k):
            open_cell[x][q] = open_cell[q][x] = False
            open_cell[y][q] = open_cell[q][y] = False
        free.remove(x)
        free.remove(y)
    for i in free:
        yield (i, -1, corr[i, i].item())


def correlate_pairs(tensors: torch.Tensor) -> torch.Tensor:
    """reference shard/tensor/functions.py:304-314 (devices fixed to cpu): mean over the trailing
    positions of the cosine similarity along dim 0, NaN -> 0; symmetric, zero diagonal."""
    k = tensors.shape[0]
    out = torch.zeros(k, k)
    for i in range(k):
        for j in range(i + 1, k):
            c = torch.nn.functional.cosine_similarity(tensors[i], tensors[j], dim=0)
            out[i, j] = out[j, i] = c.nan_to_num(0).mean().item()
    return out


def name_hash(name: str) -> str:
    """reference shard/merge/fast_fourier.py:36-41."""
    short = "_".join(part[:4] for part in name.split("_"))
    return short + "::" + hashlib.sha256(name.encode()).hexdigest()[:8]


# --------------------------------------------------------------------------
# A1/A2/A11/A13 - the per-layer tournament of FourierMerge._merge_layer
# --------------------------------------------------------------------------
@dataclass
class LayerTrace:
    branches: List[str] = field(default_factory=list)   # "add" | "arith" | "slerp" | "carry"
    pairs: List[Tuple[int, int]] = field(default_factory=list)
    target_norm: float = 0.0
    merged_delta: Optional[torch.Tensor] = None          # fp32, before add-back
    steps: List[Optional[BlendTrace]] = field(default_factory=list)   # per pairing step; None: carry / add / arith
    step_norms: List[Tuple[float, float]] = field(default_factory=list)   # (||a||, ||b||) after the swap


def merge_layer(
    finetunes: Sequence[torch.Tensor],
    bases: Sequence[torch.Tensor],
    alphas: Sequence[float],
    base_out: torch.Tensor,
    names: Optional[Sequence[str]] = None,
    target_norm_offset: float = 1e-10,
    cull_start_pct: float = 0.20,
    cutoff_pct: float = 0.08,
    trace: Optional[LayerTrace] = None,
    layer_name: str = "layer",
    _mutation: Optional[str] = None,
    ratio_b: float = 0.1,
) -> torch.Tensor:
    """The block-tensor branch of FourierMerge._merge_layer,
    reference shard/merge/fast_fourier.py:132-276 (+ base.py:117-137 for the
    deltas), with the disk cache replaced by a dict.  Inputs are the tensors the
    index would hand out for the models that pass ``use_layer_index``.
    ``ratio_b`` is merge_tensors_fft2_slerp's ``b`` (the reference never overrides its 0.1).

    ``_mutation`` selects a DELIBERATELY WRONG variant ("keep_cull_pct": the cull
    fraction is not halved per round, :254; "swap_weights": weights follow the a/b
    swap, against quirk Q4 :212-215; "sum_weights": the merged weight is a_w + b_w
    instead of their mean, :247).  Only tests/ use it, to show that the K >= 3
    parity checks can tell such a slip from the reference's own rounding chaos.
    """
    k = len(finetunes)
    names = list(names) if names is not None else [f"model{i}" for i in range(k)]
    store = {}
    layer_norms: List[torch.Tensor] = []
    stack: List[str] = []
    weights: List[float] = []
    for i in range(k):
        delta = (finetunes[i].to(torch.float32) - bases[i].to(torch.float32)).detach() * 1
        layer_norms.append(l2norm(delta))
        store[names[i]] = delta
        stack.append(names[i])
        weights.append(alphas[i])

    target_norm = torch.tensor(layer_norms).mean().item() + target_norm_offset
    cull_pct = cull_start_pct
    if trace is not None:
        trace.target_norm = target_norm

    while len(stack) > 1:
        m = len(stack)
        vec = torch.stack(layer_norms)
        corr = torch.zeros((m, m), dtype=torch.float32)
        for i in range(m):
            for j in range(i + 1, m):
                corr[i, j] = vec[i] * vec[j]          # Q1: original norm list
        nxt_stack: List[str] = []
        nxt_weights: List[float] = []
        for x, y, _ in correlated_pairs(corr, way="least"):
            if y < 0:
                nxt_stack.append(stack[x])
                nxt_weights.append(weights[x])
                if trace is not None:
                    trace.branches.append("carry")
                    trace.pairs.append((x, -1))
                    trace.steps.append(None)
                    trace.step_norms.append((0.0, 0.0))
                continue
            a_name, b_name = stack[x], stack[y]
            a_w, b_w = weights[x], weights[y]
            a, b = store[a_name], store[b_name]
            na, nb = l2norm(a).item(), l2norm(b).item()
            if abs(na) < abs(nb):
                a, b = b, a
                a_name, b_name = b_name, a_name
                na, nb = nb, na                        # Q4: weights not swapped
                if _mutation == "swap_weights":
                    a_w, b_w = b_w, a_w
            ca = abs(na / target_norm)
            cb = abs(nb / target_norm)
            ratio = cb / (ca + 1e-10)
            btrace = None
            if ca < 1e-6:
                merged = a + b
                kind = "add"
            elif cb < 1e-6 or ratio < 0.1:
                s = target_norm / na
                w = b_w / (a_w + 1e-10)
                merged = task_arithmetic_fft2(a * s, b * w * s, t=1.0, agreement=True)
                kind = "arith"
            else:
                prop = a_w / (a_w + b_w)
                btrace = BlendTrace() if trace is not None else None
                merged, n0_, n1_ = merge_tensors_fft2_slerp(
                    a, b, t=prop, b=ratio_b, t_sum=1.0, cutoff_pct=cutoff_pct, cull_pct=cull_pct, trace=btrace
                )
                merged = merged * target_norm
                kind = "slerp"
                if n1_ < 1e-4 or n0_ < 1e-4:
                    kind = "early_v0"              # functions.py:184-190 (the HIP report names these)
                elif n1_ / (n0_ + 1e-10) < ratio_b:
                    kind = "linear"                # functions.py:196-202; only with ratio_b > 0.1 (N4)
            if trace is not None:
                trace.branches.append(kind)
                trace.pairs.append((x, y))
                trace.steps.append(btrace)
                trace.step_norms.append((na, nb))
            new_name = name_hash(f"{a_name}_{b_name}")
            nxt_stack.append(new_name)
            nxt_weights.append((a_w + b_w) if _mutation == "sum_weights" else (a_w + b_w) / 2.0)
            store[new_name] = merged
        stack, weights = nxt_stack, nxt_weights
        if _mutation != "keep_cull_pct":
            cull_pct = cull_pct / 2.0

    result = store[stack[0]]
    if trace is not None:
        trace.merged_delta = result.clone()
    result = base_out.to(torch.float32) + result
    if torch.any(torch.isnan(result)):
        result[torch.isnan(result)] = 0.0
    if torch.any(torch.isinf(result)):
        raise ValueError(f"Inf in merged tensor for {layer_name}")
    return result.to(torch.bfloat16)


# --------------------------------------------------------------------------
# N3 - AdditionMerge / TaskAdditionMerge
# --------------------------------------------------------------------------
def addition_merge(finetunes: Sequence[torch.Tensor], base: torch.Tensor) -> torch.Tensor:
    """reference shard/merge/addition.py:70-76: deltas relative to output_base_model's tensor,
    accumulated one by one in the tensors' dtype; the base is not added back."""
    out = torch.zeros_like(base)
    for ft in finetunes:
        out += ft - base
    return out


def task_addition_merge(finetunes: Sequence[torch.Tensor], base: torch.Tensor) -> torch.Tensor:
    """reference shard/merge/taskaddition.py:69-79: keep, per element, the deltas whose sign is
    the sign of the sum of signs; sum them."""
    stack = torch.stack([ft - base for ft in finetunes], dim=0)
    signs = torch.sign(stack)
    majority = torch.sum(signs, dim=0).sign()
    keep = signs == majority.unsqueeze(0)
    return torch.sum(stack * keep, dim=0)


# --------------------------------------------------------------------------
# synthetic inputs shared by tests / bench
# --------------------------------------------------------------------------
SIGMAS = (0.002, 0.003, 0.0025, 0.004)
ALPHAS = (0.3, 0.5, 0.2, 0.4)


def synthetic_layer(rows: int, cols: int, k: int, seed: int = 1000, sigmas: Sequence[float] = SIGMAS):
    """base = randn*0.02 (bf16); ft_i = base + randn*sigma_i (bf16)."""
    g = torch.Generator().manual_seed(seed)
    shape = (rows, cols) if rows > 0 else (cols,)
    base = (torch.randn(*shape, generator=g) * 0.02).to(torch.bfloat16)
    fts = []
    for i in range(k):
        gi = torch.Generator().manual_seed(seed + 1 + i)
        fts.append((base.float() + torch.randn(*shape, generator=gi) * sigmas[i % len(sigmas)]).to(torch.bfloat16))
    return base, fts


def rel_err(x: torch.Tensor, ref: torch.Tensor) -> float:
    x, ref = x.detach().cpu(), ref.detach().cpu()
    num = (x.double() - ref.double()).norm().item()
    den = ref.double().norm().item()
    return num / den if den > 0 else num
