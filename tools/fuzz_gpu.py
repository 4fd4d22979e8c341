#!/usr/bin/env python3
"""Randomised parity sweep on the device: K=2 layers of random supported shapes (planned, split column lengths,
rough row lengths, both lengths rough), noise levels and alphas against the oracle (CPU), through the C ABI.
norm_mode reference_cpu (default) is held against the oracle AS IT IS, exact against the exact-norm oracle.
Sizes stay small enough for the oracle to finish in a fraction of a second each.
    python tools/fuzz_gpu.py [cases] [seed] [norm_mode] [k]
K >= 3 is held to the reference's own rounding floor: branches equal, bf16 output within 5e-3."""
import contextlib
import math
import random
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from oracle import spectral_oracle as so
from shardmerge_amd.engine import get_engine
from tests import parity_checks as pc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mode = sys.argv[3] if len(sys.argv) > 3 else "reference_cpu"
K = int(sys.argv[4]) if len(sys.argv) > 4 else 2
eng = get_engine("cuda")


def supported(nmax):
    out = []
    for n in range(2, nmax + 1):
        m = n
        for p in (2, 3, 5, 7, 11, 13):
            while m % p == 0:
                m //= p
        if m == 1:
            out.append(n)
    return out


@contextlib.contextmanager
def oracle_fft_in_fp64():
    """the oracle with its transforms evaluated in fp64 and rounded back (what oracle/chaos_probe.py does to the real
    reference): the distance between this and the plain oracle is the floor of a K >= 3 case - how far the reference's
    result moves when only the rounding of its own FFT changes"""
    f0, i0 = so.fft_transform, so.ifft_transform

    def f(x):
        x = x.to(torch.float64)
        return (torch.fft.fft(x) if x.ndim == 1 else torch.fft.fftn(x, dim=(-2, -1))).to(torch.complex64)

    def i(sp):
        sp = sp.to(torch.complex128)
        return (torch.fft.ifft(sp) if sp.ndim == 1 else torch.fft.ifftn(sp, dim=(-2, -1))).real.to(torch.float32)
    so.fft_transform, so.ifft_transform = f, i
    try:
        yield
    finally:
        so.fft_transform, so.ifft_transform = f0, i0


LENS = supported(4096)
ROUGH = sorted({p * m for p in (17, 19, 23, 29, 37, 43, 71) for m in LENS if m % 2 == 0 and p * m <= 12000})
worst = (0, None)
bad = 0
for ci in range(cases):
    while True:
        rows = rng.choice(LENS + [1] * 40)
        cols = rng.choice(LENS)
        kind = rng.random()
        if kind < 0.15:
            rows = rng.choice(ROUGH)            # split column length (k_dftp)
        elif kind < 0.30:
            cols = rng.choice(ROUGH)            # rough row length: merged transposed (also 1-D)
        elif kind < 0.42:                       # no planned length at all: column split + chirp-z row passes
            rows = rng.choice(ROUGH)
            cols = rng.choice(ROUGH + [17, 19, 23, 51, 71, 213, 323])
        if 64 <= rows * cols <= 1 << 21 and eng.lib.shape_supported(rows, cols):
            break
    k = K
    g = torch.Generator().manual_seed(rng.randrange(1 << 30))
    shape = (cols,) if rows == 1 else (rows, cols)
    base = (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
    sig = [10 ** rng.uniform(-3.2, -2.0) for _ in range(k)]
    fts = [(base.float() + torch.randn(shape, generator=g) * s).to(torch.bfloat16) for s in sig]
    alphas = [rng.uniform(0.05, 1.0) for _ in range(k)]
    out, rep, delta = eng.merge_layer([t.cuda() for t in fts], [base.cuda()] * k, alphas, base.cuda(), want_delta=True, norm_mode=mode)
    # a rough ROW length is merged transposed: the reference's threshold-tie bins depend on the orientation
    # (DESIGN.md section 3), so the better of the two oracle orientations is the bar there
    flips = (False, True) if not eng.lib.length_supported(cols) and (eng.lib.length_supported(rows) or rows % 2) else (False,)
    best = None
    for flip in flips:
        tt = (lambda x: x.reshape(rows, cols).T.contiguous()) if flip else (lambda x: x)
        trx = so.LayerTrace()
        with (so.exact_norms() if mode == "exact" else contextlib.nullcontext()):
            refx = so.merge_layer([tt(f) for f in fts], [tt(base)] * k, alphas, tt(base), trace=trx)
        r_, c_ = (cols, rows) if flip else (rows, cols)
        res = pc.spectral_residual(tt(delta.cpu()).reshape(max(r_, 1), c_), trx.merged_delta.reshape(max(r_, 1), c_))
        cand = (res[1], res[0], so.rel_err(tt(out.cpu()).float(), refx.float()), rep.branches == trx.branches, trx)
        best = cand if best is None or cand[0] < best[0] else best
    d_resid, d_total, out_err, ok, trx = best
    n = rows * cols
    tol_total = 10.0 / math.sqrt(n) + 1e-5
    fine = ok and d_resid < 5e-5 and d_total < tol_total and out_err < max(2e-3, 0.05 * tol_total)   # tiny tensors: tie-bin floor ~ 1/sqrt(n)
    if k >= 3:
        # the reference's own floor (its result moves by this much when its FFT is evaluated in fp64, oracle/chaos_probe.py):
        # merged delta 3.1e-2 at K = 3, 2e-1 at K = 4; the bf16 output sees it scaled by |delta| / |output|
        floor = 4.5e-2 if k == 3 else 2.5e-1
        fine = ok and d_total < max(floor, 3 * tol_total)
        if ok and not fine:
            # this case's OWN floor: the oracle against itself with fp64 transforms (same orientation as the comparison)
            tr64 = so.LayerTrace()
            with (so.exact_norms() if mode == "exact" else contextlib.nullcontext()), oracle_fft_in_fp64():
                so.merge_layer(list(fts), [base] * k, alphas, base, trace=tr64)
            trp = so.LayerTrace()
            with (so.exact_norms() if mode == "exact" else contextlib.nullcontext()):
                so.merge_layer(list(fts), [base] * k, alphas, base, trace=trp)
            own = so.rel_err(tr64.merged_delta, trp.merged_delta)
            fine = tr64.branches == trp.branches and d_total < 1.5 * own
            print(f"    (own floor of this case: the oracle moves by {own:.2e} with fp64 transforms)")
    tag = "ok " if fine else "BAD"
    # the reference's imaginary detour (functions.py:152-158) divides 0 by 0 on a few odd row lengths ([64 x 17]):
    # its NaN -> 0 policy then wipes most of the merged delta - reported, not counted
    if not fine and ok and float(trx.merged_delta.norm()) < 0.3 * float(delta.norm()):
        tag = "REF"
    if tag == "BAD":
        bad += 1
    if d_resid > worst[0]:
        worst = (d_resid, (rows, cols))
    print(f"{tag} [{rows}x{cols}] K={k} sig={sig[0]:.1e},{sig[1]:.1e} branches={rep.branches} vs {trx.branches} "
          f"delta total {d_total:.2e} (tol {tol_total:.1e}) beyond-ties {d_resid:.2e} out {out_err:.2e}")
print(f"norm_mode {mode}: {cases} cases, {bad} bad; worst beyond-tie residual {worst[0]:.2e} at {worst[1]}")
sys.exit(1 if bad else 0)
