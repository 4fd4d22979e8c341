"""Full-size parity probe on the device: one [28672 x 8192] merge (the Llama-3-70B MLP shape), K = 2 or 3,
against the exact-norm oracle: per-step thresholds / class counts / cosines, the merged delta beyond the N
largest bins of the difference's spectrum (K = 2) or outside the bins earlier rounds culled (K = 3).
    python tools/full70b_check.py [k] [out.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import spectral_oracle as so  # noqa: E402
from shardmerge_amd.engine import get_engine  # noqa: E402
from tests import parity_checks as pc  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.set_num_threads(16)
eng = get_engine("cuda")
rows, cols = 28672, 8192
base, fts = so.synthetic_layer(rows, cols, k, seed=4242)
t0 = time.time()
trx = so.LayerTrace()
with so.exact_norms():
    refx = so.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, trace=trx)
t_oracle = time.time() - t0
print(f"oracle K={k}: {t_oracle:.1f} s", flush=True)
out, rep, delta = eng.merge_layer(fts, [base] * k, so.ALPHAS[:k], base, want_delta=True)
out, delta = out.cpu(), delta.cpu()
rec = {"shape": [rows, cols], "k": k, "seed": 4242, "oracle_seconds": round(t_oracle, 1),
       "branches_hip": rep.branches, "branches_ref": trx.branches, "pairs_hip": [list(s[:2]) for s in rep.steps], "pairs_ref": [list(p) for p in trx.pairs],
       "out_rel_err": so.rel_err(out.float(), refx.float()), "delta_rel_err": so.rel_err(delta, trx.merged_delta),
       "bf16_outputs_that_differ": (out.view(torch.int16) != refx.view(torch.int16)).float().mean().item(),
       "steps_hip": [vars(i) if i is not None else None for i in rep.infos],
       "steps_exact": [{kk: v for kk, v in vars(b).items() if kk != "culled_mask"} if b is not None else None for b in trx.steps]}
try:
    pc.check_layer_steps(rep, trx, out.numel())
    rec["check_layer_steps"] = "passed"
except AssertionError as e:
    rec["check_layer_steps"] = "FAILED: " + str(e)
if k == 2:
    for drop in (64, 256, 1024):
        rec[f"delta_beyond_{drop}_tie_bins"] = pc.spectral_residual(delta, trx.merged_delta, drop=drop)[1]
else:
    # as tests/parity_checks.masked_spectral_check, without its asserts: the spectrum of (delta - oracle's),
    # outside the bins that earlier rounds culled (there: the statistical floor only) and outside the bins
    # whose final cull decision differs; those are counted, and how many of them do NOT sit on the threshold
    import math
    ref = trx.merged_delta.double()
    D = torch.fft.fftn(delta.double() - ref)
    Rf, Hf = torch.fft.fftn(ref), torch.fft.fftn(delta.double())
    slerp_steps = [i for i, b in enumerate(trx.steps) if b is not None and b.culled_mask is not None]
    union = torch.zeros(ref.shape, dtype=torch.bool)
    for i in slerp_steps[:-1]:
        union |= trx.steps[i].culled_mask.reshape(ref.shape)
    mirror = lambda m: torch.roll(torch.flip(m, dims=(0, 1)), shifts=(1, 1), dims=(0, 1))
    union |= mirror(union)
    last = trx.steps[slerp_steps[-1]]
    thr = last.cull_threshold * trx.target_norm
    zero_ref, zero_hip = Rf.real.abs() < 1e-3 * thr, Hf.real.abs() < 1e-3 * thr
    flips = (zero_ref ^ zero_hip) & ~union
    mag = torch.where(zero_ref, Hf.real.abs(), Rf.real.abs())
    off_thr = int(((mag[flips] - thr).abs() > 1.5 * pc.LATER_ROUND_CULL_TOL * thr).sum())
    keep = ~union & ~flips & ~mirror(flips)
    e2, r2 = D.real ** 2 + D.imag ** 2, Rf.real ** 2 + Rf.imag ** 2
    rec.update({"delta_outside_culled_bins": math.sqrt(float(e2[keep].sum()) / float(r2[keep].sum())),
                "delta_inside_culled_bins": math.sqrt(float(e2[union].sum()) / max(float(r2[union].sum()), 1e-300)),
                "final_cull_flips": int(flips.sum()), "final_cull_flips_off_the_threshold": off_thr,
                "bins": int(ref.numel()), "bins_culled_by_earlier_rounds": int(union.sum())})
print(json.dumps({kk: v for kk, v in rec.items() if not kk.startswith("steps")}, indent=1))
for a, b in zip(rec["steps_hip"], rec["steps_exact"]):
    if a and b:
        print("  cutoff %.6g / %.6g   cull %.6g / %.6g   n_slerp %d / %d   dot %.6f / %.6f" %
              (a["cutoff_threshold"], b["cutoff_threshold"], a["cull_threshold"], b["cull_threshold"], a["n_slerp"], b["n_slerp"], a["dot"], b["dot"]))
if len(sys.argv) > 2:
    json.dump(rec, open(sys.argv[2], "w"), indent=1)
