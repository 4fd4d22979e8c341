#!/usr/bin/env python3
"""K = 2 merge of one shape vs the exact-norm oracle, folded and plain column pass:
residual after dropping the N largest bins of the difference's spectrum (tie bins)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from oracle import spectral_oracle as so
from tests import parity_checks as pc
from shardmerge_amd.engine import get_engine
rows, cols = int(sys.argv[1]), int(sys.argv[2])
torch.set_num_threads(16)
eng = get_engine("cuda")
base, fts = so.synthetic_layer(rows, cols, 2, seed=900 + rows + cols)
trx = so.LayerTrace()
with so.exact_norms():
    so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=trx)
for fold in (1, 0):
    eng.ctx.debug_option("fold_columns", fold)
    out, rep, delta = eng.merge_layer(fts, [base, base], so.ALPHAS[:2], base, want_delta=True)
    d = delta.cpu()
    res = [pc.spectral_residual(d, trx.merged_delta, drop=n)[1] for n in (0, 8, 32, 128, 512)]
    i, b = rep.infos[0], trx.steps[0]
    print(f"fold={fold}: residual after dropping 0/8/32/128/512 bins: " + " ".join(f"{r:.2e}" for r in res),
          f"| cut {i.cutoff_threshold:.9g}/{b.cutoff_threshold:.9g} cull {i.cull_threshold:.9g}/{b.cull_threshold:.9g} nsl {i.n_slerp}/{b.n_slerp}")
