#!/usr/bin/env python3
"""Per-kernel device times of one merge_layer call (library HIP-event profiling).
usage: python tools/kprof.py [rows cols [k [reps]]]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from shardmerge_amd.engine import get_engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
eng = get_engine("cuda")
import os
for item in filter(None, os.environ.get("SMHIP_DEBUG", "").split(",")):      # e.g. SMHIP_DEBUG=fold_columns=0,spectral_intermediates=0
    key, val = item.split("=")
    eng.ctx.debug_option(key, int(val))
g = torch.Generator(device="cuda").manual_seed(1)
shape = (cols,) if rows == 1 else (rows, cols)
base = (torch.randn(shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
fts = [(base.float() + torch.randn(shape, generator=g, device="cuda") * s).to(torch.bfloat16) for s in (0.002, 0.003, 0.0025, 0.004)[:k]]
alphas = (0.3, 0.5, 0.2, 0.4)[:k]
mode = os.environ.get("KPROF_NORM_MODE", "reference_cpu")          # the mode bench.py measures; "exact" for the other
eng.merge_layer(fts, [base] * k, alphas, base, norm_mode=mode)
torch.cuda.synchronize()
import time
t0 = time.time()
for _ in range(reps):
    eng.merge_layer(fts, [base] * k, alphas, base, norm_mode=mode)
torch.cuda.synchronize()
wall = (time.time() - t0) / reps * 1e3
eng.ctx.profile(True); eng.ctx.profile_reset()
for _ in range(reps):
    eng.merge_layer(fts, [base] * k, alphas, base, norm_mode=mode)
torch.cuda.synchronize()
tab = eng.ctx.profile_table()
eng.ctx.profile(False)
n = rows * cols
tot = sum(ms for _, ms in tab.values()) / reps
print(f"[{rows}x{cols}] K={k} norm_mode={mode}: wall {wall:.3f} ms/layer (unprofiled), sum of kernels {tot:.3f} ms, "
      f"merged {2*n/wall/1e6:.1f} GB/s, pipeline frac {({2:60,3:122,4:182}[k])*n/(wall*1e-3)/8e12:.3f}")
for name, (cnt, ms) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:16s} launches/layer {cnt/reps:5.1f}  {ms/reps:8.3f} ms/layer  {ms/cnt*1e3:9.1f} us/launch")
