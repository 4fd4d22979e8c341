#!/usr/bin/env python3
"""Compact per-kernel table over the block-tensor shapes of Llama-3-8B (and 8192^2):
one line per shape, device microseconds per launch of the big kernels + wall per layer.
    python tools/shapes_bench.py [k] [reps]
Use SHARDMERGE_HIP_LIB to A/B two builds in one GPU session (devices differ by ~10%)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from shardmerge_amd.engine import get_engine

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
SHAPES = [(8192, 8192), (14336, 4096), (4096, 14336), (4096, 4096), (1024, 4096), (1, 4096)]
if len(sys.argv) > 3 and sys.argv[3] == "70b":
    SHAPES = [(28672, 8192), (8192, 28672), (8192, 8192), (1024, 8192), (1, 8192)]
COLS = ["f1_rows_fwd", "f2_cols_fwd", "select_lvl2", "blend", "i1_cols_inv", "i2_rows_inv", "combine"]
eng = get_engine("cuda")
print(f"{'shape':>12s} {'wall_us':>8s} {'sum_us':>8s} " + " ".join(f"{c[:11]:>11s}" for c in COLS) + "   other  GB/s  frac")
tot_wall = 0.0
for rows, cols in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    shape = (cols,) if rows == 1 else (rows, cols)
    base = (torch.randn(shape, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    fts = [(base.float() + torch.randn(shape, generator=g, device="cuda") * s).to(torch.bfloat16) for s in (0.002, 0.003, 0.0025, 0.004)[:k]]
    al = (0.3, 0.5, 0.2, 0.4)[:k]
    eng.merge_layer(fts, [base] * k, al, base)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        eng.merge_layer(fts, [base] * k, al, base)
    torch.cuda.synchronize()
    wall = (time.time() - t0) / reps * 1e6
    eng.ctx.profile(True); eng.ctx.profile_reset()
    for _ in range(reps):
        eng.merge_layer(fts, [base] * k, al, base)
    torch.cuda.synchronize()
    tab = eng.ctx.profile_table(); eng.ctx.profile(False)
    per = {n: ms / reps * 1e3 for n, (c, ms) in tab.items()}
    s = sum(per.values())
    other = s - sum(per.get(c, 0) for c in COLS)
    n = rows * cols
    print(f"{rows:>6d}x{cols:<5d} {wall:8.0f} {s:8.0f} " + " ".join(f"{per.get(c, 0):11.1f}" for c in COLS) +
          f" {other:7.1f} {2*n/wall/1e3:5.1f} {({2:60,3:122,4:182}[k])*n/(wall*1e-6)/8e12:5.3f}")
    tot_wall += wall * {(8192, 8192): 0, (14336, 4096): 2, (4096, 14336): 1, (4096, 4096): 2, (1024, 4096): 2, (1, 4096): 2}.get((rows, cols), 0)
print(f"Llama-3-8B block (9 tensors) = {tot_wall/1e3:.2f} ms -> {218112000*2/tot_wall/1e3:.1f} GB/s single-stream")
