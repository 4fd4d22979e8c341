#!/usr/bin/env python3
"""How the sampling rate of the class statistics (k_class_emf: reference_cpu mode's model of torch.norm on the gathered
slerp-class vectors) moves a layer: the same K = 2 / K = 3 layer with one piece in 16 (the default cap), 32, 64, 128, 256 -
the cosine and the class norms of every pair merge, the bf16 output, the kernel's time.
    python tools/emf_sample_check.py [rows cols k]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from shardmerge_amd.engine import get_engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 28672
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eng = get_engine("cuda")
g = torch.Generator(device="cuda").manual_seed(5)
base = (torch.randn(rows, cols, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
fts = [(base.float() + torch.randn(rows, cols, device="cuda", generator=g) * (0.002 + 0.001 * i)).to(torch.bfloat16) for i in range(k)]
alphas = [0.3 + 0.1 * i for i in range(k)]
ref = None
for cap in (16, 32, 64, 128, 256, 1):
    eng.ctx.debug_option("emf_max_sample", cap)
    out, rep = eng.merge_layer(fts, [base] * k, alphas, base, norm_mode="reference_cpu")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.merge_layer(fts, [base] * k, alphas, base, norm_mode="reference_cpu")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    dots = [i.dot for i, b in zip(rep.infos, rep.branches) if b == "slerp"]
    if ref is None:
        ref = (out.clone(), dots)
    diff = (out.float() - ref[0].float()).norm().item() / ref[0].float().norm().item()
    print(f"[{rows}x{cols}] K={k} emf_max_sample={cap:4d}: {ms:7.3f} ms/layer  dots {['%.9f' % d for d in dots]}  "
          f"max |dot - dot(16)| {max(abs(a - b) for a, b in zip(dots, ref[1])):.2e}  bf16 output vs cap 16: {diff:.2e}")
