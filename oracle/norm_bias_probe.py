#!/usr/bin/env python3
"""torch.norm on CPU fp32 tensors is not an accurate L2 norm for large tensors.

ATen's contiguous-L2 CPU kernel accumulates x*x serially in 8 fp32 lanes (AVX2
build); this script (a) shows the relative error vs size, (b) reproduces the
value bit-for-bit with an 8-lane serial emulation, and (c) measures what the
artefact does to the oracle's (= the reference's) merged output by evaluating
the oracle with torch norms and with exact norms.  Numbers are quoted in
DESIGN.md."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import spectral_oracle as so  # noqa: E402


def main():
    torch.manual_seed(0)
    print("# (a) relative error of torch.norm(fp32, cpu) vs an exact sum, Gaussian data sigma=0.002")
    for n in [1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 26]:
        d = torch.randn(n) * 0.002
        t, ex = torch.norm(d).item(), d.double().norm().item()
        print(f"n={n:>9}  torch.norm={t:.9g}  exact={ex:.9g}  rel={(t - ex) / ex:+.3e}")
    print("# (b) 8-lane serial fp32 emulation at n = 2^20")
    d = torch.randn(1 << 20) * 0.002
    x = d.numpy()
    acc = np.zeros(8, dtype=np.float32)
    for row in (x * x).reshape(-1, 8):
        acc += row
    tot = np.float32(0)
    for v in acc:
        tot = np.float32(tot + v)
    print(f"emulated={np.sqrt(tot):.9g}  torch.norm={torch.norm(d).item():.9g}")
    print("# (c) oracle(torch norms) vs oracle(exact norms), K=2 synthetic layers")
    for rows in [256, 1024, 2048]:
        base, fts = so.synthetic_layer(rows, rows, 2, seed=4000 + rows)
        t0 = time.time()
        tr1, tr2 = so.LayerTrace(), so.LayerTrace()
        o1 = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=tr1)
        with so.exact_norms():
            o2 = so.merge_layer(fts, [base, base], so.ALPHAS[:2], base, trace=tr2)
        print(f"{rows}x{rows}: target_norm {tr1.target_norm:.7g} vs {tr2.target_norm:.7g}  "
              f"bf16 out diff {so.rel_err(o1.float(), o2.float()):.2e}  merged-delta diff "
              f"{so.rel_err(tr1.merged_delta, tr2.merged_delta):.2e}  ({time.time() - t0:.1f}s)")


if __name__ == "__main__":
    main()
