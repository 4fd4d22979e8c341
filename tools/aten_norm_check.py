#!/usr/bin/env python3
"""norm_mode = reference_cpu on the device: smhip_reference_cpu_norm against torch.norm on the host (bit for bit),
per-kernel times and the walker's statistics.  Usage: python tools/aten_norm_check.py [out.json]"""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from shardmerge_amd.engine import get_engine  # noqa: E402


def main():
    eng = get_engine("cuda:0")
    dev = eng.device
    torch.manual_seed(0)
    rows = []
    cases = [("bf16delta", 1000), ("bf16delta", 4099), ("gauss", 65536 * 3 + 7), ("gauss", (1 << 20) + 5),
             ("bf16delta", 1 << 22), ("bf16delta", 4096 * 4096), ("gauss", 4096 * 4096 + 13),
             ("bf16delta", 8192 * 8192), ("gauss", 8192 * 8192), ("bf16delta", 28672 * 8192)]
    for kind, n in cases:
        if kind == "gauss":
            x = torch.randn(n) * 0.003
            b = None
        else:
            b = (torch.randn(n) * 0.02).bfloat16()
            x = (b.float() + torch.randn(n) * 0.003).bfloat16()
        t0 = time.time()
        ref = torch.norm(x.float() - (b.float() if b is not None else 0)).item()
        t_cpu = time.time() - t0
        xd = x.to(dev)
        bd = b.to(dev) if b is not None else None
        eng.reference_cpu_norm(xd, bd)            # warm-up (workspace)
        eng.ctx.profile(True)
        eng.ctx.profile_reset()
        mine = eng.reference_cpu_norm(xd, bd)
        prof = eng.ctx.profile_table()
        eng.ctx.profile(False)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            eng.reference_cpu_norm(xd, bd)
        torch.cuda.synchronize()
        wall = (time.time() - t0) / 5
        row = {"kind": kind, "n": n, "device": mine, "torch_cpu": ref, "equal": mine == ref,
               "chunks_composed": eng.ctx.debug_query("aten_fast"), "chunks_by_groups": eng.ctx.debug_query("aten_group"), "chunks_walked": eng.ctx.debug_query("aten_slow"),
               "wall_ms": wall * 1e3, "torch_cpu_ms": t_cpu * 1e3,
               "kernels_us": {k: round(v[1] * 1e3, 1) for k, v in prof.items()}}
        rows.append(row)
        print(json.dumps(row), flush=True)
        del xd, bd
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(rows, indent=1))
    assert all(r["equal"] for r in rows), "mismatch against torch.norm"


if __name__ == "__main__":
    main()
