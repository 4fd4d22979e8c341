#!/usr/bin/env python3
"""End-to-end `python -m shard merge` on an on-disk synthetic model:
writes base + K finetunes with Llama-3-8B block shapes under a RAM-backed directory, then
runs the CLI's merge with the prefetching loader on and off and prints one JSON line per run
(wall time includes reading the safetensors shards, H2D, the merge, D2H and writing the output).

    python tools/cli_bench.py [--blocks 4] [--k 2] [--root /dev/shm/smcli] [--device cuda]
"""
import argparse
import asyncio
import json
import os
import shutil
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import yaml
from safetensors.torch import save_file

BLOCK = [("self_attn.q_proj.weight", 4096, 4096), ("self_attn.k_proj.weight", 1024, 4096),
         ("self_attn.v_proj.weight", 1024, 4096), ("self_attn.o_proj.weight", 4096, 4096),
         ("mlp.gate_proj.weight", 14336, 4096), ("mlp.up_proj.weight", 14336, 4096),
         ("mlp.down_proj.weight", 4096, 14336), ("input_layernorm.weight", 1, 4096),
         ("post_attention_layernorm.weight", 1, 4096)]
BLOCK_70B = [("self_attn.q_proj.weight", 8192, 8192), ("self_attn.k_proj.weight", 1024, 8192),
             ("self_attn.v_proj.weight", 1024, 8192), ("self_attn.o_proj.weight", 8192, 8192),
             ("mlp.gate_proj.weight", 28672, 8192), ("mlp.up_proj.weight", 28672, 8192),
             ("mlp.down_proj.weight", 8192, 28672), ("input_layernorm.weight", 1, 8192),
             ("post_attention_layernorm.weight", 1, 8192)]
SIGMA = (0.002, 0.003, 0.0025, 0.004)


def write_models(root: Path, blocks: int, k: int, gen_device: str, model: str = "llama3-8b"):
    storage = root / "storage"
    uris = ["org/base"] + [f"org/ft{i}" for i in range(1, k + 1)]
    g = torch.Generator(device=gen_device).manual_seed(1000)
    total = 0
    block = BLOCK_70B if model == "llama3-70b" else BLOCK
    hidden = block[0][2]
    for b in range(-1, blocks + 1):               # -1: embed shard, blocks: norm + lm_head shard
        if b == -1:
            items = [("model.embed_tokens.weight", 2048, hidden)]
            shard = "model-embed.safetensors"
        elif b == blocks:
            items = [("model.norm.weight", 1, hidden), ("lm_head.weight", 2048, hidden)]
            shard = "model-head.safetensors"
        else:
            items = [(f"model.layers.{b}.{n}", r, c) for n, r, c in block]
            shard = f"model-{b:05d}.safetensors"
        base = {}
        for name, r, c in items:
            shape = (c,) if r == 1 else (r, c)
            base[name] = (torch.randn(shape, generator=g, device=gen_device) * 0.02).to(torch.bfloat16)
        for which, uri in enumerate(uris):
            d = storage / uri
            d.mkdir(parents=True, exist_ok=True)
            if which == 0:
                tens = {n: t.cpu() for n, t in base.items()}
            else:
                tens = {n: (t.float() + torch.randn(t.shape, generator=g, device=gen_device) * SIGMA[which - 1]).to(torch.bfloat16).cpu()
                        for n, t in base.items()}
            save_file(tens, str(d / shard), metadata={"format": "pt"})
            idx_path = d / "model.safetensors.index.json"
            doc = json.load(open(idx_path)) if idx_path.exists() else {"metadata": {"total_size": 0}, "weight_map": {}}
            doc["weight_map"].update({n: shard for n in tens})
            json.dump(doc, open(idx_path, "w"))
            if which == 0:
                total += sum(t.numel() for t in tens.values())
    cfg = {"output_base_model": "org/base",
           "finetune_merge": [{"model": f"org/ft{i}", "base": "org/base", "alpha": (0.3, 0.5, 0.2, 0.4)[i - 1], "is_input": i == 1}
                              for i in range(1, k + 1)],
           "output_dir": str(root / "merged"), "output_dtype": "bfloat16", "device": "cuda",
           "cache_dir": str(root / "cache"), "storage_dir": str(storage)}
    p = root / "merge.yaml"
    yaml.safe_dump(cfg, open(p, "w"))
    return p, total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--root", default="/dev/shm/smcli")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b"],
                    help="block shapes; llama3-70b with --k 3 is the metric's configuration (1.7 GB of bf16 per block and model)")
    ap.add_argument("--runs", default="0,1,0,1", help="SHARDMERGE_PREFETCH value of each run")
    args = ap.parse_args()
    from shardmerge_amd.constants import tune_hip_queues
    tune_hip_queues()
    root = Path(args.root)
    if root.exists():
        shutil.rmtree(root)
    root.mkdir(parents=True)
    t0 = time.time()
    cfg_path, n_params = write_models(root, args.blocks, args.k, args.device if torch.cuda.is_available() else "cpu", args.model)
    print(f"# wrote {args.k + 1} models, {n_params / 1e6:.0f} M params each, in {time.time() - t0:.1f} s under {root}", file=sys.stderr)

    from shardmerge_amd.__main__ import run_merge
    from shardmerge_amd.config import MergeConfig
    from shardmerge_amd import iostats
    for prefetch in args.runs.split(","):
        inplace = prefetch.endswith("i")              # "1i": prefetch on + in-place output shards (SHARDMERGE_INPLACE)
        prefetch = prefetch.rstrip("i")
        os.environ["SHARDMERGE_INPLACE"] = "1" if inplace else "0"
        os.environ["SHARDMERGE_PREFETCH"] = prefetch
        out_dir = root / "merged"
        if out_dir.exists():
            shutil.rmtree(out_dir)
        config = MergeConfig.from_yaml(cfg_path)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        iostats.snapshot(reset=True)
        t0 = time.time()
        asyncio.run(run_merge(config, args.device, clean_cache=False))
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        dt = time.time() - t0
        st = iostats.snapshot(reset=True)
        in_gb, out_gb = 2.0 * n_params * (args.k + 1) / 1e9, 2.0 * n_params / 1e9
        rec = {"cli_merge": "end to end", "model": args.model, "prefetch": prefetch == "1", "inplace_shards": inplace, "blocks": args.blocks, "k": args.k,
               "params": n_params, "seconds": round(dt, 3), "merged_GBps": round(out_gb / dt, 3), "input_GB": round(in_gb, 2),
               "stages": st}
        # what the PCIe link allows: every input byte crosses it once, every output byte once, at the H2D rate
        # measured on the copy stream in this run (the two directions are separate lanes: max, not sum)
        h2d = st.get("h2d", {}).get("GBps")
        if h2d:
            rec["pcie_bound_merged_GBps"] = round(out_gb / (max(in_gb, out_gb) / h2d), 3)
            rec["fraction_of_pcie_bound"] = round(rec["merged_GBps"] / rec["pcie_bound_merged_GBps"], 3)
        print(json.dumps(rec))
    if not args.keep:
        shutil.rmtree(root)


if __name__ == "__main__":
    main()
