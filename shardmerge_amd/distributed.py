"""Multi-GPU merge: one process per GPU (torchrun), tensors partitioned over ranks.

The path shards by independent units: every output tensor depends
only on the same-named tensors of the base and the finetunes, so there is no
data-path collective.  What IS shared is the base model: per base shard file,
one rank reads it from disk and ONE RCCL broadcast hands its payload to every
rank over xGMI (`torch.distributed` backend "nccl" is RCCL on ROCm); each rank
then merges the tensors it owns, reading only its own finetune tensors.

Output: every rank writes the tensors it merged of a shard into a part file;
after a barrier, shard `i` is assembled (CPU file I/O) by rank `i % world`, in
layer order, exactly as the single-process writer would have written it.
"""
from __future__ import annotations

import asyncio
import json
import logging
import os
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple

import torch

from .config import MergeConfig
from .constants import INPUT_LAYER, OUTPUT_LAYER
from .index import LocalModelIndex
from .merge.fast_fourier import FourierMerge
from .writer import ModelWriter, ShardLayer

logger = logging.getLogger(__name__)

# tests install a factory that returns the CPU-emulator engine; the product leaves it None
ENGINE_FACTORY: Optional[Callable] = None


def world_size() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def rank() -> int:
    return int(os.environ.get("RANK", "0"))


def local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def alg_bytes(numel: int, k: int) -> int:
    """Algorithmic HBM bytes of merging one tensor with k finetunes: k - 1 pair
    merges, 60n for the floor(k/2) pairs of raw deltas of the first round, 62n for every pair
    with an fp32 intermediate; 8n for k = 1."""
    if k <= 1:
        return 8 * numel
    raw_pairs = k // 2
    return numel * (60 * raw_pairs + 62 * (k - 1 - raw_pairs))


def partition_lpt(costs: List[int], world: int) -> List[int]:
    """Longest-processing-time-first assignment; deterministic (ties by index)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += costs[i]
    return owner


def _tensor_meta(index: LocalModelIndex, uri: str, shard_file: str) -> Dict[str, Tuple[List[int], str, int]]:
    """name -> (shape, dtype string, payload bytes) from a safetensors header (no payload read)."""
    path = index.storage_path / uri / shard_file
    with open(path, "rb") as fh:
        n = int.from_bytes(fh.read(8), "little")
        header = json.loads(fh.read(n))
    out = {}
    for name, rec in header.items():
        if name == "__metadata__":
            continue
        out[name] = (rec["shape"], rec["dtype"], rec["data_offsets"][1] - rec["data_offsets"][0])
    return out


_ST_DTYPES = {"BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32}


def init_process_group(device: torch.device):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if device.type == "cuda":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group("gloo")
    return dist


async def run_partitioned_merge(config: MergeConfig, index: LocalModelIndex, device: str):
    world, me = world_size(), rank()
    if ENGINE_FACTORY is not None:
        engine = ENGINE_FACTORY()
    else:
        from .engine import get_engine
        torch.cuda.set_device(local_rank())
        engine = get_engine(f"cuda:{local_rank()}")
    dev = engine.device
    dist = init_process_group(dev)

    merger = FourierMerge(config=config, index_manager=index, engine=engine)
    await merger.initialize()
    base_uri = config.output_base_model
    layer_order = index.get_layer_order(base_uri)
    weight_map = index.model_indexes[base_uri]["weight_map"]
    shards = sorted(set(weight_map.values()))
    out_dir = config.output_path
    out_dir.mkdir(parents=True, exist_ok=True)
    # part files of an earlier (crashed) run - another partition, world size or configuration -
    # must never be assembled into this run's shards
    if me == 0:
        for stale in list(out_dir.glob(".part-*")) + list(out_dir.glob(".tmp-*")):
            stale.unlink()
    if world > 1:
        dist.barrier()

    # ---- plan: the same on every rank --------------------------------------------------
    metas = {s: _tensor_meta(index, base_uri, s) for s in shards}
    names, costs = [], []
    for s in shards:
        for name in sorted((n for n in weight_map if weight_map[n] == s), key=layer_order.index):
            sl = ShardLayer(layer_order.index(name), s, name, False)
            number = sl.layer_number
            numel = 1
            for d in metas[s][name][0]:
                numel *= d
            if number in (INPUT_LAYER, OUTPUT_LAYER):
                cost = 2 * numel                       # passthrough: a copy
            else:
                k = sum(1 for m in config.finetune_merge if m.use_layer_index(number))
                cost = alg_bytes(numel, k)
            names.append((s, name))
            costs.append(cost)
    owner = partition_lpt(costs, world)
    mine = {names[i] for i in range(len(names)) if owner[i] == me}
    logger.info(f"rank {me}/{world}: {len(mine)} of {len(names)} tensors, "
                f"{sum(c for c, o in zip(costs, owner) if o == me) / 1e9:.2f} GB algorithmic traffic")

    # ---- this rank's finetune tensors are prefetched in processing order (loader.py); the base
    # comes from the broadcast, passthrough tensors from whichever model provides them
    my_order = [(s, name) for s in shards
                for name in sorted((n for (ss, n) in mine if ss == s), key=layer_order.index)]
    schedule = []
    for (s, name) in my_order:
        sl = ShardLayer(layer_order.index(name), s, name, False)
        if sl.layer_number >= 0:
            models = [m for m in config.finetune_merge if m.use_layer_index(sl.layer_number)]
            uris = [m.model for m in models] + [m.base for m in models if m.base != base_uri]
            schedule.append([(u, name) for u in dict.fromkeys(uris)])
        else:
            schedule.append(merger._layer_requests(sl))
    loader = None
    if os.environ.get("SHARDMERGE_PREFETCH", "1") != "0" and any(schedule):
        from .loader import PrefetchLoader
        loader = PrefetchLoader(index, str(dev))
        loader.start(schedule)
        merger._loader = loader
    pos = {key: i for i, key in enumerate(my_order)}
    import concurrent.futures
    part_writer = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="shardmerge-part")
    part_jobs = []

    # ---- per base shard: one broadcast, then every rank merges its own tensors -----------
    for si, s in enumerate(shards):
        root = si % world
        block = [n for n in metas[s] if ShardLayer(0, s, n, False).layer_number >= 0]
        offs, total = {}, 0
        for n in block:                                   # 256-byte aligned slots
            offs[n] = total
            total += (metas[s][n][2] + 255) // 256 * 256
        flat = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
        if me == root and total:
            for n in block:
                t = index.load_tensor(base_uri, n).contiguous()
                flat[offs[n]:offs[n] + metas[s][n][2]].copy_(t.view(torch.uint8).reshape(-1), non_blocking=False)
        if world > 1 and total:
            dist.broadcast(flat, src=root)              # the one collective: RCCL over xGMI
        views = {}
        for n in block:
            shape, dt, nbytes = metas[s][n]
            views[n] = flat[offs[n]:offs[n] + nbytes].view(_ST_DTYPES[dt]).reshape(shape)

        merged: Dict[str, torch.Tensor] = {}
        for (shard_name, name) in sorted(mine, key=lambda sn: layer_order.index(sn[1])):
            if shard_name != s:
                continue
            sl = ShardLayer(layer_order.index(name), s, name, False)
            if loader is not None:
                loader.begin_layer(pos[(s, name)])
            if sl.layer_number >= 0:
                out = await _merge_block_tensor(merger, engine, sl, views[name])
            else:
                out = await merger._merge_layer(sl, str(dev))
            merged[name] = out.detach().to("cpu").to(config.output_astype).contiguous()
        if merged:
            from safetensors.torch import save_file
            # serialised in the background while the next shard is broadcast and merged
            part_jobs.append(part_writer.submit(save_file, merged, str(out_dir / f".part-{me}-{s}"), {"format": "pt"}))
        del flat, views

    for job in part_jobs:
        job.result()
    part_writer.shutdown()
    if loader is not None:
        loader.close()
        merger._loader = None
    if world > 1:
        dist.barrier()

    # ---- assemble shards (CPU file I/O), index and README ----------------------------------
    from safetensors import safe_open
    from safetensors.torch import save_file
    for si, s in enumerate(shards):
        if si % world != me:
            continue
        tensors = {}
        planned = {names[i][1]: owner[i] for i in range(len(names)) if names[i][0] == s}
        for r in range(world):
            part = out_dir / f".part-{r}-{s}"
            if part.exists():
                with safe_open(str(part), framework="pt") as fh:
                    for k in fh.keys():
                        if planned.get(k) != r:
                            raise RuntimeError(f"{part.name} holds {k}, which the plan gave to rank {planned.get(k)}")
                        tensors[k] = fh.get_tensor(k)
        expected = {n for n in weight_map if weight_map[n] == s}
        if set(tensors) != expected:
            raise RuntimeError(f"Incomplete model output: shard {s} is missing {sorted(expected - set(tensors))}")
        ordered = {k: tensors[k] for k in sorted(tensors, key=layer_order.index)}
        tmp = out_dir / f".tmp-{me}-{s}"
        save_file(ordered, str(tmp), metadata={"format": "pt"})
        os.replace(tmp, out_dir / s)                     # a shard file is either absent or complete
    if world > 1:
        dist.barrier()
    if me == 0:
        for p in out_dir.glob(".part-*"):
            p.unlink()
        with open(out_dir / "model.safetensors.index.json", "w") as fh:
            json.dump(index.model_indexes[base_uri], fh, indent=2)
        with open(out_dir / "README.md", "w") as fh:
            fh.write(merger.get_readme())
    if world > 1:
        dist.barrier()


async def _merge_block_tensor(merger: FourierMerge, engine, sl: ShardLayer, base_view: torch.Tensor) -> torch.Tensor:
    """merge one block tensor whose output_base_model tensor arrived by broadcast"""
    cfg = merger.config
    number = sl.layer_number
    models = [m for m in cfg.finetune_merge if m.use_layer_index(number)]
    if not models:
        raise ValueError(f"No finetune covers layer {number} ({sl.layer_name})")
    dev = str(engine.device)
    loaded = {cfg.output_base_model: base_view}

    async def fetch(uri):
        if uri not in loaded:
            loaded[uri] = await merger._fetch(uri, sl.layer_name, dev)
        return loaded[uri]

    fts = [await fetch(m.model) for m in models]
    bases = [await fetch(m.base) for m in models]
    out, report = engine.merge_layer(fts, bases, [m.alpha for m in models], base_view,
                                     target_norm_offset=merger.target_norm_offset, cull_start_pct=merger.cull_start_pct,
                                     cutoff_pct=merger.cutoff_pct, t_sum=merger.t_sum, b=merger.b, layer_name=sl.layer_name)
    merger.last_report = report
    return out
