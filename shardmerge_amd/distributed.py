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


def _shard_complete(path: Path, expected) -> bool:
    """An output shard that exists with exactly the expected tensors (written atomically)."""
    if not path.exists():
        return False
    try:
        with open(path, "rb") as fh:
            n = int.from_bytes(fh.read(8), "little")
            header = json.loads(fh.read(n))
    except Exception:
        return False
    return {k for k in header if k != "__metadata__"} == set(expected)


class _BaseShardReader:
    """The root rank's side of the base-shard broadcast: the shard's block tensors are read with
    positional reads (4 threads) into ONE pinned buffer and go to the device with ONE copy; the
    next shard this rank is root of is read in the background while the current one is merged."""

    def __init__(self, index: LocalModelIndex, uri: str, pinned: bool):
        from concurrent.futures import ThreadPoolExecutor
        self.index, self.uri, self.pinned = index, uri, pinned
        self.pool = ThreadPoolExecutor(max_workers=4, thread_name_prefix="shardmerge-base")
        self.pending = {}

    def _read(self, shard: str, layout, total: int) -> torch.Tensor:
        from .loader import ShardFile
        host = torch.empty(max(total, 1), dtype=torch.uint8, pin_memory=self.pinned)
        f = ShardFile(self.index.storage_path / self.uri / shard)
        try:
            jobs = [self.pool.submit(f.read_into, name, host[off:off + nbytes]) for name, (off, nbytes) in layout.items() if nbytes]
            for j in jobs:
                j.result()
        finally:
            f.close()
        return host

    def start(self, shard: str, layout, total: int):
        import threading
        box = {}

        def run():
            try:
                box["host"] = self._read(shard, layout, total)
            except BaseException as exc:          # surfaced by take()
                box["error"] = exc
        t = threading.Thread(target=run, name="shardmerge-base-read", daemon=True)
        t.start()
        self.pending[shard] = (t, box)

    def take(self, shard: str, layout, total: int) -> torch.Tensor:
        if shard not in self.pending:
            return self._read(shard, layout, total)
        t, box = self.pending.pop(shard)
        t.join()
        if "error" in box:
            raise box["error"]
        return box["host"]

    def close(self):
        self.pool.shutdown(wait=True)


async def run_partitioned_merge(config: MergeConfig, index: LocalModelIndex, device: str):
    """One rank of the N-rank merge.  Per base shard, in file order:
      1. the root (shard index mod N) reads the shard's block tensors and ONE broadcast (RCCL over
         xGMI) hands them to every rank - the only collective of the data path, no reduction;
      2. every rank merges the tensors the plan (LPT on algorithmic bytes) gave it; its finetune
         tensors are prefetched (loader.py);
      3. results leave for the shard's writer rank (= the root): its own through the asynchronous
         pinned copy of ModelWriter, the other ranks' with one point-to-point message each (device
         to device) - no part files, every tensor is written to disk once.
    A shard whose output file is already complete is skipped by all ranks (resume)."""
    world, me = world_size(), rank()
    if ENGINE_FACTORY is not None:
        engine = ENGINE_FACTORY()
    else:
        from .engine import get_engine
        torch.cuda.set_device(local_rank())
        engine = get_engine(f"cuda:{local_rank()}")
    dev = engine.device
    on_gpu = dev.type == "cuda"
    use_dist = world > 1 or os.environ.get("SHARDMERGE_FORCE_DIST") == "1"
    dist = init_process_group(dev) if use_dist else None

    from .merge import operator_class
    merger = operator_class(getattr(config, "operator", "fourier"))(config=config, index_manager=index, engine=engine)
    fourier = isinstance(merger, FourierMerge)          # the other operators fetch the base themselves: nothing to broadcast
    await merger.initialize()
    base_uri = config.output_base_model
    layer_order = index.get_layer_order(base_uri)
    rank_of = {n: i for i, n in enumerate(layer_order)}
    weight_map = index.model_indexes[base_uri]["weight_map"]
    shards = sorted(set(weight_map.values()))
    out_dir = config.output_path
    out_dir.mkdir(parents=True, exist_ok=True)
    # leftovers of an earlier (crashed) run must never reach this run's shards
    if me == 0:
        for stale in list(out_dir.glob(".part-*")) + list(out_dir.glob(".tmp-*")):
            stale.unlink()
    if dist is not None:
        dist.barrier()

    # ---- plan: the same on every rank --------------------------------------------------
    metas = {s: _tensor_meta(index, base_uri, s) for s in shards}
    names_of = {s: sorted((n for n in weight_map if weight_map[n] == s), key=rank_of.get) for s in shards}
    done = {s: _shard_complete(out_dir / s, names_of[s]) for s in shards}
    names, costs = [], []
    for s in shards:
        if done[s]:
            continue
        for name in names_of[s]:
            number = ShardLayer(rank_of[name], s, name, False).layer_number
            numel = 1
            for d in metas[s][name][0]:
                numel *= d
            if number in (INPUT_LAYER, OUTPUT_LAYER) or not fourier:
                cost = 2 * numel * (1 if fourier else len(config.finetune_merge) + 2)      # a copy / one streaming pass
            else:
                k = sum(1 for m in config.finetune_merge if m.use_layer_index(number))
                cost = alg_bytes(numel, k)
            names.append((s, name))
            costs.append(cost)
    owner = partition_lpt(costs, world)
    owner_of = {names[i]: owner[i] for i in range(len(names))}
    mine = {key for key, o in owner_of.items() if o == me}
    logger.info(f"rank {me}/{world}: {len(mine)} of {len(names)} tensors, "
                f"{sum(c for c, o in zip(costs, owner) if o == me) / 1e9:.2f} GB algorithmic traffic"
                + (f"; {sum(done.values())} shard(s) already complete" if any(done.values()) else ""))

    # ---- this rank's finetune tensors are prefetched in processing order (loader.py); the base
    # comes from the broadcast, passthrough tensors from whichever model provides them
    todo = [s for s in shards if not done[s]]
    my_order = [(s, name) for s in todo for name in names_of[s] if (s, name) in mine]
    schedule = []
    for (s, name) in my_order:
        sl = ShardLayer(rank_of[name], s, name, False)
        if sl.layer_number >= 0 and fourier:
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

    # the shards this rank writes: a buffered, asynchronous ModelWriter over exactly those
    root_of = {s: si % world for si, s in enumerate(shards)}
    my_shards = [s for s in todo if root_of[s] == me]
    writer = None
    if my_shards:
        sub_index = {"metadata": index.model_indexes[base_uri].get("metadata", {}),
                     "weight_map": {n: s for s in my_shards for n in names_of[s]}}
        writer = ModelWriter(base_index=sub_index, output_path=out_dir, layer_order=layer_order,
                             output_astype=config.output_astype, write_index=False)

    def block_layout(s):
        block = [n for n in names_of[s] if ShardLayer(0, s, n, False).layer_number >= 0] if fourier else []
        offs, total = {}, 0
        for n in block:                                   # 256-byte aligned slots
            offs[n] = (total, metas[s][n][2])
            total += (metas[s][n][2] + 255) // 256 * 256
        return block, offs, total

    reader = _BaseShardReader(index, base_uri, pinned=on_gpu)
    mine_as_root = [s for s in todo if root_of[s] == me]
    if mine_as_root:
        _, offs0, total0 = block_layout(mine_as_root[0])
        reader.start(mine_as_root[0], offs0, total0)

    try:
        for s in todo:
            root = root_of[s]
            block, offs, total = block_layout(s)
            flat = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
            if me == root and total:
                host = reader.take(s, offs, total)
                flat.copy_(host, non_blocking=True)
                nxt = [q for q in mine_as_root if q > s]
                if nxt:                                   # read the next base shard during this merge
                    _, offs_n, total_n = block_layout(nxt[0])
                    reader.start(nxt[0], offs_n, total_n)
            if dist is not None and total:
                dist.broadcast(flat, src=root)            # THE collective: RCCL over xGMI
            views = {}
            for n in block:
                shape, dt, nbytes = metas[s][n]
                views[n] = flat[offs[n][0]:offs[n][0] + nbytes].view(_ST_DTYPES[dt]).reshape(shape)

            merged: Dict[str, torch.Tensor] = {}
            for name in names_of[s]:
                if (s, name) not in mine:
                    continue
                sl = ShardLayer(rank_of[name], s, name, False)
                if loader is not None:
                    loader.begin_layer(pos[(s, name)])
                if sl.layer_number >= 0 and fourier:
                    out = await _merge_block_tensor(merger, engine, sl, views[name])
                else:
                    out = await merger._merge_layer(sl, str(dev))
                out = out.detach().to(config.output_astype)
                if me == root:
                    writer.add_tensor(name, out)          # asynchronous pinned copy, written with the shard
                else:
                    merged[name] = out
            # results of the other ranks travel to the writer rank, one message per rank
            if dist is not None and world > 1:
                for r in range(world):
                    theirs = [n for n in names_of[s] if owner_of[(s, n)] == r]
                    if r == root or not theirs:
                        continue
                    sizes = [metas[s][n][0] for n in theirs]
                    numels = [int(torch.Size(sh).numel()) for sh in sizes]
                    if me == r:
                        buf = torch.cat([merged[n].reshape(-1) for n in theirs]) if len(theirs) > 1 else merged[theirs[0]].reshape(-1).contiguous()
                        dist.send(buf, dst=root)
                    elif me == root:
                        buf = torch.empty(sum(numels), dtype=config.output_astype, device=dev)
                        dist.recv(buf, src=r)
                        o = 0
                        for n, sh, ne in zip(theirs, sizes, numels):
                            writer.add_tensor(n, buf[o:o + ne].view(sh))
                            o += ne
            del flat, views, merged
    finally:
        reader.close()
        if loader is not None:
            loader.close()
            merger._loader = None

    if writer is not None:
        writer.finalize()                                 # raises if one of this rank's shards is incomplete
    if dist is not None:
        dist.barrier()
    if me == 0:
        with open(out_dir / "model.safetensors.index.json", "w") as fh:
            json.dump(index.model_indexes[base_uri], fh, indent=2)
        with open(out_dir / "README.md", "w") as fh:
            fh.write(merger.get_readme())
    if dist is not None:
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
                                     cutoff_pct=merger.cutoff_pct, t_sum=merger.t_sum, b=merger.b, norm_mode=merger.norm_mode,
                                     layer_name=sl.layer_name)
    merger.last_report = report
    return out
