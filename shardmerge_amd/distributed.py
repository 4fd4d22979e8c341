"""Multi-GPU merge: one process per GPU (torchrun), tensors partitioned over ranks.

The path shards by independent units: every output tensor depends
only on the same-named tensors of the base and the finetunes, so there is no
data-path collective.  What IS shared is the base model: per base shard file,
one rank reads it from disk and ONE RCCL broadcast hands its payload to every
rank over xGMI (`torch.distributed` backend "nccl" is RCCL on ROCm); each rank
then merges the tensors it owns, reading only its own finetune tensors.

Nothing else travels between ranks:

* the broadcasts are issued ahead, asynchronously, in shard order, up to
  ``SHARDMERGE_BCAST_WINDOW`` shards (default 3) in front of the shard a rank
  is merging - a rank waits for a shard's payload only when it gets there, so
  ranks are not in lock-step (a shard holds a handful of tensors of very
  different sizes: per shard no assignment is balanced, over a few it is);
* the plan hands out tensors in shard order to the rank with the least work so
  far (cost = estimated milliseconds per shape from measured kernel times), so
  every PREFIX of the model is balanced and the ranks stay within the window;
* every rank writes its own results: the output shard is a pre-sized
  safetensors file (header from the plan) and a rank's tensors go to their
  offsets with positional writes - no result ever crosses to a "writer rank".
  The rank that completes a shard (a counter in the process group's store)
  renames it into place; resume sees only complete shards.
"""
from __future__ import annotations

import asyncio
import hashlib
import json
import logging
import os
import queue
import threading
import time
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from .config import MergeConfig
from .constants import DEFAULT_NORM_MODE, INPUT_LAYER, OUTPUT_LAYER
from .index import LocalModelIndex
from .merge.fast_fourier import FourierMerge
from .writer import ShardLayer

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


# per-layer time on MI355X: a fixed part (launches; more of them per pair merge) plus a streaming part at the
# pipeline's measured rate, slower where a length needs the long 28672-point plans (profiles/r02_kprof_final.txt,
# r03: 28672x8192 K=3 5.8 ms, 8192^2 K=3 1.83 / K=2 1.2, 1024x8192 K=3 0.56, 1-D K=3 0.2)
def est_ms(shape: Sequence[int], k: int) -> float:
    """Estimated milliseconds to merge one block tensor of `shape` with k finetunes (the plan's cost)."""
    numel = 1
    for d in shape:
        numel *= int(d)
    if k <= 1:
        return 0.02 + 8.0 * numel / 4.0e9
    pairs = k - 1
    if len(shape) < 2 or min(int(d) for d in shape[-2:]) == 1:
        return 0.1 * pairs
    rate = 5.0e9 if max(int(d) for d in shape[-2:]) <= 16384 else 4.2e9        # canonical bytes per millisecond
    return 0.15 + 0.2 * pairs + alg_bytes(numel, k) / rate


def partition_lpt(costs: List[float], world: int) -> List[int]:
    """Longest-processing-time-first assignment; deterministic (ties by index)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += costs[i]
    return owner


def partition_in_order(costs: List[float], groups: List[int], world: int) -> List[int]:
    """List scheduling in GROUP (shard) order: groups in ascending order, inside a group the largest first,
    each to the rank with the least work so far.  Balanced over every prefix of the groups (within one
    largest item), which is what lets the ranks run a bounded number of shards apart."""
    order = sorted(range(len(costs)), key=lambda i: (groups[i], -costs[i], i))
    load = [0.0] * world
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


from .stformat import ST_DTYPES as _ST_DTYPES, ST_NAMES as _ST_NAMES, ST_SIZE as _ST_SIZE, shard_header  # noqa: E402,F401


def init_process_group(device: torch.device):
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # one rank per GPU over RCCL.  (Rehearsal on a one-GPU box: SHARDMERGE_DIST_BACKEND=gloo lets several ranks share the
    # card - RCCL refuses two ranks on one device; the base shards then travel through host memory.)
    backend = os.environ.get("SHARDMERGE_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return dist


# ---- output shards written in place ------------------------------------------------------------------------
def _read_metadata(path: Path):
    try:
        with open(path, "rb") as fh:
            n = int.from_bytes(fh.read(8), "little")
            header = json.loads(fh.read(n))
    except Exception:
        return None, None
    return header.get("__metadata__", {}), {k for k in header if k != "__metadata__"}


def _shard_complete(path: Path, expected, stamp: Optional[str] = None) -> bool:
    """An output shard that exists with exactly the expected tensors (shards appear by an atomic rename) and,
    when a stamp is given, was written by a run with the same models and options."""
    if not path.exists():
        return False
    meta, names = _read_metadata(path)
    if names is None or names != set(expected):
        return False
    return stamp is None or meta.get("shardmerge_config") == stamp


def config_stamp(config: MergeConfig) -> str:
    """What decides a shard's content: models, layer windows, weights, flags, operator, options, output dtype."""
    from dataclasses import asdict
    doc = {"output_base_model": config.output_base_model, "output_dtype": config.output_dtype,
           "finetune_merge": [asdict(m) for m in config.finetune_merge],
           "merge_options": dict(sorted((config.merge_options or {}).items())),
           "operator": getattr(config, "operator", "fourier"), "norm_mode": getattr(config, "norm_mode", None) or DEFAULT_NORM_MODE}
    return hashlib.sha256(json.dumps(doc, sort_keys=True, default=str).encode()).hexdigest()[:16]


class _InPlaceShardWriter:
    """This rank's side of the output: results are copied to pinned memory asynchronously and a thread writes
    them at their offsets of the pre-sized shard files (`.tmp-<shard>`, created by rank 0 from the plan)."""

    def __init__(self, out_dir: Path, plans: Dict[str, dict], astype: torch.dtype, on_gpu: bool, on_written):
        self.out_dir, self.plans, self.astype, self.on_gpu = out_dir, plans, astype, on_gpu
        self.on_written = on_written                 # (shard) -> None, after each tensor of it is on disk
        self.fds: Dict[str, int] = {}
        self.jobs: "queue.Queue" = queue.Queue(maxsize=8)
        self.error: Optional[BaseException] = None
        self.thread = threading.Thread(target=self._run, name="shardmerge-writer", daemon=True)
        self.thread.start()

    def add(self, shard: str, name: str, tensor: torch.Tensor):
        if self.error is not None:
            raise self.error
        plan = self.plans[shard]
        t = tensor.detach().to(self.astype).contiguous()
        if list(t.shape) != list(plan["shapes"][name]):
            raise ValueError(f"{name}: merged shape {list(t.shape)} but the output shard was laid out for {plan['shapes'][name]}")
        ev = None
        if t.device.type == "cuda":
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(t.device))
        else:
            host = t.to("cpu")
        self.jobs.put((shard, name, host, ev))

    def _run(self):
        while True:
            job = self.jobs.get()
            try:
                if job is None:
                    return
                if self.error is not None:
                    continue
                shard, name, host, ev = job
                if ev is not None:
                    ev.synchronize()
                plan = self.plans[shard]
                if shard not in self.fds:
                    self.fds[shard] = os.open(self.out_dir / f".tmp-{shard}", os.O_WRONLY)
                view = memoryview(host.reshape(-1).view(torch.uint8).numpy()).cast("B")      # (reshape first: a 0-dim tensor has no byte view)
                pos, done = plan["data_start"] + plan["offsets"][name][0], 0
                while done < len(view):
                    done += os.pwrite(self.fds[shard], view[done:done + (1 << 30)], pos + done)
                self.on_written(shard)
            except BaseException as exc:       # surfaced by the next add() / close()
                self.error = exc
            finally:
                self.jobs.task_done()

    def close(self, abort: bool = False):
        """join the writer thread and close the files; abort=True (another error is already on its way up): no fsync,
        and a write error found here does not mask the first one"""
        if self.thread.is_alive():
            self.jobs.put(None)
            self.thread.join()
        for fd in self.fds.values():
            try:
                if not abort:
                    os.fsync(fd)
                os.close(fd)
            except OSError:
                if not abort:
                    raise
        self.fds.clear()
        if self.error is not None and not abort:
            raise self.error


class _BaseShardReader:
    """The root rank's side of the base-shard broadcast: the shard's block tensors are read with
    positional reads (4 threads) into ONE pinned buffer and go to the device with ONE copy; the
    next shards this rank is root of are read in the background."""

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
        if shard in self.pending:
            return
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
    """One rank of the N-rank merge (module docstring).  Per base shard, in file order, the root (shard index
    mod N) reads the shard's block tensors and ONE broadcast (RCCL over xGMI) hands them to every rank - the
    only collective of the data path, no reduction, issued asynchronously a few shards ahead; every rank merges
    the tensors the plan gave it (its finetune tensors prefetched by loader.py) and writes them into the output
    shard in place.  A shard whose output file is already complete is skipped by all ranks (resume)."""
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
    store = None
    if dist is not None:
        from torch.distributed import distributed_c10d
        store = distributed_c10d._get_default_store()

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
    stamp = config_stamp(config)
    run_id = ""

    # ---- plan: made by rank 0, the same on every rank ---------------------------------------------
    metas = {s: _tensor_meta(index, base_uri, s) for s in shards}
    names_of = {s: sorted((n for n in weight_map if weight_map[n] == s), key=rank_of.get) for s in shards}
    if me == 0:
        # leftovers of an earlier (crashed) run must never reach this run's shards
        for stale in list(out_dir.glob(".part-*")) + list(out_dir.glob(".tmp-*")):
            stale.unlink()
        done_list = [_shard_complete(out_dir / s, names_of[s], stamp) for s in shards]
        run_id = hashlib.sha256(f"{time.time_ns()}-{os.getpid()}".encode()).hexdigest()[:12]
    else:
        done_list = [False] * len(shards)
    if dist is not None:
        box = [done_list, run_id]
        dist.broadcast_object_list(box, src=0)          # ONE decision (a rank scanning later would see other files)
        done_list, run_id = box
    done = dict(zip(shards, done_list))
    todo = [s for s in shards if not done[s]]

    # shapes of the outputs: block tensors have the base's, a passthrough tensor (embedding, final norm, lm_head)
    # has its PROVIDER's - a finetune flagged is_input / is_output may bring an extended vocabulary
    provider_meta: Dict[Tuple[str, str], Dict[str, tuple]] = {}

    def out_shape(s: str, name: str) -> List[int]:
        sl = ShardLayer(rank_of[name], s, name, False)
        if fourier and sl.layer_number in (INPUT_LAYER, OUTPUT_LAYER):
            uri = merger._layer_requests(sl)[0][0]
            if uri != base_uri:
                f = index.model_indexes[uri]["weight_map"][name]
                if (uri, f) not in provider_meta:
                    provider_meta[(uri, f)] = _tensor_meta(index, uri, f)
                return list(provider_meta[(uri, f)][name][0])
        return list(metas[s][name][0])

    names, costs, groups = [], [], []
    for si, s in enumerate(shards):
        if done[s]:
            continue
        for name in names_of[s]:
            number = ShardLayer(rank_of[name], s, name, False).layer_number
            shape = out_shape(s, name)
            numel = 1
            for d in shape:
                numel *= d
            if number in (INPUT_LAYER, OUTPUT_LAYER) or not fourier:
                cost = 0.02 + 2.0 * numel * (1 if fourier else len(config.finetune_merge) + 2) / 4.0e9   # a copy / one streaming pass
            else:
                k = sum(1 for m in config.finetune_merge if m.use_layer_index(number))
                cost = est_ms(shape, k)
            names.append((s, name))
            costs.append(cost)
            groups.append(si)
    owner = partition_in_order(costs, groups, world)
    owner_of = {names[i]: owner[i] for i in range(len(names))}
    mine = {key for key, o in owner_of.items() if o == me}
    logger.info(f"rank {me}/{world}: {len(mine)} of {len(names)} tensors, "
                f"{sum(c for c, o in zip(costs, owner) if o == me):.1f} of {sum(costs):.1f} ms estimated"
                + (f"; {sum(done.values())} shard(s) already complete" if any(done.values()) else ""))

    # ---- output shards: laid out from the plan, created by rank 0, filled in place by every rank ----
    astype_name = _ST_NAMES[config.output_astype]
    plans: Dict[str, dict] = {}
    for s in todo:
        shapes = {n: out_shape(s, n) for n in names_of[s]}
        head, offsets = shard_header([(n, astype_name, shapes[n]) for n in names_of[s]], {"format": "pt", "shardmerge_config": stamp})
        plans[s] = {"head": head, "offsets": offsets, "data_start": len(head), "shapes": shapes,
                    "size": len(head) + max((e for _, e in offsets.values()), default=0),
                    "owners": sorted({owner_of[(s, n)] for n in names_of[s]})}
    if me == 0:
        for s in todo:
            with open(out_dir / f".tmp-{s}", "wb") as fh:
                fh.write(plans[s]["head"])
                fh.truncate(plans[s]["size"])
    if dist is not None:
        dist.barrier()

    left_of = {s: sum(1 for n in names_of[s] if (s, n) in mine) for s in todo}
    finished: List[str] = []

    def tensor_written(s: str):
        left_of[s] -= 1
        if left_of[s] == 0:
            finished.append(s)

    writer = _InPlaceShardWriter(out_dir, plans, config.output_astype, on_gpu, tensor_written)

    def publish_finished():
        # a rank that has written its last tensor of a shard counts itself in; the one that completes the
        # count renames the shard into place (a counter in the process group's store: no collective)
        while finished:
            s = finished.pop()
            os.fsync(writer.fds[s])
            n_done = store.add(f"shardmerge/{run_id}/{s}", 1) if store is not None else len(plans[s]["owners"])
            if n_done == len(plans[s]["owners"]):
                os.replace(out_dir / f".tmp-{s}", out_dir / s)
                logger.info(f"rank {me}: shard {s} complete")

    # ---- this rank's finetune tensors are prefetched in processing order (loader.py); the base
    # comes from the broadcast, passthrough tensors from whichever model provides them
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

    root_of = {s: si % world for si, s in enumerate(shards)}

    def block_layout(s):
        block = [n for n in names_of[s] if ShardLayer(0, s, n, False).layer_number >= 0] if fourier else []
        offs, total = {}, 0
        for n in block:                                   # 256-byte aligned slots
            offs[n] = (total, metas[s][n][2])
            total += (metas[s][n][2] + 255) // 256 * 256
        return block, offs, total

    window = max(1, int(os.environ.get("SHARDMERGE_BCAST_WINDOW", "3")))
    reader = _BaseShardReader(index, base_uri, pinned=on_gpu)
    mine_as_root = [s for s in todo if root_of[s] == me]
    for s in mine_as_root[:2]:
        _, offs0, total0 = block_layout(s)
        reader.start(s, offs0, total0)
    inflight: Dict[str, tuple] = {}                       # shard -> (device buffer, work handle, block, offsets)
    issued = 0
    via_host = dist is not None and on_gpu and dist.get_backend() != "nccl"      # (the gloo rehearsal above)

    def issue_broadcasts(upto: int):
        """payloads of todo[issued : upto] on their way: every rank issues the same broadcasts in the same order"""
        nonlocal issued
        while issued < min(upto, len(todo)):
            s = todo[issued]
            issued += 1
            block, offs, total = block_layout(s)
            flat = torch.empty(max(total, 1), dtype=torch.uint8, device="cpu" if via_host else dev, pin_memory=via_host)
            work = None
            if total:
                if me == root_of[s]:
                    flat.copy_(reader.take(s, offs, total), non_blocking=True)
                    nxt = [q for q in mine_as_root if q > s][:2]
                    for q in nxt:                         # read the next base shards of this root meanwhile
                        _, offs_n, total_n = block_layout(q)
                        reader.start(q, offs_n, total_n)
                if dist is not None and world > 1:
                    work = dist.broadcast(flat, src=root_of[s], async_op=True)      # THE collective: RCCL over xGMI
            inflight[s] = (flat, work, block, offs)

    # A rank that fails says so in the process group's store before it goes down: the others look there at every
    # shard and in front of the barrier and stop with an error of their own instead of waiting for ever.  (A rank
    # blocked INSIDE a collective whose root has died is beyond this: run with a process-group timeout.)  The output
    # directory must be a node-local or coherent POSIX file system: several ranks pwrite() into one shard file.
    fail_key = f"shardmerge/{run_id}/failed"

    def peers_alive():
        if store is not None and world > 1 and store.add(fail_key, 0) > 0:
            raise RuntimeError(f"rank {me}: another rank of this merge has failed (see its log); stopping")

    t_wait = t_merge = 0.0
    t_start = time.time()
    ok = False
    try:
        for si, s in enumerate(todo):
            peers_alive()
            issue_broadcasts(si + window)
            flat, work, block, offs = inflight.pop(s)
            if any((s, n) in mine for n in names_of[s]):
                t0 = time.time()
                if work is not None:
                    work.wait()
                t_wait += time.time() - t0
                if via_host:
                    flat = flat.to(dev, non_blocking=True)
                views = {}
                for n in block:
                    shape, dt, nbytes = metas[s][n]
                    views[n] = flat[offs[n][0]:offs[n][0] + nbytes].view(_ST_DTYPES[dt]).reshape(shape)
                t0 = time.time()
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
                    writer.add(s, name, out)              # asynchronous pinned copy, written at its offset
                    del out
                    publish_finished()
                if on_gpu:
                    torch.cuda.current_stream(dev).synchronize()
                t_merge += time.time() - t0
                del views
            elif work is not None:
                work.wait()                               # (the buffer must outlive the collective)
            del flat
        writer.jobs.join()
        publish_finished()
        writer.close()
        ok = True
    finally:
        reader.close()
        if loader is not None:
            loader.close()
            merger._loader = None
        if not ok:
            # this rank is going down: tell the others, then release the writer thread and its files (the .tmp-* shards
            # stay behind for the next run, which recreates them)
            try:
                if store is not None and world > 1:
                    store.add(fail_key, 1)
            except Exception:
                pass
            writer.close(abort=True)
    busy = time.time() - t_start
    logger.info(f"rank {me}: merged {len(mine)} tensors in {busy:.2f} s; waited {t_wait:.2f} s for base shards "
                f"(idle {100.0 * t_wait / max(busy, 1e-9):.1f} %)")

    peers_alive()
    if dist is not None:
        dist.barrier()
    if me == 0:
        missing = [s for s in todo if not (out_dir / s).exists()]
        if missing:
            if store is not None and world > 1:
                store.add(fail_key, 1)
            raise RuntimeError(f"Incomplete model output: shards {missing} were not completed")
        with open(out_dir / "model.safetensors.index.json", "w") as fh:
            json.dump(index.model_indexes[base_uri], fh, indent=2)
        with open(out_dir / "README.md", "w") as fh:
            fh.write(merger.get_readme())
    if dist is not None:
        dist.barrier()
    # what this rank did (bench.py --product-path gathers these): tensors and output bytes it owned, the wall time
    # of its loop, how much of it it spent waiting for a base shard's broadcast and how much inside merges
    return {"rank": me, "tensors": len(mine), "out_bytes": sum(int(metas[s_][n_][2]) for (s_, n_) in mine if n_ in metas[s_]),
            "loop_s": busy, "wait_base_s": t_wait, "merge_s": t_merge, "shards": len(todo)}


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
