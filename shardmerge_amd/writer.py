"""Output writer: merged tensors go into safetensors shards that mirror the base
model's layout.

Same contract as the reference's ModelWriter (shard/writer.py:60-180): the
output directory gets the base model's ``model.safetensors.index.json``, shard
files named like the base's, tensors ordered by ``layer_order``, a resume scan
of shards that already exist and a completeness check in ``finalize``.  One
difference in mechanism: the reference re-reads and re-writes the
whole shard for every tensor (O(T^2) bytes); here a shard's tensors are buffered
and the file is written once when the shard is complete (or at ``finalize`` /
``flush`` for a partial shard), which keeps resume working at shard granularity.
"""
from __future__ import annotations

import json
import os
import queue
import threading
import logging
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Generator, List, Set, Tuple

import torch

from . import iostats, stformat
from safetensors import safe_open
from safetensors.torch import save_file

from .constants import INPUT_LAYER, OUTPUT_LAYER

logger = logging.getLogger(__name__)


@dataclass
class ShardLayer:
    layer_order_idx: int
    shard_name: str
    layer_name: str
    written: bool

    @property
    def layer_number(self) -> int:
        """-1 for the embedding, -2 for final norm / lm_head, N for model.layers.N.*;
        anything else is an error (reference writer.py:39-57)."""
        name = self.layer_name
        if name.startswith("model.embed_tokens.weight"):
            return INPUT_LAYER
        if name.startswith("model.norm.weight") or name.startswith("lm_head.weight"):
            return OUTPUT_LAYER
        if name.startswith("model.layers."):
            field_ = name.split(".")[2]
            number = int(field_)
            if str(number) == field_:
                return number
        raise ValueError(f"Unknown layer name: {name}")


@dataclass
class ModelWriter:
    base_index: dict
    output_path: Path
    layer_order: List[str]
    output_astype: torch.dtype
    written_shard_layers: Set[Tuple[str, str]] = field(default_factory=set)
    shard_to_tensors: Dict[str, Set[str]] = field(default_factory=dict)
    write_index: bool = True        # False: a rank of the multi-GPU merge writes only its shards (rank 0 writes the index)

    def __post_init__(self):
        self.output_path = Path(self.output_path)
        self.output_path.mkdir(parents=True, exist_ok=True)
        self.index_path = self.output_path / "model.safetensors.index.json"
        if not self.write_index:
            pass
        elif self.index_path.exists():
            logger.info(f"Index already exists: {self.index_path}")
            with open(self.index_path) as fh:
                self.base_index = json.load(fh)
        else:
            with open(self.index_path, "w") as fh:
                json.dump(self.base_index, fh, indent=2)
        self.shard_to_tensors = {}
        for tensor_name, shard_name in self.base_index["weight_map"].items():
            self.shard_to_tensors.setdefault(shard_name, set()).add(tensor_name)
        self._rank = {name: i for i, name in enumerate(self.layer_order)}
        self._pending: Dict[str, Dict[str, tuple]] = {}
        self._lock = threading.Lock()
        self._jobs: "queue.Queue" = queue.Queue(maxsize=4)      # at most four shards waiting to be written
        self._thread = None
        self._worker_error = None
        self._check_existing_shards()

    # -- resume ---------------------------------------------------------------------
    def _check_existing_shards(self):
        for shard_name, expected in self.shard_to_tensors.items():
            path = self.output_path / shard_name
            if not path.exists():
                continue
            with safe_open(str(path), framework="pt") as fh:
                for name in fh.keys():
                    if name not in expected:
                        raise ValueError(f"Tensor {name} found in {path} but not in base model")
                    self.written_shard_layers.add((shard_name, name))

    # -- writing ----------------------------------------------------------------------
    # Device results leave through a pinned buffer with an asynchronous copy on the caller's
    # stream, and a complete shard is serialised by a background thread, so neither the
    # device-to-host copy nor save_file() stalls the merge of the next tensors.
    def add_tensor(self, layer_name: str, tensor: torch.Tensor):
        shard_name = self.base_index["weight_map"][layer_name]
        if (shard_name, layer_name) in self.written_shard_layers:
            logger.info(f"Skipping {layer_name} as it's already in written shard {shard_name}")
            return
        self._raise_worker_error()
        # the device -> host copy and the cast happen here, as in writer.py:133
        t = tensor.detach()
        if t.device.type == "cuda":
            t = t.to(self.output_astype).contiguous()
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(t.device))
            entry = (host, ev)
        else:
            entry = (t.to("cpu").to(self.output_astype).contiguous(), None)
        self._pending.setdefault(shard_name, {})[layer_name] = entry
        with self._lock:
            have = {n for (s, n) in self.written_shard_layers if s == shard_name} | set(self._pending[shard_name])
        if have >= self.shard_to_tensors[shard_name]:
            self.flush(shard_name)

    def _raise_worker_error(self):
        if self._worker_error is not None:
            err, self._worker_error = self._worker_error, None
            raise err

    def _write_shard(self, name: str, fresh):
        path = self.output_path / name
        merged: Dict[str, torch.Tensor] = {}
        if path.exists():                       # partial shard from an earlier run
            with safe_open(str(path), framework="pt") as fh:
                for k in fh.keys():
                    merged[k] = fh.get_tensor(k)
        for k, (t, ev) in fresh.items():
            if ev is not None:
                with iostats.timed("d2h_wait", t.numel() * t.element_size()):
                    ev.synchronize()
            merged[k] = t
        ordered = {k: merged[k] for k in sorted(merged, key=lambda k: self._rank.get(k, len(self._rank)))}
        tmp = path.with_name(f".tmp-{path.name}")
        nbytes = sum(t.numel() * t.element_size() for t in ordered.values())
        with iostats.timed("save", nbytes):
            if all(t.dtype in stformat.ST_NAMES and t.is_contiguous() for t in ordered.values()):
                # the file is laid out by hand (header bytes identical to safetensors' own writer, pinned by
                # tests/test_distributed_gloo.py) and the tensors go from their pinned buffers straight to their
                # offsets, several pieces at a time: save_file() serialises through one more copy on one thread
                head, offsets = stformat.shard_header([(k, stformat.ST_NAMES[t.dtype], list(t.shape)) for k, t in ordered.items()],
                                                      {"format": "pt"})
                fd = os.open(tmp, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
                try:
                    os.pwrite(fd, head, 0)
                    os.ftruncate(fd, len(head) + nbytes)
                    # (pwrite in pieces: 2x a shared mapping's page faults on tmpfs; writes to ONE file still serialise
                    #  on its inode - 4.8 GB/s on the MI355X box - which is why several shards are written at a time)
                    stformat.pwrite_tensors(fd, len(head), offsets, ordered, self._io_pool())
                finally:
                    os.close(fd)
            else:
                save_file(ordered, str(tmp), metadata={"format": "pt"})
        os.replace(tmp, path)                  # resume must never see a half-written shard
        with self._lock:
            for k in fresh:
                self.written_shard_layers.add((name, k))
        logger.info(f"Wrote {len(fresh)} tensor(s) to shard {name}")

    def _io_pool(self):
        if getattr(self, "_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=int(os.environ.get("SHARDMERGE_WRITE_THREADS", "8")),
                                            thread_name_prefix="shardmerge-write")
        return self._pool

    def _worker(self):
        while True:
            job = self._jobs.get()
            if job is None:
                self._jobs.task_done()
                return
            try:
                self._write_shard(*job)
            except BaseException as exc:       # surfaced by the next add_tensor()/flush()/finalize()
                self._worker_error = exc
            finally:
                self._jobs.task_done()

    def flush(self, shard_name: str = None, wait: bool = False):
        """Hand buffered tensors (of one shard, or of all) to the writer thread."""
        for name in ([shard_name] if shard_name else list(self._pending)):
            fresh = self._pending.pop(name, None)
            if not fresh:
                continue
            if self._thread is None:
                # several shards in flight: one file's writes serialise in the kernel (its inode lock; tmpfs: page
                # allocation), different files' do not
                n = max(1, int(os.environ.get("SHARDMERGE_SHARD_WRITERS", "3")))
                self._thread = [threading.Thread(target=self._worker, name=f"shardmerge-writer-{i}", daemon=True) for i in range(n)]
                for th in self._thread:
                    th.start()
            self._jobs.put((name, fresh))
        if wait or shard_name is None:
            self._jobs.join()
            self._raise_worker_error()

    def wait(self):
        """Block until every shard handed to the writer thread is on disk."""
        self._jobs.join()
        self._raise_worker_error()

    def finalize(self):
        self.flush()
        if self._thread is not None:
            for _ in self._thread:
                self._jobs.put(None)
            for th in self._thread:
                th.join()
            self._thread = None
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._raise_worker_error()
        missing = [(s, n) for s, names in self.shard_to_tensors.items() for n in names
                   if (s, n) not in self.written_shard_layers]
        if missing:
            logger.error(f"Failed to write all layers. Missing: {missing}")
            raise RuntimeError(f"Incomplete model output: missing {len(missing)} layers")

    # -- iteration ----------------------------------------------------------------------
    def shard_layers(self) -> Generator[List[ShardLayer], None, None]:
        """Shards in file-name order; inside a shard, tensors in layer_order order."""
        for shard_name in sorted(self.shard_to_tensors):
            names = sorted(self.shard_to_tensors[shard_name], key=lambda n: self.layer_order.index(n))
            group = []
            for name in names:
                sl = ShardLayer(self.layer_order.index(name), shard_name, name, (shard_name, name) in self.written_shard_layers)
                sl.layer_number            # raises on unknown names, as the reference does while iterating
                group.append(sl)
            yield group
