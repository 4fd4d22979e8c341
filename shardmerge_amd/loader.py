"""Prefetching tensor loader (replaces the reference's
``HFMultiModelIndex.get_tensor/_load_tensor``, shard/index.py:195-270, on the merge path).

The reference opens a safetensors shard per tensor, keeps every tensor it ever loaded in an
unbounded RAM cache and loads strictly on demand, so the device waits for the disk and the
host for the device.  Here a background thread walks the merge's own schedule (the tensors
of layer i+1, i+2 while layer i is being merged): it reads each tensor's bytes straight from
the shard file into a pinned staging buffer (headers are parsed once per shard, no
intermediate copy), issues the host-to-device copy on a dedicated copy stream and hands the
device tensor over with an event.  At most ``depth`` layers are in flight and nothing is
cached: a tensor is dropped as soon as the merge has taken it.
"""
from __future__ import annotations

import json
import logging
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import iostats

logger = logging.getLogger(__name__)

_ST_DTYPES = {
    "BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32, "F64": torch.float64,
    "I64": torch.int64, "I32": torch.int32, "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8, "BOOL": torch.bool,
}

Request = Tuple[str, str]          # (model uri, tensor name)
_READ_PIECE = 32 << 20


class ShardFile:
    """One safetensors file: header parsed once, payload read by offset."""

    def __init__(self, path: Path):
        self.path = Path(path)
        self.fh = open(self.path, "rb", buffering=0)
        n = int.from_bytes(self.fh.read(8), "little")
        header = json.loads(self.fh.read(n))
        self.data_start = 8 + n
        self.entries = {k: v for k, v in header.items() if k != "__metadata__"}
        self.lock = threading.Lock()

    def meta(self, name: str):
        rec = self.entries[name]
        return rec["shape"], _ST_DTYPES[rec["dtype"]], rec["data_offsets"][1] - rec["data_offsets"][0]

    def read_into(self, name: str, dst: torch.Tensor, pool: Optional[ThreadPoolExecutor] = None):
        """dst: contiguous uint8 CPU tensor of exactly the payload size.  pool: the payload is read in 32 MB pieces
        side by side (os.preadv releases the GIL; one thread copies ~8 GB/s out of the page cache)."""
        rec = self.entries[name]
        off0, off1 = rec["data_offsets"]
        view = memoryview(dst.numpy()).cast("B")
        fd, pos = self.fh.fileno(), self.data_start + off0

        def piece(a: int, b: int):
            got = a
            while got < b:                             # positional reads: safe from several threads
                k = os.preadv(fd, [view[got:min(b, got + (1 << 30))]], pos + got)
                if not k:
                    raise IOError(f"short read of {name} in {self.path}")
                got += k
        n = off1 - off0
        if pool is None or n <= _READ_PIECE:
            piece(0, n)
            return
        for f in [pool.submit(piece, a, min(n, a + _READ_PIECE)) for a in range(0, n, _READ_PIECE)]:
            f.result()

    def close(self):
        self.fh.close()


class PrefetchLoader:
    def __init__(self, index, device: str, depth: int = 2):
        depth = int(os.environ.get("SHARDMERGE_PREFETCH_DEPTH", depth))          # layers in flight
        self.index = index
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.depth = max(1, depth)
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.files: Dict[Path, ShardFile] = {}
        self.schedule: List[List[Request]] = []
        self.ready: Dict[Tuple[int, str, str], tuple] = {}
        self.cv = threading.Condition()
        self.taken_layers = 0          # layers whose tensors have all been handed over
        self.remaining: List[int] = []
        self.error: Optional[BaseException] = None
        self.thread: Optional[threading.Thread] = None
        self.stop = False
        self.cursor = 0                # layer the consumer is at
        self.bytes_read = 0
        self.pool = ThreadPoolExecutor(max_workers=int(os.environ.get("SHARDMERGE_READ_THREADS", "4")),
                                       thread_name_prefix="shardmerge-read")
        # ... and the pieces of one tensor's payload (a 28672 x 8192 bf16 tensor is 470 MB: one thread would copy it
        # at a sixth of what the PCIe link takes)
        self.piece_pool = ThreadPoolExecutor(max_workers=int(os.environ.get("SHARDMERGE_READ_PIECE_THREADS", "12")),
                                             thread_name_prefix="shardmerge-read-piece")

    # ---- producer ---------------------------------------------------------------------
    def start(self, schedule: Sequence[Sequence[Request]]):
        self.schedule = [list(dict.fromkeys(reqs)) for reqs in schedule]
        self.remaining = [len(r) for r in self.schedule]
        self.thread = threading.Thread(target=self._run, name="shardmerge-prefetch", daemon=True)
        self.thread.start()

    def _file(self, uri: str, name: str) -> ShardFile:
        path = self.index.shard_path(uri, name)
        f = self.files.get(path)
        if f is None:
            f = self.files[path] = ShardFile(path)
        return f

    def _load(self, uri: str, name: str):
        if self.on_gpu:
            torch.cuda.set_device(self.device)
        f = self.files[self.index.shard_path(uri, name)]
        shape, dtype, nbytes = f.meta(name)
        with iostats.timed("pinned_alloc"):
            host = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=self.on_gpu)[:nbytes]
        if nbytes:
            with iostats.timed("read", nbytes):          # (summed over the read threads)
                f.read_into(name, host, self.piece_pool)
        self.bytes_read += nbytes
        cpu = host.view(dtype).reshape(shape) if nbytes else torch.empty(shape, dtype=dtype)
        if not self.on_gpu:
            return cpu, None, None
        with torch.cuda.stream(self.copy_stream):
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(self.copy_stream)
            dev = cpu.to(self.device, non_blocking=True)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self.copy_stream)
        return dev, ev, (host, ev0, nbytes)          # host stays referenced until the copy has been waited for

    def _run(self):
        try:
            if self.on_gpu:
                torch.cuda.set_device(self.device)
            for li, reqs in enumerate(self.schedule):
                with self.cv:
                    while not self.stop and li >= self.taken_layers + self.depth:
                        self.cv.wait()
                    if self.stop:
                        return
                for uri, name in reqs:                 # open the shard files here, not in the workers
                    self._file(uri, name)
                futs = [(uri, name, self.pool.submit(self._load, uri, name)) for uri, name in reqs]
                for uri, name, fut in futs:
                    item = fut.result()
                    with self.cv:
                        self.ready[(li, uri, name)] = item
                        self.cv.notify_all()
        except BaseException as exc:          # surfaced by take()
            with self.cv:
                self.error = exc
                self.cv.notify_all()

    # ---- consumer ---------------------------------------------------------------------
    def begin_layer(self, layer_index: int):
        """The merge is now at schedule entry `layer_index` (layers may be skipped on resume)."""
        with self.cv:
            self.cursor = layer_index
            if layer_index > self.taken_layers:
                self.taken_layers = layer_index
                for key in [k for k in self.ready if k[0] < layer_index]:
                    del self.ready[key]
                self.cv.notify_all()

    def take(self, uri: str, name: str) -> Optional[torch.Tensor]:
        """The tensor of the current layer, on the device; None if it was not scheduled."""
        li = self.cursor
        if li >= len(self.schedule) or (uri, name) not in self.schedule[li]:
            return None
        key = (li, uri, name)
        t_wait = time.perf_counter()
        with self.cv:
            while key not in self.ready and self.error is None:
                self.cv.wait()
            if self.error is not None:
                raise self.error
            dev, ev, host = self.ready.pop(key)
            self.remaining[li] -= 1
            if self.remaining[li] == 0 and li + 1 > self.taken_layers:
                self.taken_layers = li + 1
                self.cv.notify_all()
        if ev is not None:
            ev.synchronize()                  # copy done: the pinned buffer may go
            _, ev0, nbytes = host
            iostats.add("h2d", ev0.elapsed_time(ev) * 1e-3, nbytes)      # on the copy stream (overlaps the merge)
            del host
            # `dev` was allocated on the copy stream's pool but is consumed (merge kernels, the
            # writer's asynchronous device-to-host copy of passthrough tensors) on the caller's
            # stream: without this the block returns to the copy stream's pool the moment the
            # caller drops it and the next prefetch may overwrite it under a copy still in flight
            dev.record_stream(torch.cuda.current_stream(self.device))
        iostats.add("take_wait", time.perf_counter() - t_wait)           # the merge stalled on its inputs
        return dev

    def close(self):
        with self.cv:
            self.stop = True
            self.cv.notify_all()
        if self.thread is not None:
            self.thread.join(timeout=30)
        self.pool.shutdown(wait=True)
        self.piece_pool.shutdown(wait=True)
        for f in self.files.values():
            f.close()
        self.files.clear()
        self.ready.clear()
