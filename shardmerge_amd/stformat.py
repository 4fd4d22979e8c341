"""safetensors on-disk layout, written by hand where a file is filled in place: the pre-sized output shards of the
multi-GPU merge (distributed.py) and the single-process writer's direct shard write (writer.py)."""
from __future__ import annotations

import json
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Sequence, Tuple

import torch

ST_DTYPES = {"BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32, "F64": torch.float64}
ST_NAMES = {torch.bfloat16: "BF16", torch.float16: "F16", torch.float32: "F32", torch.float64: "F64"}
ST_SIZE = {"BF16": 2, "F16": 2, "F32": 4, "F64": 8}
_ST_SIZE = ST_SIZE


def shard_header(entries: Sequence[Tuple[str, str, Sequence[int]]], metadata: Dict[str, str]) -> Tuple[bytes, Dict[str, Tuple[int, int]]]:
    """The bytes in front of a safetensors payload (8-byte length + JSON padded to 8 bytes) for tensors
    (name, dtype string, shape), laid out as safetensors' own writer does it (by dtype, widest first, then by
    name - tests/test_host_logic.py pins the bytes against safetensors.torch.save_file), and the payload
    offsets (begin, end) per name."""
    rank_of_dtype = {"F64": 0, "F32": 1, "BF16": 2, "F16": 2}
    order = sorted(entries, key=lambda e: (rank_of_dtype.get(e[1], 9), e[0]))
    doc: Dict[str, object] = {"__metadata__": dict(metadata)}
    offsets, pos = {}, 0
    for name, dt, shape in order:
        n = _ST_SIZE[dt]
        for d in shape:
            n *= int(d)
        doc[name] = {"dtype": dt, "shape": [int(d) for d in shape], "data_offsets": [pos, pos + n]}
        offsets[name] = (pos, pos + n)
        pos += n
    blob = json.dumps(doc, separators=(",", ":")).encode()
    blob += b" " * ((8 - len(blob) % 8) % 8)
    return len(blob).to_bytes(8, "little") + blob, offsets



_CHUNK = 32 << 20


def pwrite_tensors(fd: int, data_start: int, offsets: Dict[str, Tuple[int, int]], tensors: Dict[str, torch.Tensor],
                   pool: ThreadPoolExecutor) -> int:
    """contiguous CPU tensors -> their payload offsets of an open shard file, in 32 MB pieces on `pool`
    (os.pwrite releases the GIL: the copies into the page cache run side by side).  Returns the bytes written."""
    jobs, total = [], 0
    for name, t in tensors.items():
        view = memoryview(t.reshape(-1).view(torch.uint8).numpy()).cast("B")
        begin, end = offsets[name]
        assert end - begin == len(view), (name, end - begin, len(view))
        total += len(view)
        for off in range(0, len(view), _CHUNK):
            jobs.append(pool.submit(_pwrite_all, fd, view[off:off + _CHUNK], data_start + begin + off))
    for j in jobs:
        j.result()
    return total


def mmap_write_tensors(fd: int, data_start: int, total: int, offsets: Dict[str, Tuple[int, int]],
                       tensors: Dict[str, torch.Tensor], pool: ThreadPoolExecutor) -> int:
    """The same through a shared mapping of the (pre-sized) file: writes to ONE file serialise on its inode lock
    (measured on tmpfs: 8 threads of pwrite 4.8 GB/s), page faults of a mapping do not - numpy's copy releases the GIL."""
    import mmap

    import numpy as np
    mm = mmap.mmap(fd, data_start + total, access=mmap.ACCESS_WRITE)
    try:
        dst = np.frombuffer(mm, dtype=np.uint8)
        jobs, done = [], 0
        for name, t in tensors.items():
            src = t.reshape(-1).view(torch.uint8).numpy()
            begin, end = offsets[name]
            assert end - begin == src.size, (name, end - begin, src.size)
            done += src.size
            for off in range(0, src.size, _CHUNK):
                n = min(_CHUNK, src.size - off)
                jobs.append(pool.submit(np.copyto, dst[data_start + begin + off:data_start + begin + off + n], src[off:off + n]))
        for j in jobs:
            j.result()
        del dst
    finally:
        mm.close()
    return done


def _pwrite_all(fd: int, view, pos: int):
    done = 0
    while done < len(view):
        done += os.pwrite(fd, view[done:], pos + done)
