"""Where the wall time of an end-to-end merge goes: cumulative seconds and bytes per stage -
read (positional reads into pinned buffers), h2d (the copy stream), take_wait (the merge stalled on its inputs),
merge (inside the HIP library's merge_layer call), d2h_wait + save (the writer thread).  Cheap enough to stay on;
tools/cli_bench.py prints the snapshot."""
from __future__ import annotations

import threading
import time
from collections import defaultdict
from typing import Dict

_lock = threading.Lock()
_seconds: Dict[str, float] = defaultdict(float)
_bytes: Dict[str, int] = defaultdict(int)
_count: Dict[str, int] = defaultdict(int)


def add(key: str, seconds: float = 0.0, nbytes: int = 0) -> None:
    with _lock:
        _seconds[key] += seconds
        _bytes[key] += nbytes
        _count[key] += 1


class timed:
    def __init__(self, key: str, nbytes: int = 0):
        self.key, self.nbytes = key, nbytes

    def __enter__(self):
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        add(self.key, time.perf_counter() - self.t0, self.nbytes)
        return False


def snapshot(reset: bool = False) -> Dict[str, dict]:
    with _lock:
        out = {k: {"seconds": round(_seconds[k], 4), "GB": round(_bytes[k] / 1e9, 3), "calls": _count[k],
                   **({"GBps": round(_bytes[k] / 1e9 / _seconds[k], 2)} if _bytes[k] and _seconds[k] > 0 else {})}
               for k in sorted(_seconds)}
        if reset:
            _seconds.clear(); _bytes.clear(); _count.clear()
    return out
