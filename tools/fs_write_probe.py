#!/usr/bin/env python3
"""How fast does this box take file writes?  N threads, each pwrite()-ing 1 GiB in 32 MB pieces into its OWN file under
--root (default /dev/shm): fresh files (new pages) and a second pass over the same files (pages exist).  The end-to-end
merge (tools/cli_bench.py) cannot write its output faster than this."""
import argparse, os, time, threading
import numpy as np
ap = argparse.ArgumentParser(); ap.add_argument("--root", default="/dev/shm"); args = ap.parse_args()
buf = np.random.randint(0, 255, 32 << 20, dtype=np.uint8)
view = memoryview(buf)
def work(path, n):
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o644)
    for i in range(n):
        os.pwrite(fd, view, i * len(view))
    os.close(fd)
for nthreads in (1, 2, 4, 8, 12):
    paths = [f"{args.root}/_probe_{i}" for i in range(nthreads)]
    for p in paths:
        if os.path.exists(p): os.unlink(p)
    for label in ("fresh files", "rewrite"):
        ths = [threading.Thread(target=work, args=(p, 32)) for p in paths]
        t0 = time.time(); [t.start() for t in ths]; [t.join() for t in ths]; dt = time.time() - t0
        print(f"{nthreads:2d} thread(s), {label:11s}: {nthreads * 32 * len(view) / dt / 1e9:6.2f} GB/s")
    for p in paths: os.unlink(p)
# one file, 8 threads writing disjoint pieces (what a single shard's write looks like)
p = f"{args.root}/_probe_one"
fd = os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644); os.ftruncate(fd, 8 * 16 * len(view))
def piece(k):
    for i in range(16): os.pwrite(fd, view, (k * 16 + i) * len(view))
ths = [threading.Thread(target=piece, args=(k,)) for k in range(8)]
t0 = time.time(); [t.start() for t in ths]; [t.join() for t in ths]; dt = time.time() - t0
print(f" 8 threads into ONE fresh file: {8 * 16 * len(view) / dt / 1e9:6.2f} GB/s")
os.close(fd); os.unlink(p)
