"""Build (g++) and bind the CPU work-group emulator of the HIP kernels.
TEST INFRASTRUCTURE ONLY - see smhip_emul.cpp."""
import subprocess
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
SO = HERE / "libshardmerge_emul.so"
CSRC = HERE.parents[1] / "shardmerge_amd" / "csrc"


def build(force: bool = False, flags=(), tag: str = "") -> Path:
    """flags / tag: a second build of the same sources with -D switches (an engine mode that is off by default), kept
    beside the default one as libshardmerge_emul_<tag>.so"""
    SO = HERE / (f"libshardmerge_emul_{tag}.so" if tag else "libshardmerge_emul.so")
    srcs = [HERE / "smhip_emul.cpp"] + sorted(CSRC.glob("*.hpp")) + sorted(CSRC.glob("*.inc")) + \
           [HERE.parents[1] / "include" / "shardmerge_hip.h"]
    newest = max(p.stat().st_mtime for p in srcs)
    stale = lambda: force or not SO.exists() or SO.stat().st_mtime < newest
    if stale():
        # several pytest-xdist workers may get here together: one builds (into a temporary name, renamed when
        # complete), the others wait for the lock and find the library fresh
        import fcntl
        import os
        with open(HERE / ".build.lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                tmp = SO.with_suffix(f".so.{os.getpid()}.tmp")
                subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unknown-pragmas",
                                "-Wno-unused-variable", *flags, str(HERE / "smhip_emul.cpp"), "-o", str(tmp)], check=True)
                os.replace(tmp, SO)
    return SO


_engine = None


def emul_engine_variant(flags, tag):
    """a fresh Engine on an alternate build (see build())"""
    from shardmerge_amd._lib import SmhipLibrary
    from shardmerge_amd.engine import Engine
    return Engine(lib=SmhipLibrary(build(flags=tuple(flags), tag=tag)), device=torch.device("cpu"))


def emul_engine():
    """Engine bound to the emulator; tensors live on the CPU."""
    global _engine
    if _engine is None:
        from shardmerge_amd._lib import SmhipLibrary
        from shardmerge_amd.engine import Engine
        _engine = Engine(lib=SmhipLibrary(build()), device=torch.device("cpu"))
    return _engine
