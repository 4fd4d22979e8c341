"""N > 1 path on CPU: two gloo ranks run the partitioned merge (tensor partition by
LPT, ONE broadcast per base shard, per-rank part files, shard assembly) and must
produce byte-identical shards to the single-process writer path."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch
from click.testing import CliRunner
from safetensors import safe_open

from shardmerge_amd import distributed
from tests.golden import inputs as gi

REPO = Path(__file__).resolve().parents[1]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_lpt_is_balanced_and_deterministic():
    costs = [distributed.alg_bytes(n, k) for n, k in [(16, 2), (4, 2), (56, 3), (56, 2), (1, 1), (16, 4), (8, 2)]]
    owner = distributed.partition_lpt(costs, 3)
    assert owner == distributed.partition_lpt(costs, 3)
    loads = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(3)]
    assert max(loads) <= sum(costs) / 3 + max(costs)
    assert distributed.alg_bytes(10, 2) == 600 and distributed.alg_bytes(10, 3) == 1220 and distributed.alg_bytes(10, 4) == 1820   # 60n + 60n + 62n


def test_two_rank_merge_equals_single_process(tmp_path, monkeypatch):
    from tests.emul.loader import emul_engine, build
    build()
    # single process through the CLI (emulator as the device)
    one = tmp_path / "one"
    one.mkdir()
    cfg1 = gi.write_cli_model(one)
    from shardmerge_amd import engine as engine_mod
    eng = emul_engine()
    monkeypatch.setattr(engine_mod, "get_engine", lambda device=None: eng)
    from shardmerge_amd.__main__ import cli
    res = CliRunner().invoke(cli, ["merge", str(cfg1)])
    assert res.exit_code == 0, res.output

    # two ranks
    two = tmp_path / "two"
    two.mkdir()
    cfg2 = gi.write_cli_model(two)
    # a part file left by an earlier, crashed run (other partition): must not be assembled
    from safetensors.torch import save_file
    (two / "merged").mkdir()
    stale_name = sorted(gi.CLI_SHARDS)[0]
    victim = gi.CLI_SHARDS[stale_name][0][0]
    save_file({victim: torch.full((2, 2), 7.0)}, str(two / "merged" / f".part-1-{stale_name}"))
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(REPO / "tests" / "dist_worker.py"), str(cfg2)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)

    a, b = one / "merged", two / "merged"
    assert not list(b.glob(".part-*")) and not list(b.glob(".tmp-*"))      # results travel rank to rank, shards are written once
    assert sorted(p.name for p in a.iterdir()) == sorted(p.name for p in b.iterdir())
    assert (a / "README.md").read_text() == (b / "README.md").read_text()
    for shard in gi.CLI_SHARDS:
        with safe_open(str(a / shard), framework="pt") as fa, safe_open(str(b / shard), framework="pt") as fb:
            assert list(fa.keys()) == list(fb.keys())
            for k in fa.keys():
                assert torch.equal(fa.get_tensor(k), fb.get_tensor(k)), k


def test_two_rank_merge_resumes_at_shard_granularity(tmp_path):
    """A complete output shard from an earlier run is kept (its mtime does not change) and only
    the missing shards are merged, on every rank consistently."""
    from tests.emul.loader import build
    build()
    cfg = gi.write_cli_model(tmp_path)

    def run():
        port = free_port()
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, str(REPO / "tests" / "dist_worker.py"), str(cfg)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=300)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        return "\n".join(outs)

    run()
    out = tmp_path / "merged"
    shards = sorted(gi.CLI_SHARDS)
    keep, redo = shards[0], shards[-1]
    before = {s: (out / s).read_bytes() for s in shards}
    stamp = (out / keep).stat().st_mtime_ns
    (out / redo).unlink()
    log = run()
    assert "already complete" in log
    assert (out / keep).stat().st_mtime_ns == stamp
    assert {s: (out / s).read_bytes() for s in shards} == before
