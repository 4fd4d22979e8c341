"""Prefetching loader: byte-exact reads, schedule order, bounded depth,
and the CLI result does not depend on whether prefetching is on."""
import threading

import pytest
import torch
from click.testing import CliRunner
from safetensors import safe_open
from safetensors.torch import save_file

from shardmerge_amd.index import LocalModelIndex
from shardmerge_amd.loader import PrefetchLoader, ShardFile
from tests.golden import inputs as gi


def _model(tmp_path, uri="org/m", n=6):
    d = tmp_path / uri
    d.mkdir(parents=True)
    g = torch.Generator().manual_seed(3)
    tensors = {
        "a.bf16": torch.randn(33, 17, generator=g).to(torch.bfloat16),
        "b.f16": torch.randn(5, generator=g).to(torch.float16),
        "c.f32": torch.randn(7, 3, 2, generator=g),
        "d.empty": torch.empty(0, 4),
        "e.i64": torch.arange(11),
    }
    for i in range(n):
        tensors[f"model.layers.{i}.w"] = torch.randn(16, 8, generator=g).to(torch.bfloat16)
    names = list(tensors)
    half = len(names) // 2
    shards = {"s1.safetensors": names[:half], "s2.safetensors": names[half:]}
    weight_map = {}
    for s, ns in shards.items():
        save_file({k: tensors[k] for k in ns}, str(d / s), metadata={"format": "pt"})
        weight_map.update({k: s for k in ns})
    import json
    json.dump({"metadata": {}, "weight_map": weight_map}, open(d / "model.safetensors.index.json", "w"))
    return uri, tensors


def test_shard_file_reads_are_byte_exact(tmp_path):
    uri, tensors = _model(tmp_path)
    for shard in ("s1.safetensors", "s2.safetensors"):
        f = ShardFile(tmp_path / uri / shard)
        with safe_open(str(tmp_path / uri / shard), framework="pt") as ref:
            for k in ref.keys():
                shape, dtype, nbytes = f.meta(k)
                buf = torch.empty(nbytes, dtype=torch.uint8)
                if nbytes:
                    f.read_into(k, buf)
                got = buf.view(dtype).reshape(shape) if nbytes else torch.empty(shape, dtype=dtype)
                want = ref.get_tensor(k)
                assert got.dtype == want.dtype and got.shape == want.shape
                assert torch.equal(got.view(torch.uint8) if nbytes else got, want.view(torch.uint8) if nbytes else want)
        f.close()


def test_prefetch_hands_over_the_schedule_and_stays_within_depth(tmp_path):
    import asyncio
    uri, tensors = _model(tmp_path, n=8)
    index = LocalModelIndex(tmp_path)
    asyncio.run(index.add_model(uri))
    schedule = [[(uri, f"model.layers.{i}.w"), (uri, "c.f32")] for i in range(8)]
    loader = PrefetchLoader(index, "cpu", depth=2)
    loader.start(schedule)
    try:
        for i in range(8):
            loader.begin_layer(i)
            # never more than `depth` layers ahead of the consumer
            with loader.cv:
                ahead = max((k[0] for k in loader.ready), default=i)
            assert ahead < i + 2
            assert loader.take(uri, "not.scheduled") is None
            w = loader.take(uri, f"model.layers.{i}.w")
            c = loader.take(uri, "c.f32")
            assert torch.equal(w, tensors[f"model.layers.{i}.w"]) and torch.equal(c, tensors["c.f32"])
        assert not loader.ready                       # nothing is cached
    finally:
        loader.close()
    assert not any(t.name == "shardmerge-prefetch" and t.is_alive() for t in threading.enumerate())


def test_prefetch_error_reaches_the_consumer(tmp_path):
    import asyncio
    uri, _ = _model(tmp_path)
    index = LocalModelIndex(tmp_path)
    asyncio.run(index.add_model(uri))
    loader = PrefetchLoader(index, "cpu")
    loader.start([[(uri, "no.such.tensor")]])
    try:
        loader.begin_layer(0)
        with pytest.raises(KeyError):
            loader.take(uri, "no.such.tensor")
    finally:
        loader.close()


def test_cli_output_is_the_same_with_and_without_prefetch(tmp_path, monkeypatch):
    from tests.emul.loader import emul_engine
    import shardmerge_amd.engine as eng_mod
    from shardmerge_amd.__main__ import cli
    eng = emul_engine()
    monkeypatch.setattr(eng_mod, "get_engine", lambda device=None: eng)
    outs = []
    for prefetch in ("1", "0"):
        root = tmp_path / f"p{prefetch}"
        root.mkdir()
        monkeypatch.setenv("SHARDMERGE_PREFETCH", prefetch)
        cfg_path = gi.write_cli_model(root)
        res = CliRunner().invoke(cli, ["merge", str(cfg_path), "--cache-dir", str(root / "cache")])
        assert res.exit_code == 0, res.output
        out = {}
        for shard in gi.CLI_SHARDS:
            with safe_open(str(root / "merged" / shard), framework="pt") as f:
                for k in f.keys():
                    out[k] = f.get_tensor(k)
        outs.append(out)
    assert outs[0].keys() == outs[1].keys()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_resume_recomputes_only_the_missing_shard(tmp_path, monkeypatch):
    """A second run over a partly written output directory merges (and prefetches) only the
    tensors of the shards that are missing (reference writer.py:93-113 resume semantics)."""
    from tests.emul.loader import emul_engine
    import shardmerge_amd.engine as eng_mod
    import shardmerge_amd.loader as loader_mod
    from shardmerge_amd.__main__ import cli
    eng = emul_engine()
    monkeypatch.setattr(eng_mod, "get_engine", lambda device=None: eng)
    cfg_path = gi.write_cli_model(tmp_path)
    res = CliRunner().invoke(cli, ["merge", str(cfg_path), "--cache-dir", str(tmp_path / "cache")])
    assert res.exit_code == 0, res.output
    out_dir = tmp_path / "merged"
    shards = sorted(gi.CLI_SHARDS)
    first = {}
    for shard in shards:
        with safe_open(str(out_dir / shard), framework="pt") as f:
            first[shard] = {k: f.get_tensor(k) for k in f.keys()}
    victim = shards[-1]
    (out_dir / victim).unlink()
    seen = []
    real_start = loader_mod.PrefetchLoader.start

    def spy(self, schedule):
        seen.extend(name for reqs in schedule for (_, name) in reqs)
        return real_start(self, schedule)

    monkeypatch.setattr(loader_mod.PrefetchLoader, "start", spy)
    res = CliRunner().invoke(cli, ["merge", str(cfg_path), "--cache-dir", str(tmp_path / "cache")])
    assert res.exit_code == 0, res.output
    assert seen and set(seen) <= {n for n, _ in gi.CLI_SHARDS[victim]}       # nothing of the finished shards is read again
    with safe_open(str(out_dir / victim), framework="pt") as f:
        for k in f.keys():
            assert torch.equal(f.get_tensor(k), first[victim][k])
