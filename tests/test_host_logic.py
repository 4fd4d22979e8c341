"""Host-side logic of the drop-in boundary (no GPU): config, writer, index ordering,
pairing, the C ABI's symbol table, and the CLI end to end - with the HIP kernels'
CPU work-group emulator standing in for the device (tests/emul)."""
import asyncio
import ctypes
import json
import re
from pathlib import Path

import click
import pytest
import torch
import yaml
from click.testing import CliRunner
from safetensors import safe_open
from safetensors.torch import save_file

from shardmerge_amd import _lib
from shardmerge_amd.config import MergeConfig, MergeModel
from shardmerge_amd.constants import INPUT_LAYER, OUTPUT_LAYER
from shardmerge_amd.index import LocalModelIndex, order_weights
from shardmerge_amd.merge.fast_fourier import FourierMerge, name_hash
from shardmerge_amd.writer import ModelWriter, ShardLayer
from tests.golden import inputs as gi

REPO = Path(__file__).resolve().parents[1]


# ---- config (reference tests/test_config.py) -------------------------------------------
def test_merge_model_layer_window():
    m = MergeModel(model="a", base="b", start_layer=2, end_layer=5)
    assert [m.use_layer_index(i) for i in (1, 2, 5, 6)] == [False, True, True, False]
    assert MergeModel(model="a", base="b").use_layer_index(10 ** 6)
    assert MergeModel(model="a", base="b").alpha == 1.0


def test_config_from_yaml_roundtrip(tmp_path):
    doc = {"output_base_model": "org/base", "output_dir": "out",
           "finetune_merge": [{"model": "org/a", "base": "org/base", "alpha": 0.8, "is_input": True},
                              {"model": "org/b", "base": "org/base", "start_layer": 2, "end_layer": -1, "is_output": True}]}
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(doc))
    cfg = MergeConfig.from_yaml(p)
    assert cfg.output_dtype == "bfloat16" and cfg.device == "cpu" and cfg.cache_dir == "cache" and cfg.storage_dir == "storage"
    assert cfg.output_astype is torch.bfloat16
    assert cfg.input_model.model == "org/a" and cfg.output_model.model == "org/b"
    assert cfg.to_dict()["finetune_merge"] == ["org/a", "org/b"]
    cfg.update({"device": "cuda", "nonsense": 1}, clean_cache=True)
    assert cfg.device == "cuda" and cfg.clean_cache and not hasattr(cfg, "nonsense")


def test_config_errors(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump({"output_base_model": "x", "finetune_merge": []}))
    with pytest.raises(click.BadParameter, match="output_dir"):
        MergeConfig.from_yaml(p)
    p.write_text(yaml.safe_dump({"output_base_model": "x", "finetune_merge": "nope", "output_dir": "o"}))
    with pytest.raises(click.BadParameter, match="must be a list"):
        MergeConfig.from_yaml(p)


# ---- writer / index ------------------------------------------------------------------------
def test_merge_options_are_additive_and_validated(tmp_path):
    doc = {"output_base_model": "o/b", "finetune_merge": [{"model": "o/f", "base": "o/b"}], "output_dir": "out"}
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(doc))
    cfg = MergeConfig.from_yaml(p)
    assert cfg.merge_options == {}
    m = FourierMerge(config=cfg, index_manager=LocalModelIndex(tmp_path), engine=object())
    assert (m.cutoff_pct, m.cull_start_pct, m.t_sum, m.target_norm_offset) == (0.08, 0.20, 1.0, 1e-10)   # reference defaults
    doc["merge_options"] = {"cutoff_pct": 0.05, "cull_start_pct": 0.1}
    p.write_text(yaml.safe_dump(doc))
    cfg = MergeConfig.from_yaml(p)
    m = FourierMerge(config=cfg, index_manager=LocalModelIndex(tmp_path), engine=object())
    assert (m.cutoff_pct, m.cull_start_pct, m.t_sum) == (0.05, 0.1, 1.0)
    for bad in ({"cutoff": 0.1}, {"cutoff_pct": 2.0}, {"t_sum": "x"}):
        doc["merge_options"] = bad
        p.write_text(yaml.safe_dump(doc))
        with pytest.raises(click.BadParameter):
            MergeConfig.from_yaml(p)


def test_shard_layer_numbers():
    mk = lambda n: ShardLayer(0, "s", n, False).layer_number
    assert mk("model.embed_tokens.weight") == INPUT_LAYER
    assert mk("model.norm.weight") == OUTPUT_LAYER and mk("lm_head.weight") == OUTPUT_LAYER
    assert mk("model.layers.17.mlp.up_proj.weight") == 17
    for bad in ("model.layers.x.w", "encoder.block.0.w", "model.layers.01.w"):
        with pytest.raises(ValueError):
            mk(bad)


def test_order_weights():
    names = ["lm_head.weight", "model.layers.1.b", "model.layers.0.b", "model.norm.weight",
             "model.layers.0.a", "model.embed_tokens.weight", "model.layers.1.a", "rotary.inv_freq"]
    assert order_weights(names) == ["model.embed_tokens.weight", "model.layers.0.a", "model.layers.0.b",
                                    "model.layers.1.a", "model.layers.1.b", "model.norm.weight", "lm_head.weight", "rotary.inv_freq"]


def test_writer_buffers_orders_and_resumes(tmp_path):
    index = {"metadata": {}, "weight_map": {"model.embed_tokens.weight": "s1", "model.layers.0.w": "s1",
                                            "model.norm.weight": "s2", "lm_head.weight": "s2"}}
    order = ["model.embed_tokens.weight", "model.layers.0.w", "model.norm.weight", "lm_head.weight"]
    w = ModelWriter(base_index=index, output_path=tmp_path / "o", layer_order=order, output_astype=torch.bfloat16)
    groups = list(w.shard_layers())
    assert [[sl.layer_name for sl in g] for g in groups] == [order[:2], order[2:]]
    w.add_tensor("model.layers.0.w", torch.ones(2, 2))
    assert not (tmp_path / "o" / "s1").exists()               # shard not complete yet: nothing written
    w.add_tensor("model.embed_tokens.weight", torch.zeros(2, 2))
    w.wait()                                                  # complete shards are written in the background
    with safe_open(str(tmp_path / "o" / "s1"), framework="pt") as f:
        assert list(f.keys()) == sorted(order[:2]) or set(f.keys()) == set(order[:2])
        assert f.get_tensor("model.layers.0.w").dtype == torch.bfloat16
    with pytest.raises(RuntimeError, match="missing 2 layers"):
        w.finalize()
    # resume: a new writer sees shard s1 as written and only needs s2
    w2 = ModelWriter(base_index=index, output_path=tmp_path / "o", layer_order=order, output_astype=torch.bfloat16)
    flags = {sl.layer_name: sl.written for g in w2.shard_layers() for sl in g}
    assert flags == {order[0]: True, order[1]: True, order[2]: False, order[3]: False}
    w2.add_tensor("model.norm.weight", torch.ones(2))
    w2.add_tensor("lm_head.weight", torch.ones(2, 2))
    w2.finalize()


def test_writer_direct_shard_write_is_safetensors_own_bytes(tmp_path):
    """The single-process writer lays a complete shard out by hand (header + tensors pwritten from their host buffers in
    32 MB pieces on a thread pool): the file must be byte for byte what safetensors.torch.save_file writes (N2;
    reference shard/writer.py:124-143 goes through save_file)."""
    from safetensors.torch import save_file
    g = torch.Generator().manual_seed(5)
    names = ["model.embed_tokens.weight", "model.layers.0.a", "model.layers.0.b", "model.layers.0.norm"]
    tensors = {names[0]: torch.randn(17, 8, generator=g), names[1]: torch.randn(300, 70, generator=g),
               names[2]: torch.randn(5, generator=g), names[3]: torch.randn(2, 3, 4, generator=g)}
    index = {"metadata": {}, "weight_map": {n: "s1" for n in names}}
    for dtype in (torch.bfloat16, torch.float32, torch.float16):
        out = tmp_path / f"o_{str(dtype).split('.')[-1]}"
        w = ModelWriter(base_index=index, output_path=out, layer_order=names, output_astype=dtype)
        for n in names:
            w.add_tensor(n, tensors[n])
        w.finalize()
        save_file({n: tensors[n].to(dtype).contiguous() for n in names}, str(tmp_path / "ref"), metadata={"format": "pt"})
        assert (out / "s1").read_bytes() == (tmp_path / "ref").read_bytes()


def test_name_hash():
    h = name_hash("model_layer_weight")                       # reference test_fast_fourier.py:62-68
    assert h.startswith("mode_laye_weig::") and re.fullmatch(r"[0-9a-f]{8}", h.split("::")[1])


@pytest.mark.parametrize("case", gi.SCHED_CASES, ids=lambda c: c["id"])
def test_correlated_pairs_matches_reference(golden, case):
    from shardmerge_amd.tensor.functions import correlated_pairs
    got = [[x, y, c] for x, y, c in correlated_pairs(gi.sched_matrix(case), case["way"])]
    assert got == golden.manifest["sched"][case["id"]]


# ---- the C ABI ------------------------------------------------------------------------------
def declared_symbols():
    text = (REPO / "include" / "shardmerge_hip.h").read_text()
    return sorted(set(re.findall(r"\b(smhip_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    """The C-ABI library loads (no GPU needed for dlopen) and exports everything the
    header declares; no compute call is made here."""
    path = _lib.hip_library_path()
    assert path.exists(), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    dll = ctypes.CDLL(str(path))
    syms = declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(dll, s), f"{s} is declared in include/shardmerge_hip.h but not exported"
    dll.smhip_version.restype = ctypes.c_char_p
    assert b"gfx950" in dll.smhip_version()
    assert dll.smhip_length_supported(8192) == 0 and dll.smhip_length_supported(17) != 0


def test_product_has_no_cpu_fallback(monkeypatch, tmp_path):
    """Without the HIP library the product path raises; without a GPU it raises too."""
    from shardmerge_amd import engine
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_HIP_LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.get_engine("cuda")
    monkeypatch.undo()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no ROCm GPU"):
            engine.get_engine("cuda")


# ---- operator + CLI with the emulator as the device --------------------------------------------
@pytest.fixture()
def emul(monkeypatch):
    from tests.emul.loader import emul_engine
    from shardmerge_amd import engine as engine_mod
    eng = emul_engine()
    monkeypatch.setattr(engine_mod, "get_engine", lambda device=None: eng)
    return eng


def test_passthrough_and_block_layers(tmp_path, emul, golden):
    cfg_path = gi.write_cli_model(tmp_path)
    cfg = MergeConfig.from_yaml(cfg_path)
    idx = LocalModelIndex(cfg.storage_path)
    merger = FourierMerge(config=cfg, index_manager=idx, engine=emul)
    assert merger.target_norm_offset == 1e-10 and merger.cull_start_pct == 0.20 and merger.task_add_models == []
    assert "SLERP-FFT" in merger.get_readme() and "org/base" in merger.get_readme()
    asyncio.run(merger.initialize())
    emb = asyncio.run(merger._merge_layer(ShardLayer(0, "s", "model.embed_tokens.weight", False), "cpu"))
    assert torch.equal(emb, gi.cli_model_tensors(1)["model.embed_tokens.weight"])      # is_input model's tensor
    head = asyncio.run(merger._merge_layer(ShardLayer(0, "s", "lm_head.weight", False), "cpu"))
    assert torch.equal(head, gi.cli_model_tensors(0)["lm_head.weight"])                # default: output_base_model
    blk = asyncio.run(merger._merge_layer(ShardLayer(0, "s", "model.layers.0.self_attn.q_proj.weight", False), "cpu"))
    assert blk.dtype == torch.bfloat16 and blk.shape == (128, 128)
    with pytest.raises(ValueError):
        asyncio.run(merger._merge_layer(ShardLayer(0, "s", "model.rotary.inv_freq", False), "cpu"))


def test_cli_end_to_end_matches_reference(tmp_path, emul, golden):
    """G8: `merge CONFIG` on a tiny on-disk model; output files, index, README and every
    tensor against what the reference CLI produced."""
    from shardmerge_amd.__main__ import cli
    cfg_path = gi.write_cli_model(tmp_path)
    res = CliRunner().invoke(cli, ["merge", str(cfg_path), "--cache-dir", str(tmp_path / "cache")])
    assert res.exit_code == 0, res.output
    out_dir = tmp_path / "merged"
    want = golden.manifest["cli"]
    assert sorted(p.name for p in out_dir.iterdir()) == want["files"]
    assert json.load(open(out_dir / "model.safetensors.index.json")) == want["index"]
    assert (out_dir / "README.md").read_text() == want["readme"]
    from oracle import spectral_oracle as so
    for shard in gi.CLI_SHARDS:
        with safe_open(str(out_dir / shard), framework="pt") as f:
            assert list(f.keys()) == sorted(n for n, _ in gi.CLI_SHARDS[shard])
            for k in f.keys():
                ref = golden.get("g8_cli.safetensors", f"{shard}::{k}")
                got = f.get_tensor(k)
                assert got.dtype == ref.dtype and got.shape == ref.shape
                if "layers" in k:
                    assert so.rel_err(got.float(), ref.float()) < 2e-3      # 128x128: threshold-tie floor
                else:
                    assert torch.equal(got, ref)


def test_cli_aborts_on_missing_model(tmp_path, emul):
    from shardmerge_amd.__main__ import cli
    cfg_path = gi.write_cli_model(tmp_path)
    doc = yaml.safe_load(cfg_path.read_text())
    doc["finetune_merge"][0]["model"] = "org/absent"
    cfg_path.write_text(yaml.safe_dump(doc))
    res = CliRunner().invoke(cli, ["merge", str(cfg_path)])
    assert res.exit_code != 0


def test_shard_alias_package():
    import importlib
    assert importlib.import_module("shard.__main__").cli.name == "cli"


def test_unsupported_lengths_fail_before_any_layer_is_merged(tmp_path, emul):
    """A block tensor with a transform length the library cannot do (here 2 * 16411, above the
    longest transform) stops the run in initialize(), before any shard has been written."""
    from shardmerge_amd.__main__ import cli
    cfg_path = gi.write_cli_model(tmp_path)
    cfg = MergeConfig.from_yaml(cfg_path)
    victim = "model.layers.1.self_attn.q_proj.weight"
    for uri in ["org/base", "org/ft1", "org/ft2"]:
        d = cfg.storage_path / uri
        shard = json.load(open(d / "model.safetensors.index.json"))["weight_map"][victim]
        with safe_open(str(d / shard), framework="pt") as f:
            tens = {k: f.get_tensor(k) for k in f.keys()}
        tens[victim] = torch.zeros(2, 40009, dtype=torch.bfloat16)
        save_file(tens, str(d / shard), metadata={"format": "pt"})
    if emul.lib.length_supported(40009):
        pytest.skip("every length is supported by this build")
    merger = FourierMerge(config=cfg, index_manager=LocalModelIndex(cfg.storage_path), engine=emul)
    with pytest.raises(NotImplementedError, match="model.layers.1.self_attn.q_proj.weight"):
        asyncio.run(merger.initialize())
    res = CliRunner().invoke(cli, ["merge", str(cfg_path)])
    assert res.exit_code != 0
    assert not list((tmp_path / "merged").glob("*.safetensors"))          # nothing was written


def test_cli_merges_block_tensors_whose_lengths_have_no_plan(tmp_path, emul):
    """The pre-flight accepts, and `merge CONFIG` merges, a [86 x 64] tensor (86 = 43 * 2: split column
    length) and a [64 x 86] one (rough row length: merged transposed) - the shapes of a Llama-2-7B MLP in
    miniature - and the results are the oracle's."""
    from shardmerge_amd.__main__ import cli
    from oracle import spectral_oracle as so
    cfg_path = gi.write_cli_model(tmp_path)
    cfg = MergeConfig.from_yaml(cfg_path)
    victims = {"model.layers.0.self_attn.q_proj.weight": (86, 64), "model.layers.1.self_attn.q_proj.weight": (64, 86)}
    g = torch.Generator().manual_seed(11)
    made = {}
    for name, shape in victims.items():
        base = (torch.randn(shape, generator=g) * 0.02).to(torch.bfloat16)
        made[name] = [base] + [(base.float() + torch.randn(shape, generator=g) * s_).to(torch.bfloat16) for s_ in (0.002, 0.003)]
    for which, uri in enumerate(["org/base", "org/ft1", "org/ft2"]):
        d = cfg.storage_path / uri
        wm = json.load(open(d / "model.safetensors.index.json"))["weight_map"]
        for shard in sorted({wm[v] for v in victims}):
            with safe_open(str(d / shard), framework="pt") as f:
                tens = {k: f.get_tensor(k) for k in f.keys()}
            for v in victims:
                if wm[v] == shard:
                    tens[v] = made[v][which]
            save_file(tens, str(d / shard), metadata={"format": "pt"})
    res = CliRunner().invoke(cli, ["merge", str(cfg_path), "--cache-dir", str(tmp_path / "cache")])
    assert res.exit_code == 0, res.output
    out_dir = tmp_path / "merged"
    wm = json.load(open(out_dir / "model.safetensors.index.json"))["weight_map"]
    alphas = [m.alpha for m in cfg.finetune_merge]
    for name, shape in victims.items():
        with safe_open(str(out_dir / wm[name]), framework="pt") as f:
            got = f.get_tensor(name)
        base, ft1, ft2 = made[name]
        errs = []
        for flip in ((False, True) if shape[1] == 86 else (False,)):      # transposed merge: either orientation
            tt = (lambda x: x.T.contiguous()) if flip else (lambda x: x)
            ref = so.merge_layer([tt(ft1), tt(ft2)], [tt(base)] * 2, alphas, tt(base))
            errs.append(so.rel_err(tt(got).float(), ref.float()))
        assert got.shape == shape and got.dtype == torch.bfloat16 and min(errs) < 2e-3, (name, errs)


# ---- N3 / N4: operator choice and b through the YAML ------------------------------------------------
def test_merge_options_operator_and_b(tmp_path):
    doc = {"output_base_model": "o/b", "finetune_merge": [{"model": "o/f", "base": "o/b"}], "output_dir": "out"}
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(doc))
    assert MergeConfig.from_yaml(p).operator == "fourier"
    doc["merge_options"] = {"operator": "task_addition", "b": 0.3}
    p.write_text(yaml.safe_dump(doc))
    cfg = MergeConfig.from_yaml(p)
    assert cfg.operator == "task_addition" and cfg.merge_options == {"b": 0.3}
    from shardmerge_amd.merge import operator_class
    from shardmerge_amd.merge.addition import AdditionMerge
    from shardmerge_amd.merge.taskaddition import TaskAdditionMerge
    assert operator_class("task_addition") is TaskAdditionMerge and operator_class("addition") is AdditionMerge
    assert operator_class("fourier") is FourierMerge
    doc["merge_options"] = {"operator": "median"}
    p.write_text(yaml.safe_dump(doc))
    with pytest.raises(click.BadParameter, match="operator"):
        MergeConfig.from_yaml(p)


@pytest.mark.parametrize("op", ["addition", "task_addition"])
def test_cli_with_the_addition_operators(tmp_path, emul, op):
    """`merge CONFIG` with merge_options.operator: every tensor (embeddings and head too, as in the
    reference: these operators have no passthrough) becomes the (sign-agreeing) sum of deltas."""
    from oracle import spectral_oracle as so
    from shardmerge_amd.__main__ import cli
    cfg_path = gi.write_cli_model(tmp_path)
    doc = yaml.safe_load(cfg_path.read_text())
    doc["merge_options"] = {"operator": op}
    cfg_path.write_text(yaml.safe_dump(doc))
    res = CliRunner().invoke(cli, ["merge", str(cfg_path)])
    assert res.exit_code == 0, res.output
    out_dir = tmp_path / "merged"
    readme = (out_dir / "README.md").read_text()
    assert "Merged Model" in readme and "org/base" in readme and ("sign agreement" in readme) == (op == "task_addition")
    models = [gi.cli_model_tensors(i) for i in range(3)]
    fn = so.task_addition_merge if op == "task_addition" else so.addition_merge
    n = 0
    for shard in gi.CLI_SHARDS:
        with safe_open(str(out_dir / shard), framework="pt") as f:
            for k in f.keys():
                assert torch.equal(f.get_tensor(k), fn([models[1][k], models[2][k]], models[0][k]).to(torch.bfloat16)), k
                n += 1
    assert n == sum(len(v) for v in gi.CLI_SHARDS.values())


def test_legacy_operator_is_selectable_from_yaml(tmp_path):
    """merge_options.operator: fourier_legacy (+ task_add_models) -> LegacyFourierMerge (reference shard/merge/fourier.py)"""
    import yaml
    from shardmerge_amd.config import MergeConfig
    from shardmerge_amd.merge import operator_class
    from shardmerge_amd.merge.fourier_legacy import LegacyFourierMerge
    doc = {"output_base_model": "org/base", "output_dir": str(tmp_path / "o"),
           "finetune_merge": [{"model": "org/a", "base": "org/base", "alpha": 0.5, "is_input": True, "is_output": True},
                              {"model": "org/b", "base": "org/base", "alpha": 0.5}],
           "merge_options": {"operator": "fourier_legacy", "task_add_models": ["org/b"]}}
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(doc))
    cfg = MergeConfig.from_yaml(p)
    assert cfg.operator == "fourier_legacy" and cfg.task_add_models == ["org/b"]
    op = operator_class(cfg.operator)(config=cfg, index_manager=None)
    assert isinstance(op, LegacyFourierMerge) and op.task_add_models == ["org/b"]
    assert op.get_readme() == "# SLERP-FFT Merged Model\nBase: org/base\nModels merged:\n- org/a\n- org/b\n"
    doc["merge_options"]["task_add_models"] = "org/b"
    p.write_text(yaml.safe_dump(doc))
    with pytest.raises(Exception):
        MergeConfig.from_yaml(p)
