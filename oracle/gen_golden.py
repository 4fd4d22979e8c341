#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Run in the build container only (the reference lives at /root/reference and
never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

It imports the reference's own ``shard`` package read-only, feeds it seeded
inputs (regenerable from ``tests/golden/inputs.py``) and stores the outputs as
small safetensors files plus one JSON manifest.  Fixtures are data only: no
reference source text is stored.
"""
import asyncio
import json
import os
import sys
import tempfile
from pathlib import Path
from unittest.mock import AsyncMock, patch

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("golden_inputs", REPO / "tests" / "golden" / "inputs.py")
gi = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gi)

import shard.tensor.functions as ref_fn  # noqa: E402  (the reference)
from shard.config import MergeConfig, MergeModel  # noqa: E402
from shard.download import DownloadManager  # noqa: E402
from shard.index import HFMultiModelIndex  # noqa: E402
from shard.merge.fast_fourier import FourierMerge, TensorDiskCache  # noqa: E402
from shard.writer import ShardLayer  # noqa: E402

OUT = REPO / "tests" / "golden"
torch.set_num_threads(8)


def cplx(store, key, z):
    store[key + ".re"] = z.real.contiguous().clone()
    store[key + ".im"] = z.imag.contiguous().clone()


def gen_fft(manifest):
    store = {}
    for case in gi.FFT_CASES:
        x = gi.fft_input(case)
        f = ref_fn.fft_transform(x, "cpu")
        cplx(store, f"{case['id']}.fft", f)
        store[f"{case['id']}.ifft"] = ref_fn.ifft_transform(f, "cpu").contiguous().clone()
        manifest["inputs"][case["id"]] = gi.checksum(x)
    save_file(store, str(OUT / "g1_fft.safetensors"))


def gen_interp(manifest):
    store = {}
    for case in gi.INTERP_CASES:
        a, b = gi.pair_input(case)
        fa = ref_fn.fft_transform(a, "cpu")
        fb = ref_fn.fft_transform(b, "cpu")
        r = ref_fn.interpolate_fft_components(
            fa, fb, t=case["t"], device="cpu", t_sum=case["t_sum"],
            cutoff_pct=case["cutoff"], cull_pct=case["cull"], interp_imag=case["imag"])
        cplx(store, case["id"], r)
        manifest["inputs"][case["id"]] = gi.checksum(a) + gi.checksum(b)
    save_file(store, str(OUT / "g2_interp.safetensors"))


def gen_slerp(manifest):
    store = {}
    for case in gi.SLERP_CASES:
        a, b = gi.pair_input(case)
        store[case["id"]] = ref_fn.slerp(a, b, case["t"]).contiguous().clone()
        manifest["inputs"][case["id"]] = gi.checksum(a) + gi.checksum(b)
    save_file(store, str(OUT / "g3_slerp.safetensors"))


def gen_pair(manifest):
    store = {}
    meta = {}
    for case in gi.PAIR_CASES:
        a, b = gi.pair_input(case)
        m, n0, n1 = ref_fn.merge_tensors_fft2_slerp(
            a, b, t=case["t"], device="cpu", b=case["b"], t_sum=case["t_sum"],
            cutoff_pct=case["cutoff"], cull_pct=case["cull"])
        store[case["id"]] = m.contiguous().clone()
        meta[case["id"]] = {"n0": n0, "n1": n1}
        manifest["inputs"][case["id"]] = gi.checksum(a) + gi.checksum(b)
    for case in gi.ARITH_CASES:
        a, b = gi.pair_input(case)
        m = ref_fn.task_arithmetic_fft2(a, b, t=case["t"], device="cpu", agreement=case["agreement"])
        store[case["id"]] = m.contiguous().clone()
        manifest["inputs"][case["id"]] = gi.checksum(a) + gi.checksum(b)
    save_file(store, str(OUT / "g4_pair.safetensors"))
    manifest["pair_meta"] = meta


def gen_pairs_sched(manifest):
    out = {}
    for case in gi.SCHED_CASES:
        corr = gi.sched_matrix(case)
        out[case["id"]] = [[int(x), int(y), float(c)] for x, y, c in ref_fn.correlated_pairs(corr, case["way"])]
    manifest["sched"] = out


def run_ref_layer(case, tmp):
    """Drive the reference FourierMerge._merge_layer with faked tensor loads
    (same pattern as the reference's tests/merge/test_fast_fourier.py:299-316)."""
    tensors, models, cfg_kw, layer_name = gi.layer_inputs(case)
    cfg = MergeConfig(
        finetune_merge=[MergeModel(**m) for m in models],
        output_base_model=cfg_kw["output_base_model"],
        output_dir=str(Path(tmp) / "out"), device="cpu",
        cache_dir=str(Path(tmp) / f"cache_{case['id']}"), storage_dir=str(Path(tmp) / "storage"))
    idx = HFMultiModelIndex(download_manager=DownloadManager(storage_path=Path(tmp) / "storage"),
                            cache_path=Path(tmp) / "cache_idx")
    merger = FourierMerge(config=cfg, index_manager=idx)

    def fake_get(model_uri, tensor_name, device="cpu"):
        p = AsyncMock()
        p.get = AsyncMock(return_value=tensors[model_uri])
        return p

    sl = ShardLayer(layer_order_idx=1, shard_name="model-00001.safetensors", layer_name=layer_name, written=False)

    async def go():
        with patch.object(idx, "get_tensor", side_effect=fake_get):
            with patch.object(idx, "preload_tensor", new_callable=AsyncMock):
                return await merger._merge_layer(sl, device="cpu")

    return asyncio.run(go())


class fp64_fft:
    """Context: the reference's torch.fft calls evaluated in float64 and rounded back.
    Used to measure how much the REFERENCE's own output moves under a (more
    accurate) FFT: its reproducibility floor (see oracle/chaos_probe.py)."""
    NAMES = ("fft", "fftn", "ifft", "ifftn")

    def __enter__(self):
        self.orig = {n: getattr(torch.fft, n) for n in self.NAMES}

        def wrap(f):
            def g(x, *a, **k):
                x = x.to(torch.complex128) if x.is_complex() else x.to(torch.float64)
                return f(x, *a, **k).to(torch.complex64)
            return g
        for n in self.NAMES:
            setattr(torch.fft, n, wrap(self.orig[n]))

    def __exit__(self, *exc):
        for n in self.NAMES:
            setattr(torch.fft, n, self.orig[n])


def gen_layers(manifest):
    store = {}
    floors = {}
    with tempfile.TemporaryDirectory() as tmp:
        for case in gi.LAYER_CASES:
            out = run_ref_layer(case, tmp)
            store[case["id"]] = out.contiguous().clone()
            tensors, _, _, _ = gi.layer_inputs(case)
            manifest["inputs"][case["id"]] = sum((gi.checksum(v) for v in tensors.values()), [])
            with fp64_fft():
                out64 = run_ref_layer(case, tmp)
            d = (out.double() - out64.double()).norm() / out.double().norm()
            floors[case["id"]] = float(d)
    save_file(store, str(OUT / "g7_layer.safetensors"))
    manifest["layer_self_floor"] = floors


def run_ref_layer_with_delta(case, tmp):
    """run_ref_layer plus the fp32 merged delta: the last tensor the reference's TensorDiskCache hands back
    (fast_fourier.py:256-257, `result_tensor`) before the add-back and the bf16 cast."""
    seen = []
    orig_get = TensorDiskCache.get

    def spy(self, *a, **k):
        t = orig_get(self, *a, **k)
        if t is not None:
            seen.append(t)
        return t
    with patch.object(TensorDiskCache, "get", spy):
        out = run_ref_layer(case, tmp)
    return out, seen[-1].float().clone()


def gen_floor(manifest):
    """G11: K = 3 / K = 4 layers at 1024 x 1024: the reference's output and merged delta, as it is and with its FFTs
    evaluated in float64 (fp64_fft) - the distance between the two is the reference's own reproducibility floor."""
    store, meta = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for case in gi.FLOOR_CASES:
            out, delta = run_ref_layer_with_delta(case, tmp)
            with fp64_fft():
                out64, delta64 = run_ref_layer_with_delta(case, tmp)
            cid = case["id"]
            store[cid + "/out"] = out.contiguous().clone()
            store[cid + "/out_fp64"] = out64.contiguous().clone()
            store[cid + "/delta_f16"] = (delta * gi.FLOOR_DELTA_SCALE).to(torch.float16).contiguous()
            store[cid + "/delta_fp64_f16"] = (delta64 * gi.FLOOR_DELTA_SCALE).to(torch.float16).contiguous()
            tensors, _, _, _ = gi.layer_inputs(case)
            manifest["inputs"][cid] = sum((gi.checksum(v) for v in tensors.values()), [])
            rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
            meta[cid] = {"out_floor": rel(out64.float(), out.float()), "delta_floor": rel(delta64, delta),
                         "delta_norm": float(delta.double().norm())}
            print(cid, meta[cid])
    save_file(store, str(OUT / "g11_floor.safetensors"))
    manifest["floor_meta"] = meta


def run_ref_legacy_layer(case, tmp):
    """The reference's LEGACY operator (shard/merge/fourier.py:35-205) on a layer, tensor loads faked as for the fast
    one.  It subtracts the base from the finetunes IN PLACE (fourier.py:113): every load hands out a clone."""
    from shard.merge.fourier import FourierMerge as LegacyFourierMerge
    tensors, models, cfg_kw, layer_name = gi.layer_inputs(case)
    cfg = MergeConfig(finetune_merge=[MergeModel(**m) for m in models], output_base_model=cfg_kw["output_base_model"],
                      output_dir=str(Path(tmp) / "out"), device="cpu", cache_dir=str(Path(tmp) / f"cache_{case['id']}"),
                      storage_dir=str(Path(tmp) / "storage"))
    idx = HFMultiModelIndex(download_manager=DownloadManager(storage_path=Path(tmp) / "storage"), cache_path=Path(tmp) / "cache_idx")
    merger = LegacyFourierMerge(config=cfg, task_add_models=case.get("task_add"), index_manager=idx)

    def fake_get(model_uri, tensor_name, device="cpu"):
        p = AsyncMock()
        p.get = AsyncMock(return_value=tensors[model_uri].clone())
        return p
    sl = ShardLayer(layer_order_idx=1, shard_name="model-00001.safetensors", layer_name=layer_name, written=False)

    async def go():
        with patch.object(idx, "get_tensor", side_effect=fake_get):
            return await merger._merge_layer(sl, device="cpu")
    return asyncio.run(go())


def gen_legacy(manifest):
    """G12: outputs of the legacy operator, and its self-distance under fp64 FFTs (the K >= 3 floor, as for G7)."""
    store, floors = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for case in gi.LEGACY_CASES:
            out = run_ref_legacy_layer(case, tmp)
            with fp64_fft():
                out64 = run_ref_legacy_layer(case, tmp)
            store[case["id"]] = out.float().contiguous().clone()
            floors[case["id"]] = float((out.double() - out64.double()).norm() / out.double().norm())
            manifest["inputs"][case["id"]] = sum((gi.checksum(v) for v in gi.layer_inputs(case)[0].values()), [])
            print(case["id"], out.dtype, floors[case["id"]])
    save_file(store, str(OUT / "g12_legacy.safetensors"))
    manifest["legacy_self_floor"] = floors


def gen_cli(manifest):
    """G8: the reference CLI end to end on a tiny local model."""
    from click.testing import CliRunner
    from safetensors import safe_open
    from shard.__main__ import cli
    store = {}
    with tempfile.TemporaryDirectory() as tmp:
        cfg_path = gi.write_cli_model(Path(tmp))
        res = CliRunner().invoke(cli, ["merge", str(cfg_path), "--cache-dir", str(Path(tmp) / "cache")])
        assert res.exit_code == 0, res.output
        out_dir = Path(tmp) / "merged"
        for shard_file in sorted(out_dir.glob("*.safetensors")):
            with safe_open(str(shard_file), framework="pt") as f:
                for k in f.keys():
                    store[f"{shard_file.name}::{k}"] = f.get_tensor(k).contiguous().clone()
        manifest["cli"] = {
            "files": sorted(p.name for p in out_dir.iterdir()),
            "index": json.load(open(out_dir / "model.safetensors.index.json")),
            "readme": (out_dir / "README.md").read_text(),
        }
    save_file(store, str(OUT / "g8_cli.safetensors"))


def gen_addition(manifest):
    """G9: the reference's AdditionMerge / TaskAdditionMerge on seeded tensors (their own tests pin
    only all-ones cases: tests/merge/test_addition.py:92,139,186,233, test_taskaddition.py:45-93)."""
    from shard.merge.addition import AdditionMerge
    from shard.merge.taskaddition import TaskAdditionMerge
    store = {}
    with tempfile.TemporaryDirectory() as tmp:
        for case in gi.ADDITION_CASES:
            base, fts = gi.addition_inputs(case)
            tensors = {"org/base": base}
            for i, t in enumerate(fts):
                tensors[f"org/ft{i}"] = t
            cfg = MergeConfig(
                finetune_merge=[MergeModel(model=f"org/ft{i}", base="org/base") for i in range(len(fts))],
                output_base_model="org/base", output_dir=str(Path(tmp) / "out"), device="cpu",
                cache_dir=str(Path(tmp) / "cache"), storage_dir=str(Path(tmp) / "storage"))
            idx = HFMultiModelIndex(download_manager=DownloadManager(storage_path=Path(tmp) / "storage"),
                                    cache_path=Path(tmp) / "cache_idx")

            def fake_get(model_uri, tensor_name, device="cpu"):
                p = AsyncMock()
                p.get = AsyncMock(return_value=tensors[model_uri].clone())
                return p

            sl = ShardLayer(layer_order_idx=1, shard_name="model-00001.safetensors", layer_name="model.layers.0.w", written=False)
            for tag, cls in (("addition", AdditionMerge), ("task_addition", TaskAdditionMerge)):
                merger = cls(config=cfg, index_manager=idx)

                async def go():
                    with patch.object(idx, "get_tensor", side_effect=fake_get):
                        return await merger._merge_layer(sl, device="cpu")

                out = asyncio.run(go())
                store[f"{case['id']}.{tag}"] = out.contiguous().clone()
            manifest["inputs"][case["id"]] = gi.checksum(base) + sum((gi.checksum(t) for t in fts), [])
    save_file(store, str(OUT / "g9_addition.safetensors"))


def gen_corr(manifest):
    """G10: the reference's correlate_pairs (functions.py:304-314)."""
    manifest["corr"] = {}
    for case in gi.CORR_CASES:
        t = gi.corr_input(case)
        m = ref_fn.correlate_pairs(t.clone(), "cpu", "cpu")
        manifest["corr"][case["id"]] = [[float(v) for v in row] for row in m]
        manifest["inputs"][case["id"]] = gi.checksum(t)


def main():
    if "--only-corr" in sys.argv:
        manifest = json.load(open(OUT / "manifest.json"))
        gen_corr(manifest)
        with open(OUT / "manifest.json", "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if "--only-legacy" in sys.argv:              # add G12 without touching the other fixtures
        manifest = json.load(open(OUT / "manifest.json"))
        gen_legacy(manifest)
        with open(OUT / "manifest.json", "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if "--only-floor" in sys.argv:               # add G11 without touching the other fixtures
        manifest = json.load(open(OUT / "manifest.json"))
        gen_floor(manifest)
        with open(OUT / "manifest.json", "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    if "--only-addition" in sys.argv:            # add G9 without touching the other fixtures
        manifest = json.load(open(OUT / "manifest.json"))
        gen_addition(manifest)
        with open(OUT / "manifest.json", "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        return
    manifest = {"torch": torch.__version__, "threads": torch.get_num_threads(), "inputs": {}}
    gen_fft(manifest)
    gen_interp(manifest)
    gen_slerp(manifest)
    gen_pair(manifest)
    gen_pairs_sched(manifest)
    gen_layers(manifest)
    gen_cli(manifest)
    gen_addition(manifest)
    gen_corr(manifest)
    gen_floor(manifest)
    gen_legacy(manifest)
    with open(OUT / "manifest.json", "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    total = sum(p.stat().st_size for p in OUT.glob("*.safetensors"))
    print(f"golden written: {total/1e6:.2f} MB")


if __name__ == "__main__":
    main()
