"""FourierMerge: the SLERP-FFT merge operator the CLI uses
(reference shard/merge/fast_fourier.py:79-276), with the whole block-tensor
branch of ``_merge_layer`` - deltas, norms, pairing tournament, SLERP-FFT /
Arithmetic-FFT / add branches, add-back, NaN/Inf policy, bf16 cast - executed by
one call into the HIP library (``Engine.merge_layer`` -> ``smhip_merge_layer``).
Intermediates stay in HBM; the reference's TensorDiskCache has no counterpart
(the ``cache_dir`` option is accepted and unused)."""
from __future__ import annotations

import asyncio
import hashlib
import logging
from typing import List, Optional

import torch

from .. import iostats
from ..config import MergeConfig, MergeModel
from ..constants import DEFAULT_NORM_MODE, INPUT_LAYER, OUTPUT_LAYER
from ..index import LocalModelIndex
from ..writer import ShardLayer
from .base import MergeTensorsBase

logger = logging.getLogger(__name__)


def name_hash(name: str) -> str:
    """'model_layer_weight' -> 'mode_laye_weig::<sha256[:8]>' (reference fast_fourier.py:36-41)."""
    short = "_".join(piece[:4] for piece in name.split("_"))
    return f"{short}::{hashlib.sha256(name.encode()).hexdigest()[:8]}"


class FourierMerge(MergeTensorsBase):
    def __init__(self, config: MergeConfig, task_add_models: Optional[List[str]] = None,
                 target_norm_offset: float = 1e-10, cull_start_pct: float = 0.20,
                 index_manager: Optional[LocalModelIndex] = None, engine=None, **kwargs):
        # **kwargs swallows what run_merge forwards from MergeConfig.to_dict()
        super().__init__(config, index_manager)
        self.task_add_models = task_add_models or []
        self.target_norm_offset = target_norm_offset
        self.cull_start_pct = cull_start_pct
        self.cutoff_pct = 0.08          # fast_fourier.py:239
        self.t_sum = 1.0                # fast_fourier.py:238
        self.b = 0.1                    # functions.py:164 (merge_tensors_fft2_slerp's default, never overridden there)
        self.norm_mode = getattr(config, "norm_mode", None) or DEFAULT_NORM_MODE
        # optional YAML overrides (config.merge_options); absent keys keep the reference's values
        for key, value in (getattr(config, "merge_options", None) or {}).items():
            setattr(self, key, float(value))
        self._engine = engine
        self.last_report = None

    def engine(self, device):
        if self._engine is None:
            from ..engine import get_engine
            self._engine = get_engine(device)
        return self._engine

    async def initialize(self):
        """reference base.py:139-162, plus a shape pre-flight: every block tensor's transform
        lengths are checked against the HIP library BEFORE any layer is merged (an unsupported
        length used to surface only when its first layer came up, after earlier shards had been
        written - and, under torchrun, with the other ranks parked in a collective)."""
        await super().initialize()
        self.check_transform_lengths()

    def check_transform_lengths(self):
        import json
        lib = self._engine.lib if self._engine is not None else None
        if lib is None:
            from .. import _lib
            lib = _lib.get_lib()
        uri = self.config.output_base_model
        weight_map = self.index_manager.model_indexes[uri]["weight_map"]
        bad, seen = [], {}
        for shard in sorted(set(weight_map.values())):
            path = self.index_manager.storage_path / uri / shard
            with open(path, "rb") as fh:
                n = int.from_bytes(fh.read(8), "little")
                header = json.loads(fh.read(n))
            for name, rec in header.items():
                if not name.startswith("model.layers."):
                    continue
                shape = tuple(int(v) for v in rec["shape"])
                if not shape:
                    continue
                # a layer only ONE finetune covers (K = 1) is never transformed: any shape goes
                try:
                    number = ShardLayer(0, shard, name, False).layer_number
                except ValueError:
                    continue                  # (the merge itself reports unknown names, as the reference does)
                if sum(1 for m in self.config.finetune_merge if m.use_layer_index(number)) < 2:
                    continue
                if shape not in seen:
                    if len(shape) <= 2:
                        rows, cols = (1, shape[0]) if len(shape) == 1 else shape
                        seen[shape] = lib.shape_supported(rows, cols)
                    else:       # rank > 2: slices, both lengths need a plan
                        seen[shape] = all(lib.length_supported(v) for v in shape[-2:])
                if not seen[shape]:
                    bad.append(f"{name} {list(shape)}")
        if bad:
            raise NotImplementedError(
                f"{len(bad)} block tensor(s) have a shape the HIP library does not support "
                f"(include/shardmerge_hip.h, smhip_shape_supported), "
                f"e.g. {bad[0]}; nothing was merged")

    def get_readme(self) -> str:
        models = "\n".join(f"- {m.model} (vs {m.base})" for m in self.config.finetune_merge)
        return f"# SLERP-FFT Merged Model\nBase: {self.config.output_base_model}\nModels merged:\n{models}\n"

    def _loader_device(self, device: str) -> str:
        return str(self.engine(device).device)

    def _layer_requests(self, shard_layer: ShardLayer):
        """what _merge_layer below fetches (kept next to it on purpose)"""
        number, name = shard_layer.layer_number, shard_layer.layer_name
        if number in (INPUT_LAYER, OUTPUT_LAYER):
            flag = "is_input" if number == INPUT_LAYER else "is_output"
            src = next((m for m in self.config.finetune_merge if getattr(m, flag)), None)
            return [(src.model if src is not None else self.config.output_base_model, name)]
        models = [m for m in self.config.finetune_merge if m.use_layer_index(number)]
        if not models:
            return []
        uris = [m.model for m in models] + [m.base for m in models] + [self.config.output_base_model]
        return [(u, name) for u in dict.fromkeys(uris)]

    async def _passthrough(self, flag: str, shard_layer: ShardLayer, device: str) -> torch.Tensor:
        src = next((m for m in self.config.finetune_merge if getattr(m, flag)), None)
        uri = src.model if src is not None else self.config.output_base_model
        logger.info(f"Passthrough - {shard_layer.layer_name} comes from {uri}")
        return await self._fetch(uri, shard_layer.layer_name, device)

    async def _merge_layer(self, shard_layer: ShardLayer, device: str) -> torch.Tensor:
        number = shard_layer.layer_number
        if number == INPUT_LAYER:
            return await self._passthrough("is_input", shard_layer, device)
        if number == OUTPUT_LAYER:
            return await self._passthrough("is_output", shard_layer, device)

        eng = self.engine(device)
        dev = str(eng.device)
        models = [m for m in self.config.finetune_merge if m.use_layer_index(number)]
        if not models:
            # the reference indexes an empty stack here (IndexError); say what is wrong instead
            raise ValueError(f"No finetune covers layer {number} ({shard_layer.layer_name})")
        name = shard_layer.layer_name
        await asyncio.gather(*(self.index_manager.preload_tensor(m.model, name) for m in models))
        loaded = {}

        async def fetch(uri):
            if uri not in loaded:
                loaded[uri] = await self._fetch(uri, name, dev)
            return loaded[uri]

        fts = [await fetch(m.model) for m in models]
        bases = [await fetch(m.base) for m in models]
        base_out = await fetch(self.config.output_base_model)
        with iostats.timed("merge", 2 * base_out.numel()):       # (the library syncs for its norms: ~ the layer's device time)
            out, report = eng.merge_layer(
                fts, bases, [m.alpha for m in models], base_out,
                target_norm_offset=self.target_norm_offset, cull_start_pct=self.cull_start_pct,
                cutoff_pct=self.cutoff_pct, t_sum=self.t_sum, b=self.b, norm_mode=self.norm_mode, layer_name=name)
        self.last_report = report
        logger.info(f"Merged {name}: {len(models)} model(s), branches {report.branches}, target norm {report.target_norm:.6g}")
        return out
