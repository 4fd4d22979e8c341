"""Merge-operator base class: the drop-in boundary of the path
(reference shard/merge/base.py:96-223).  Subclasses implement
``async _merge_layer(shard_layer, device) -> Tensor`` and ``get_readme()``;
``merge(device)`` drives initialize -> per-shard loop -> finalize -> README."""
from __future__ import annotations

import logging
import os
from abc import ABC, abstractmethod
from typing import List, Optional, Tuple

import torch

from ..config import MergeConfig, MergeModel
from ..index import LocalModelIndex
from ..writer import ModelWriter, ShardLayer

logger = logging.getLogger(__name__)


class MergeTensorsBase(ABC):
    def __init__(self, config: MergeConfig, index_manager: Optional[LocalModelIndex] = None):
        self.config = config
        self.index_manager = index_manager or LocalModelIndex(config.storage_path, config.cache_path)
        self._loader = None             # PrefetchLoader while merge() runs (loader.py)
        self._layer_pos = {}

    @abstractmethod
    def get_readme(self) -> str:
        return "No readme defined"

    @abstractmethod
    async def _merge_layer(self, shard_layer: ShardLayer, device: str) -> torch.Tensor:
        raise NotImplementedError

    def _layer_requests(self, shard_layer: ShardLayer) -> List[Tuple[str, str]]:
        """(model uri, tensor name) pairs `_merge_layer` will fetch for this layer, in any order;
        operators that list them get their inputs prefetched (loader.py), others load on demand."""
        return []

    async def _fetch(self, model_uri: str, layer_name: str, device: str) -> torch.Tensor:
        if self._loader is not None:
            t = self._loader.take(model_uri, layer_name)
            if t is not None:
                return t
        return await self.index_manager.get_tensor(model_uri, layer_name, device=device).get()

    async def get_base_output_tensor(self, shard_layer: ShardLayer, device: str) -> torch.Tensor:
        """fp32 tensor of output_base_model (reference base.py:117-119)."""
        return (await self._fetch(self.config.output_base_model, shard_layer.layer_name, device)).to(torch.float32)

    async def get_delta_for_models(self, models: List[MergeModel], shard_layer: ShardLayer, device: str,
                                   apply_alpha: bool = True) -> List[torch.Tensor]:
        """fp32 (finetune - base) per model, times alpha on request (reference base.py:121-137).
        Kept for API compatibility; the HIP operator forms its deltas on the device instead."""
        bases, out = {}, []
        for m in models:
            if m.base not in bases:
                bases[m.base] = (await self._fetch(m.base, shard_layer.layer_name, device)).to(torch.float32)
            ft = (await self._fetch(m.model, shard_layer.layer_name, device)).to(torch.float32)
            out.append((ft - bases[m.base]).detach() * (m.alpha if apply_alpha else 1))
        return out

    async def initialize(self):
        cfg = self.config
        await self.index_manager.add_model(cfg.output_base_model)
        self.index_doc = self.index_manager.model_indexes[cfg.output_base_model]
        for m in cfg.finetune_merge:
            await self.index_manager.add_model(m.base)
            await self.index_manager.add_model(m.model)
        base_keys = self.index_manager.get_model_keys(cfg.output_base_model)
        for m in cfg.finetune_merge:
            keys = self.index_manager.get_model_keys(m.model)
            if keys != base_keys:
                # (the reference means to raise this ValueError too but trips over a missing
                # attribute first - SURVEY quirk Q9; the intended error is raised here)
                raise ValueError(
                    f"Model {m.model} architecture mismatch with base model {cfg.output_base_model}\n"
                    f"Missing keys: {base_keys - keys}\nExtra keys: {keys - base_keys}")

    def _loader_device(self, device: str) -> str:
        """device the prefetched tensors must land on (operators with an engine override this)"""
        return device

    def get_writer(self, layer_order: List[str]) -> ModelWriter:
        return ModelWriter(base_index=self.index_doc, output_path=self.config.output_path,
                           layer_order=layer_order, output_astype=self.config.output_astype)

    async def merge(self, device: str):
        await self.initialize()
        layer_order = self.index_manager.get_layer_order(self.config.output_base_model)
        writer = self.get_writer(layer_order)
        groups = [[sl for sl in group if not sl.written] for group in writer.shard_layers()]
        todo = [sl for group in groups for sl in group]
        schedule = [self._layer_requests(sl) for sl in todo]
        if os.environ.get("SHARDMERGE_PREFETCH", "1") != "0" and any(schedule):
            from ..loader import PrefetchLoader
            self._layer_pos = {sl.layer_name: i for i, sl in enumerate(todo)}
            self._loader = PrefetchLoader(self.index_manager, self._loader_device(device))
            self._loader.start(schedule)
        try:
            for group in groups:
                await self._process_layers(writer, group, device)
        finally:
            if self._loader is not None:
                self._loader.close()
                self._loader = None
        writer.finalize()
        readme = self.get_readme() or "No README defined"
        with open(self.config.output_path / "README.md", "w") as fh:
            fh.write(readme)
        logger.info(f"Merge complete. Output saved to {self.config.output_path}")

    async def _process_layers(self, writer: ModelWriter, shard_layers: List[ShardLayer], device: str):
        shard_layer = None
        try:
            for shard_layer in shard_layers:
                if self._loader is not None:
                    self._loader.begin_layer(self._layer_pos[shard_layer.layer_name])
                out = await self._merge_layer(shard_layer, device)
                writer.add_tensor(shard_layer.layer_name, out)
                del out
        except Exception as exc:
            logger.error(f"Error processing {shard_layer.layer_name if shard_layer else '?'}: {exc}")
            raise
