"""Local model index: lazy per-tensor reads from safetensors shards.

Keeps the interface FourierMerge / MergeTensorsBase use on the reference's
HFMultiModelIndex (shard/index.py:60-276): ``add_model``, ``get_tensor`` ->
promise with ``await .get()``, ``preload_tensor``, ``get_layer_order``,
``get_model_keys``, ``model_indexes``.  Models are read from
``{storage_dir}/{org}/{name}/`` (the reference's local-hit path,
index.py:88-95); there is no network on the GPU box, so a model that is not in
storage is an error instead of a download.  Unlike the reference there is no
unbounded RAM cache: a tensor is read when asked for and handed over.
"""
from __future__ import annotations

import asyncio
import json
import logging
import re
from pathlib import Path
from typing import Dict, List, Optional, Set

import torch
from safetensors import safe_open

logger = logging.getLogger(__name__)


class TensorPromise:
    """Awaitable handle of one tensor (reference index.py:38-58)."""

    def __init__(self, model_uri: str, tensor_name: str, device: str, loader):
        self.model_uri = model_uri
        self.tensor_name = tensor_name
        self.device = device
        self._loader = loader

    async def get(self) -> torch.Tensor:
        tensor = await asyncio.to_thread(self._loader)
        return tensor.to(self.device)


def order_weights(names: List[str]) -> List[str]:
    """embed -> layers by (number, component name) -> model.norm -> lm_head -> rest
    (reference index.py:132-187; components are discovered from layer 0)."""
    embed = sorted(n for n in names if "embed_tokens" in n)
    layered = [n for n in names if "layers." in n]
    norm = sorted(n for n in names if "model.norm.weight" in n)
    head = sorted(n for n in names if "lm_head" in n)
    taken = set(embed) | set(layered) | set(norm) | set(head)
    other = sorted(n for n in names if n not in taken)
    numbers = sorted({int(n.split("layers.")[1].split(".")[0]) for n in layered})
    components = sorted(n[len("model.layers.0."):] for n in layered if n.startswith("model.layers.0."))
    body = [f"model.layers.{i}.{c}" for i in numbers for c in components]
    ordered = embed + body + norm + head + other
    if set(ordered) != set(names):
        raise ValueError(f"Weight ordering mismatch! Missing: {set(names) - set(ordered)}, Extra: {set(ordered) - set(names)}")
    return ordered


class LocalModelIndex:
    def __init__(self, storage_path: Path, cache_path: Optional[Path] = None, download_manager=None):
        self.storage_path = Path(storage_path)
        self.cache_path = Path(cache_path) if cache_path else None
        self.download_manager = download_manager        # accepted for signature compatibility
        self.model_indexes: Dict[str, Dict] = {}
        self._ordered_weights: Dict[str, List[str]] = {}

    async def add_model(self, model_uri: str, revision: str = "main"):
        if model_uri in self.model_indexes:
            return
        index_path = self.storage_path / model_uri / "model.safetensors.index.json"
        if not index_path.exists():
            raise FileNotFoundError(
                f"{index_path} not found: this build reads models from storage_dir only (no network download)")
        with open(index_path) as fh:
            index = json.load(fh)
        self.model_indexes[model_uri] = index
        self._ordered_weights[model_uri] = order_weights(list(index["weight_map"].keys()))
        logger.info(f"Model {model_uri}: {len(set(index['weight_map'].values()))} shards, {len(index['weight_map'])} tensors")

    def _require(self, model_uri: str, tensor_name: Optional[str] = None) -> Dict:
        if model_uri not in self.model_indexes:
            raise KeyError(f"Model {model_uri} not found in index")
        index = self.model_indexes[model_uri]
        if tensor_name is not None and tensor_name not in index["weight_map"]:
            raise KeyError(f"Tensor {tensor_name} not found in model {model_uri}")
        return index

    def get_layer_order(self, model_uri: str) -> List[str]:
        self._require(model_uri)
        return list(self._ordered_weights[model_uri])

    def get_model_keys(self, model_uri: str) -> Set[str]:
        return set(self._require(model_uri)["weight_map"].keys())

    def shard_path(self, model_uri: str, tensor_name: str) -> Path:
        index = self._require(model_uri, tensor_name)
        return self.storage_path / model_uri / index["weight_map"][tensor_name]

    def load_tensor(self, model_uri: str, tensor_name: str) -> torch.Tensor:
        """Synchronous read of one tensor (CPU)."""
        with safe_open(str(self.shard_path(model_uri, tensor_name)), framework="pt") as fh:
            return fh.get_tensor(tensor_name)

    def get_tensor(self, model_uri: str, tensor_name: str, device: str = "cpu") -> TensorPromise:
        self._require(model_uri, tensor_name)
        return TensorPromise(model_uri, tensor_name, device, lambda: self.load_tensor(model_uri, tensor_name))

    async def preload_tensor(self, model_uri: str, tensor_name: str):
        """The reference starts a shard download here; local storage needs nothing."""
        self._require(model_uri, tensor_name)
