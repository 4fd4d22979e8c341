"""AdditionMerge: sum of the finetunes' deltas relative to output_base_model
(reference shard/merge/addition.py:27-83).  As in the reference every tensor goes through the
same path (no passthrough of embeddings / head, no per-model base, alpha unused) and the result
is the DELTA sum - the base is not added back.  One streaming HIP kernel
(``smhip_addition_merge``), arithmetic in the tensors' dtype as torch does it on CPU."""
from __future__ import annotations

import logging

import torch

from ..writer import ShardLayer
from .base import MergeTensorsBase

logger = logging.getLogger(__name__)


class AdditionMerge(MergeTensorsBase):
    sign_agreement = False
    _how = "from each finetuned model relative to the base model."

    def __init__(self, *args, engine=None, **kwargs):
        kwargs = {k: v for k, v in kwargs.items() if k in ("config", "index_manager")}      # run_merge forwards more
        super().__init__(*args, **kwargs)
        self._engine = engine

    def engine(self, device):
        if self._engine is None:
            from ..engine import get_engine
            self._engine = get_engine(device)
        return self._engine

    def _loader_device(self, device: str) -> str:
        return str(self.engine(device).device)

    def get_readme(self) -> str:
        models = "\n".join("- " + m.model for m in self.config.finetune_merge)
        return (f"# Merged Model\n\nBase Model: {self.config.output_base_model}\nFinetuned Models:\n{models}\n\n"
                f"This model was created by computing and combining the delta weights\n{self._how}\n")

    def _layer_requests(self, shard_layer: ShardLayer):
        uris = [self.config.output_base_model] + [m.model for m in self.config.finetune_merge]
        return [(u, shard_layer.layer_name) for u in dict.fromkeys(uris)]

    async def _merge_layer(self, shard_layer: ShardLayer, device: str = "cuda") -> torch.Tensor:
        eng = self.engine(device)
        dev = str(eng.device)
        name = shard_layer.layer_name
        logger.info(f"Processing layer: {name}")
        base = await self._fetch(self.config.output_base_model, name, dev)
        fts = [await self._fetch(m.model, name, dev) for m in self.config.finetune_merge]
        return eng.addition_merge(fts, base, sign_agreement=self.sign_agreement)
