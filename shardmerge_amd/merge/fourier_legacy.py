"""The reference's older in-RAM FourierMerge (shard/merge/fourier.py:35-205; not reachable from its CLI, which imports
fast_fourier) behind the same boundary - `merge_options.operator: fourier_legacy`.

What differs from the operator the CLI uses (each point cites the reference line it follows):
  * passthrough layers REQUIRE a finetune flagged is_input / is_output (`:64-67,78-81`: ValueError otherwise);
  * deltas are formed in the tensors' OWN dtype (`ft_tensor -= base_tensor`, `:113`): bf16 arithmetic for bf16
    models, and their first-round norms are bf16 values (`:118`);
  * `target_norm` is the MEDIAN of the delta norms (`:124`), torch's lower median;
  * pairing is least-|cosine similarity| on the tensors themselves (`correlate_pairs`, `:132`;
    `smhip_correlate_pairs`), not on norm products;
  * the SLERP fraction takes alpha of the models at the PAIR's stack positions (`:170-172`), the Arithmetic-FFT
    branch leaves b unscaled (`:165-166`);
  * models named in `task_add_models` skip the tournament and are folded in afterwards with
    `task_arithmetic_fft2(result, delta, t=1, agreement=False)` (`:191-196`);
  * the result is `base + delta` in fp32, NaN -> 0, Inf -> ValueError, NOT cast to bf16 (`:198-205`).
Quirk kept: model i of the FILTERED list is looked up as `finetune_merge[i]` (`:114`) - with layer windows that
is another model than the one loaded; as in the reference.

Every transform, blend and pairing matrix runs in the HIP library through the function-level C ABI
(`smhip_merge_tensors_fft2_slerp`, `smhip_task_arithmetic_fft2`, `smhip_correlate_pairs`, `smhip_addition_merge`,
`smhip_reference_cpu_norm`, `smhip_div_scalar`); torch is used for the handful of element-wise glue ops the reference itself does
between them (scale by target_norm, the add-back) on device tensors."""
from __future__ import annotations

import logging
from typing import List, Optional

import torch

from ..config import MergeConfig
from ..constants import INPUT_LAYER, OUTPUT_LAYER
from ..index import LocalModelIndex
from ..tensor.functions import correlated_pairs
from ..writer import ShardLayer
from .base import MergeTensorsBase

logger = logging.getLogger(__name__)


class LegacyFourierMerge(MergeTensorsBase):
    def __init__(self, config: MergeConfig, task_add_models: Optional[List[str]] = None,
                 target_norm_offset: float = 1e-10, cull_start_pct: float = 0.20,
                 index_manager: Optional[LocalModelIndex] = None, engine=None, **kwargs):
        super().__init__(config, index_manager)
        self.task_add_models = task_add_models or list(getattr(config, "task_add_models", None) or [])
        self.target_norm_offset = target_norm_offset
        self.cull_start_pct = cull_start_pct
        self._engine = engine

    def engine(self, device):
        if self._engine is None:
            from ..engine import get_engine
            self._engine = get_engine(device)
        return self._engine

    def _loader_device(self, device: str) -> str:
        return str(self.engine(device).device)

    def get_readme(self) -> str:                                   # fourier.py:49-55
        models = "\n".join(f"- {m.model}" for m in self.config.finetune_merge)
        return f"# SLERP-FFT Merged Model\nBase: {self.config.output_base_model}\nModels merged:\n{models}\n"

    def _layer_requests(self, shard_layer: ShardLayer):
        number, name = shard_layer.layer_number, shard_layer.layer_name
        if number in (INPUT_LAYER, OUTPUT_LAYER):
            flag = "is_input" if number == INPUT_LAYER else "is_output"
            src = next((m for m in self.config.finetune_merge if getattr(m, flag)), None)
            return [(src.model, name)] if src is not None else []
        uris = [self.config.output_base_model] + [m.model for m in self.config.finetune_merge if m.use_layer_index(number)]
        return [(u, name) for u in dict.fromkeys(uris)]

    def _norm(self, eng, t: torch.Tensor) -> float:
        """`torch.norm(t).item()` as the reference's CPU run gets it: ATen's biased fp32 kernel, bit for bit
        (smhip_reference_cpu_norm); a 16-bit tensor's norm comes back rounded to its dtype."""
        n = eng.reference_cpu_norm(t)
        if t.dtype in (torch.bfloat16, torch.float16):
            n = float(torch.tensor(n, dtype=torch.float32).to(t.dtype))
        return n

    def _slerp_pair(self, eng, v0: torch.Tensor, v1: torch.Tensor, t: float, cull_pct: float, b: float = 0.1) -> torch.Tensor:
        """merge_tensors_fft2_slerp as the legacy operator reaches it (functions.py:164-221), i.e. with tensors still in
        the models' dtype: `normalize_tensor` (functions.py:75-88) then divides a bf16 tensor by its bf16-rounded norm IN
        bf16 - every normalised element is rounded to 8 bits before `fft_transform` upcasts it - and nothing renormalises
        afterwards.  The fused `smhip_merge_tensors_fft2_slerp` normalises fp32 values by their exact norm, so the
        pair goes through the function-level transforms and blend instead (A4, A5-A7, A8)."""
        n0, n1 = self._norm(eng, v0), self._norm(eng, v1)
        v0 = eng.div_scalar(v0, n0) if n0 != 0 else v0                      # (smhip_div_scalar: the tensors' dtype, one rounding)
        v1 = eng.div_scalar(v1, n1) if n1 != 0 else v1
        if n1 < 1e-4 or n0 < 1e-4:
            return v0.float()
        f0, f1 = eng.fft_transform(v0.float()), eng.fft_transform(v1.float())
        if n1 / (n0 + 1e-10) < b:
            spec = f0 + f1 * t
        else:
            spec, _ = eng.interpolate_fft_components(f0, f1, t, t_sum=1.0, cutoff_pct=0.08, cull_pct=cull_pct, interp_imag=True)
        merged = eng.ifft_transform(spec)
        nan = torch.isnan(merged)
        if bool(nan.any()):
            merged = torch.where(nan, torch.zeros_like(merged), merged)
        if bool(torch.isinf(merged).any()):
            raise ValueError("Inf in ifft output")
        return merged

    async def _merge_layer(self, shard_layer: ShardLayer, device: str) -> torch.Tensor:
        number, name = shard_layer.layer_number, shard_layer.layer_name
        if number in (INPUT_LAYER, OUTPUT_LAYER):
            flag, what = ("is_input", "input") if number == INPUT_LAYER else ("is_output", "output")
            src = next((m for m in self.config.finetune_merge if getattr(m, flag)), None)
            if src is None:
                raise ValueError(f"No {what} model found")
            logger.info(f"Passthrough - {name} is an {what} layer, using {src.model} as {what}")
            return await self._fetch(src.model, name, device)

        eng = self.engine(device)
        dev = str(eng.device)
        base = await self._fetch(self.config.output_base_model, name, dev)
        used = [m for m in self.config.finetune_merge if m.use_layer_index(number)]
        layer_stack, add_stack, norms = [], [], []
        for i, m in enumerate(used):
            ft = await self._fetch(m.model, name, dev)
            if ft.dtype != base.dtype:
                raise ValueError(f"{name}: {m.model} is {ft.dtype}, the base {base.dtype} (the legacy operator subtracts in place)")
            delta = eng.addition_merge([ft], base)              # ft - base in the tensors' dtype, one rounding (fourier.py:113)
            model = self.config.finetune_merge[i]               # (quirk, fourier.py:114: position in the FILTERED list)
            if model.model in self.task_add_models:
                add_stack.append((model.model, delta))
            else:
                norms.append(self._norm(eng, delta))
                layer_stack.append((model.model, delta))
        if not layer_stack:
            raise ValueError(f"No finetune takes part in the tournament of layer {number} ({name})")
        target_norm = float(torch.tensor(norms).median().item()) + self.target_norm_offset      # fourier.py:124
        cull_pct = self.cull_start_pct

        while len(layer_stack) > 1:
            corr = eng.correlate_pairs([t for _, t in layer_stack])                              # fourier.py:132
            next_stack = []
            for x, y, _ in correlated_pairs(corr, way="least"):
                if y < 0:
                    next_stack.append(layer_stack[x])
                    continue
                (a_key, a), (b_key, b) = layer_stack[x], layer_stack[y]
                norm_a, norm_b = self._norm(eng, a), self._norm(eng, b)
                if abs(norm_a) < abs(norm_b):
                    a, b, a_key, b_key, norm_a, norm_b = b, a, b_key, a_key, norm_b, norm_a
                cnorm_a, cnorm_b = abs(norm_a / target_norm), abs(norm_b / target_norm)
                n_ratio = cnorm_b / (cnorm_a + 1e-10)
                if cnorm_a < 1e-6:
                    merged = a + b
                    logger.info(f"Merged {a_key} and {b_key}")
                elif cnorm_b < 1e-6 or n_ratio < 0.1:
                    scaled_a = a * target_norm / norm_a                                          # fourier.py:165 (b stays as it is)
                    merged = eng.task_arithmetic_fft2(scaled_a.float(), b.float(), 1.0, agreement=True)
                    logger.info(f"Arithmetic-FFT Merged {a_key} and {b_key} with norm {norm_a} -> {target_norm}")
                else:
                    a_weight = self.config.finetune_merge[x].alpha                               # fourier.py:170-172
                    b_weight = self.config.finetune_merge[y].alpha
                    a_prop = a_weight / (a_weight + b_weight)
                    merged = self._slerp_pair(eng, a, b, a_prop, cull_pct) * target_norm
                    logger.info(f"SLERP-FFT Merged {a_key} and {b_key} with weight {a_prop}")
                next_stack.append((f"{a_key}_{b_key}", merged))
            layer_stack = next_stack
            cull_pct = cull_pct / 2.0

        result = layer_stack[0][1]
        for model_name, delta in add_stack:                                                      # fourier.py:191-196
            result = eng.task_arithmetic_fft2(result.float(), delta.float(), 1.0, agreement=False)
            logger.info(f"Arithmetic Merged {model_name} with weight 1")
        result = base + result                                     # (bf16 + fp32 -> fp32; a 16-bit delta alone stays 16-bit, as there)
        nan = torch.isnan(result)
        if bool(nan.any()):
            result = torch.where(nan, torch.zeros_like(result), result)
        if bool(torch.isinf(result).any()):
            raise ValueError(f"Inf in merged tensor for {name}")
        return result
