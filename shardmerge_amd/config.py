"""YAML merge configuration: same keys, defaults and override rules as the
reference's shard/config.py:24-126, so existing config files work unchanged.

    output_base_model: org/base
    finetune_merge:
      - {model: org/ft1, base: org/base, alpha: 0.5, is_input: true}
      - {model: org/ft2, base: org/base, alpha: 0.3, start_layer: 2, end_layer: 30}
    output_dir: merged
    output_dtype: bfloat16      # optional
    device: cuda                # optional (this build always computes on the MI355X)
    cache_dir / storage_dir / clean_cache: optional
    merge_options:              # optional, additive: the operator's hard-coded
      cutoff_pct: 0.08          # hyper-parameters (fast_fourier.py:84-85,238-240); defaults as there
      cull_start_pct: 0.20
      t_sum: 1.0
      target_norm_offset: 1.0e-10
      b: 0.1                    # merge_tensors_fft2_slerp's linear-blend threshold (functions.py:164)
      norm_mode: reference_cpu  # reference_cpu (default: every norm as torch's CPU kernel returns it - the reference's
                                # device="cpu" output) | exact (accurate norms: the reference's device="cuda" numerics)
      operator: fourier         # fourier (default: what the reference CLI hard-wires, __main__.py:22,67)
                                # | addition | task_addition (shard/merge/addition.py, taskaddition.py)
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

import click
import torch
import yaml

from .constants import DEFAULT_NORM_MODE, NORM_MODES

_REQUIRED = ("output_base_model", "finetune_merge", "output_dir")
# the FFT operator's hyper-parameters and the values the reference hard-codes for them
MERGE_OPTION_DEFAULTS = {"cutoff_pct": 0.08, "cull_start_pct": 0.20, "t_sum": 1.0, "target_norm_offset": 1e-10, "b": 0.1}
MERGE_OPTION_RANGES = {"cutoff_pct": (0.0, 1.0), "cull_start_pct": (0.0, 1.0), "t_sum": (-1e6, 1e6), "target_norm_offset": (0.0, 1e6),
                       "b": (0.0, 1e6)}
OPERATORS = ("fourier", "addition", "task_addition", "fourier_legacy")


@dataclass
class MergeModel:
    """One finetune taking part in the merge (reference config.py:24-40)."""
    model: str
    base: str
    alpha: float = 1.0
    is_input: bool = False      # provides model.embed_tokens.weight
    is_output: bool = False     # provides model.norm.weight / lm_head.weight
    is_norm: bool = False       # parsed for compatibility; the reference never reads it
    start_layer: int = 0
    end_layer: int = -1         # -1: no upper bound

    def use_layer_index(self, layer_index: int) -> bool:
        below = layer_index < self.start_layer
        above = self.end_layer != -1 and layer_index > self.end_layer
        return not (below or above)


@dataclass
class MergeConfig:
    finetune_merge: List[MergeModel]
    output_base_model: str
    output_dir: str
    output_dtype: str = "bfloat16"
    device: str = "cpu"
    clean_cache: bool = False
    cache_dir: str = "cache"
    storage_dir: str = "storage"
    merge_options: Dict[str, float] = field(default_factory=dict)
    operator: str = "fourier"
    norm_mode: str = DEFAULT_NORM_MODE
    task_add_models: List[str] = field(default_factory=list)      # operator fourier_legacy only (reference fourier.py:39,115)

    # -- derived views ------------------------------------------------------------
    def _first(self, flag: str) -> Optional[MergeModel]:
        return next((m for m in self.finetune_merge if getattr(m, flag)), None)

    @property
    def input_model(self) -> Optional[MergeModel]:
        return self._first("is_input")

    @property
    def output_model(self) -> Optional[MergeModel]:
        return self._first("is_output")

    @property
    def output_path(self) -> Path:
        return Path(self.output_dir)

    @property
    def cache_path(self) -> Path:
        return Path(self.cache_dir)

    @property
    def storage_path(self) -> Path:
        return Path(self.storage_dir)

    @property
    def output_astype(self) -> torch.dtype:
        return getattr(torch, self.output_dtype)

    # -- mutation / export (reference config.py:83-101) ------------------------------
    def update(self, config: Optional[Dict[str, Any]] = None, **kwargs):
        """Set known attributes from a dict and/or keywords; unknown keys are ignored."""
        for source in (config or {}, kwargs):
            for key, value in source.items():
                if hasattr(self, key):
                    setattr(self, key, value)

    def to_dict(self) -> Dict[str, Any]:
        """What run_merge receives as **kwargs: note finetune_merge collapses to model names."""
        return {
            "output_base_model": self.output_base_model,
            "finetune_merge": [m.model for m in self.finetune_merge],
            "output_dir": self.output_dir,
            "device": self.device,
            "clean_cache": self.clean_cache,
            "cache_dir": self.cache_dir,
            "storage_dir": self.storage_dir,
        }

    @classmethod
    def from_yaml(cls, config_path) -> "MergeConfig":
        with open(config_path) as fh:
            raw = yaml.safe_load(fh) or {}
        absent = [k for k in _REQUIRED if k not in raw]
        if absent:
            raise click.BadParameter(f"Missing required configuration fields: {', '.join(absent)}")
        if not isinstance(raw["finetune_merge"], list):
            raise click.BadParameter("finetune_merge must be a list of model URIs")
        raw["finetune_merge"] = [MergeModel(**entry) for entry in raw["finetune_merge"]]
        opts = dict(raw.get("merge_options") or {})
        norm_mode = opts.pop("norm_mode", DEFAULT_NORM_MODE)
        if norm_mode not in NORM_MODES:
            raise click.BadParameter("merge_options.norm_mode must be 'exact' or 'reference_cpu'")
        raw["norm_mode"] = norm_mode
        task_add = opts.pop("task_add_models", [])
        if not isinstance(task_add, list) or not all(isinstance(x, str) for x in task_add):
            raise click.BadParameter("merge_options.task_add_models must be a list of model names")
        raw["task_add_models"] = task_add
        operator = opts.pop("operator", "fourier")
        if operator not in OPERATORS:
            raise click.BadParameter(f"merge_options.operator must be one of {list(OPERATORS)}")
        raw["operator"] = operator
        unknown = set(opts) - set(MERGE_OPTION_DEFAULTS)
        if not isinstance(opts, dict) or unknown:
            raise click.BadParameter(f"merge_options: unknown keys {sorted(unknown)}; known: {sorted(MERGE_OPTION_DEFAULTS) + ['norm_mode', 'operator']}")
        for key, value in opts.items():
            lo, hi = MERGE_OPTION_RANGES[key]
            if not isinstance(value, (int, float)) or isinstance(value, bool) or not (lo <= float(value) <= hi):
                raise click.BadParameter(f"merge_options.{key} must be a number in [{lo}, {hi}]")
        raw["merge_options"] = {k: float(v) for k, v in opts.items()}
        return cls(**raw)
