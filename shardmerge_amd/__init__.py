"""shardmerge_amd: MI355X-native (gfx950) implementation of shardmerge's per-layer
spectral-merge hot path behind the reference's own operator / CLI interface."""
__version__ = "0.1.0"

from .constants import DEFAULT_NORM_MODE, NORM_MODES  # noqa: F401
