"""shardmerge_amd: MI355X-native (gfx950) implementation of shardmerge's per-layer
spectral-merge hot path behind the reference's own operator / CLI interface."""
__version__ = "0.1.0"

import os as _os

# One hardware queue per stream: the bench and the CLI drive the GPU from 8 engines (8 streams, plus 8 side streams in
# norm_mode = reference_cpu), ROCm's default is 4 queues per process and streams beyond that share one (measured on
# MI355X, default bench: +1 % with 16 queues, tools/ab_hwq.sh).  Read by the HIP runtime when it initialises, i.e.
# after this import; a value the user has set wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
