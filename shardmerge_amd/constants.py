"""Layer sentinels, as the reference defines them (shard/constants.py:4-5)."""
INPUT_LAYER = -1    # model.embed_tokens.weight
OUTPUT_LAYER = -2   # model.norm.weight / lm_head.weight

# THE default numerics of every entry point (YAML, CLI, Engine.merge_layer, FourierMerge, the multi-GPU stamp, bench.py):
# every norm as torch's CPU kernel returns it - the reference's device="cpu" output depends on those at the 1e-2
# level.  "exact": accurate L2 norms, the reference's device="cuda" numerics.
DEFAULT_NORM_MODE = "reference_cpu"
NORM_MODES = ("exact", "reference_cpu")


def tune_hip_queues(n: int = 16) -> None:
    """One hardware queue per stream for the entry points that drive 8 engines (CLI, bench): ROCm's default is 4
    queues per process (+1 % on the default bench with 16, tools/ab_hwq.sh).  Read by the HIP runtime when it
    initialises, so this must run before the first GPU call; a value the user has set wins.  Called by the entry
    points, not on import: a library must not change its host's environment."""
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(n))
