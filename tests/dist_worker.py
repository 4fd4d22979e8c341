"""Worker of tests/test_distributed_gloo.py: one rank of a gloo job that runs the partitioned merge with the CPU
work-group emulator as its device (or, SHARDMERGE_TEST_REAL_ENGINE=1, the HIP library on the one GPU of the box)."""
import asyncio
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

import torch  # noqa: E402

from shardmerge_amd import distributed  # noqa: E402
from shardmerge_amd.config import MergeConfig  # noqa: E402
from shardmerge_amd.index import LocalModelIndex  # noqa: E402
from tests.emul.loader import emul_engine  # noqa: E402


def main():
    import logging
    logging.basicConfig(level=logging.INFO)
    torch.set_num_threads(1)
    import os
    cfg = MergeConfig.from_yaml(sys.argv[1])
    real = os.environ.get("SHARDMERGE_TEST_REAL_ENGINE") == "1"      # the GPU tier: real engines, several ranks on one card (gloo)
    if not real:
        distributed.ENGINE_FACTORY = emul_engine
        if os.environ.get("SHARDMERGE_TEST_FAULT_RANK") == os.environ.get("RANK"):
            # fault injection (test_a_failing_rank_stops_the_others): this rank's second merge raises
            def faulty():
                eng = emul_engine()
                real_merge, calls = eng.merge_layer, [0]

                def merge_layer(*a, **k):
                    calls[0] += 1
                    if calls[0] >= 2:
                        raise RuntimeError("injected fault")
                    return real_merge(*a, **k)
                eng.merge_layer = merge_layer
                return eng
            distributed.ENGINE_FACTORY = faulty
    idx = LocalModelIndex(cfg.storage_path)
    asyncio.run(distributed.run_partitioned_merge(cfg, idx, "cuda" if real else "cpu"))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
