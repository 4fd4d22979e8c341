"""`python -m shardmerge_amd merge CONFIG` (and `python -m shard merge CONFIG`
through the alias package): the merge entry point of the reference CLI
(shard/__main__.py:78-158) with the same arguments and options."""
from __future__ import annotations

import asyncio
import logging
import traceback
from pathlib import Path
from typing import Optional

import click

from .config import MergeConfig
from .index import LocalModelIndex
from .merge.fast_fourier import FourierMerge

logger = logging.getLogger(__name__)


def setup_logging(verbose: bool):
    logging.basicConfig(level=logging.DEBUG if verbose else logging.INFO,
                        format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")


async def run_merge(config: MergeConfig, device: str, clean_cache: bool, **kwargs):
    """Build the index and the operator, run the merge (reference __main__.py:47-76).
    With more than one rank (torchrun) the tensor list is partitioned over the GPUs."""
    index_manager = LocalModelIndex(storage_path=config.storage_path, cache_path=config.cache_path)
    from . import distributed
    import os
    # SHARDMERGE_INPLACE=1: a single process takes the multi-GPU path's in-place output shards too (pre-sized files,
    # every tensor pwritten at its offset as soon as its copy to the host is done) instead of writing a shard when it
    # is complete - one file takes ~7 GB/s on tmpfs whatever the thread count, so the sooner its writes start the better
    # (tools/cli_bench.py: 5.1 against 3.9 GB/s of merged weights on the Llama-3-70B slice)
    if distributed.world_size() > 1 or os.environ.get("SHARDMERGE_FORCE_DIST") == "1" or os.environ.get("SHARDMERGE_INPLACE") == "1":
        await distributed.run_partitioned_merge(config, index_manager, device)
        return
    from .merge import operator_class
    merger = operator_class(config.operator)(config=config, index_manager=index_manager, **kwargs)
    await merger.merge(device=device)


@click.group()
def cli():
    """Shard merge utility (MI355X-native merge path)."""


@cli.command("merge")
@click.argument("config_file", type=click.Path(exists=True, path_type=Path))
@click.option("--cache-dir", type=click.Path(path_type=Path), default=None, help="Directory for caching downloaded files")
@click.option("--clean_cache", is_flag=True, help="Delete cached files after merging")
@click.option("--device", type=str, default=None, help="Device to perform tensor operations on (cuda/cpu)")
@click.option("--verbose", is_flag=True, help="Enable verbose logging")
def merge_command(config_file: Path, cache_dir: Optional[Path], verbose: bool, **kwargs):
    """Merge multiple finetuned models by computing and combining their deltas.

    CONFIG_FILE is a YAML file with output_base_model, finetune_merge (list of
    {model, base, alpha, is_input, is_output, start_layer, end_layer}) and output_dir.
    """
    setup_logging(verbose)
    from .constants import tune_hip_queues
    tune_hip_queues()                   # before the first GPU call (8 engines, 8 side streams)
    try:
        config = MergeConfig.from_yaml(config_file)
        logger.info(f"Loaded configuration: {config}")
        if cache_dir:
            config.cache_dir = cache_dir
        config.update({k: v for k, v in kwargs.items() if v is not None})
        asyncio.run(run_merge(config=config, **config.to_dict()))
    except Exception as exc:
        logging.error(f"Error during merge: {exc}", exc_info=verbose)
        traceback.print_exc()
        raise click.Abort()


if __name__ == "__main__":
    cli()
