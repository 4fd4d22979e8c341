"""Alias package: `python -m shard merge config.yaml` keeps working as with the
reference, and runs the MI355X-native path in `shardmerge_amd`."""
