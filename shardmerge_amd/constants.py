"""Layer sentinels, as the reference defines them (shard/constants.py:4-5)."""
INPUT_LAYER = -1    # model.embed_tokens.weight
OUTPUT_LAYER = -2   # model.norm.weight / lm_head.weight
