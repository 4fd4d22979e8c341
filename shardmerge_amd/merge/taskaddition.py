"""TaskAdditionMerge: AdditionMerge with sign agreement (reference
shard/merge/taskaddition.py:27-83): per element only the deltas whose sign equals the sign of
the sum of signs are added."""
from __future__ import annotations

from .addition import AdditionMerge


class TaskAdditionMerge(AdditionMerge):
    sign_agreement = True
    _how = "from each finetuned model relative to the base model, using sign agreement."
