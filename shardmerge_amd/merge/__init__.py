"""Merge operators behind the reference's MergeTensorsBase interface."""


def operator_class(name: str = "fourier"):
    """merge_options.operator -> class (the reference CLI hard-wires FourierMerge, __main__.py:22,67)."""
    if name == "fourier":
        from .fast_fourier import FourierMerge
        return FourierMerge
    if name == "addition":
        from .addition import AdditionMerge
        return AdditionMerge
    if name == "task_addition":
        from .taskaddition import TaskAdditionMerge
        return TaskAdditionMerge
    if name == "fourier_legacy":
        from .fourier_legacy import LegacyFourierMerge
        return LegacyFourierMerge
    raise ValueError(f"unknown merge operator {name!r}")
