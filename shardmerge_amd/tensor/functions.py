"""Function-level mirror of the reference's shard/tensor/functions.py: same
names, argument order and error behaviour, computed by the HIP library.

Differences a caller can see: results stay on the compute device (the reference
moves every intermediate back to the CPU, functions.py:56-58), and ``device`` is
only a hint - this build always computes on the MI355X.  ``slerp`` and
``normalize_tensor`` have kernels of their own at this level (inside the merge they are
fused into the transform and blend kernels).
"""
from __future__ import annotations

from typing import Generator, Literal, Tuple

import torch

from ..engine import get_engine


def slerp(v0: torch.Tensor, v1: torch.Tensor, t: float, device: str = "cuda") -> torch.Tensor:
    """reference functions.py:24-43 (quirk Q5 kept: cosine of un-normalised vectors,
    unit-length relative vector); ``smhip_slerp``."""
    return get_engine(device).slerp(v0, v1, t)


def normalize_tensor(tensor: torch.Tensor, device: str = "cuda") -> Tuple[torch.Tensor, float]:
    """reference functions.py:75-88: ``smhip_reference_cpu_norm`` (or ``smhip_exact_norm``: the engine's
    norm_mode default decides) and ``smhip_div_scalar``."""
    return get_engine(device).normalize_tensor(tensor)


def fft_transform(tensor: torch.Tensor, device: str = "cuda") -> torch.Tensor:
    """reference functions.py:45-58: fft (1-D) / fftn over the last two dims -> complex64."""
    return get_engine(device).fft_transform(tensor)


def ifft_transform(tensor: torch.Tensor, device: str = "cuda") -> torch.Tensor:
    """reference functions.py:60-73: real part of the inverse transform."""
    return get_engine(device).ifft_transform(tensor)


def interpolate_fft_components(v0_fft, v1_fft, t: float, device: str = "cuda", t_sum: float = 1.0,
                               cutoff_pct: float = 0.0, cull_pct: float = 0.0, interp_imag: bool = True) -> torch.Tensor:
    """reference functions.py:90-162."""
    out, _ = get_engine(device).interpolate_fft_components(v0_fft, v1_fft, t, t_sum, cutoff_pct, cull_pct, interp_imag)
    return out


def merge_tensors_fft2_slerp(v0, v1, t: float, device: str = "cuda", b: float = 0.1, t_sum: float = 1.0,
                             cutoff_pct: float = 0.0, cull_pct: float = 0.0) -> Tuple[torch.Tensor, float, float]:
    """reference functions.py:164-221; raises ValueError("Inf in ifft output") like it."""
    out, n0, n1, _ = get_engine(device).merge_tensors_fft2_slerp(v0, v1, t, b=b, t_sum=t_sum, cutoff_pct=cutoff_pct, cull_pct=cull_pct)
    return out, n0, n1


def task_arithmetic_fft2(v0, v1, t: float, device: str = "cuda", agreement: bool = True) -> torch.Tensor:
    """reference functions.py:224-254."""
    return get_engine(device).task_arithmetic_fft2(v0, v1, t, agreement=agreement)


def arithmetic_fft_components(v0_fft, v1_fft, t: float, agreement: bool, device: str = "cuda", do_imag: bool = True) -> torch.Tensor:
    """reference functions.py:256-302 (quirk Q3 kept: disagreeing bins take v1)."""
    return get_engine(device).arithmetic_fft_components(v0_fft, v1_fft, t, agreement=agreement, do_imag=do_imag)


def correlate_pairs(tensors: torch.Tensor, work_device: str = "cuda", store_device: str = "cpu") -> torch.Tensor:
    """reference functions.py:304-314: matrix[i, j] = mean of cosine_similarity(t_i, t_j, dim=0)
    (NaN -> 0), zero diagonal; one HIP kernel pair instead of K^2/2 host round trips."""
    return get_engine(work_device).correlate_pairs(tensors).to(store_device)


def correlated_pairs(correlation_matrix: torch.Tensor, way: Literal["least", "most"] = "least") -> Generator[Tuple[int, int, float], None, None]:
    """Greedy pairing of a K x K matrix (reference functions.py:316-365): repeatedly take
    the first (row-major) unused upper-triangle cell whose |value| is minimal ('least') or
    maximal ('most'); leftover indices come out as (i, -1, m[i, i]).  Host-side: K <= 16."""
    if way not in ("least", "most"):
        raise ValueError("Invalid way. Choose 'least' or 'most'.")
    m = correlation_matrix.detach().to("cpu")
    k = m.size(0)
    used = [False] * k
    while True:
        cells = [(i, j) for i in range(k) for j in range(i + 1, k) if not used[i] and not used[j]]
        if not cells:
            break
        mags = [abs(m[i, j].item()) for i, j in cells]
        want = min(mags) if way == "least" else max(mags)
        pick = next((c for c, v in zip(cells, mags) if v == want), None)
        if pick is None:
            break
        x, y = pick
        yield (x, y, m[x, y].item())
        used[x] = used[y] = True
    for i in range(k):
        if not used[i]:
            yield (i, -1, m[i, i].item())
