"""Engine: torch-tensor front end of the C ABI (include/shardmerge_hip.h).

PyTorch is used here for device memory and streams only: tensors are handed to
the HIP library as raw device pointers (``Tensor.data_ptr()``) together with the
current HIP stream; every transform, order statistic, blend and cast runs in the
hand-written gfx950 kernels behind ``libshardmerge_hip.so``.
"""
from __future__ import annotations

import ctypes as C
import logging
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .constants import DEFAULT_NORM_MODE, NORM_MODES
from ._lib import BlendInfo, Context, LayerDesc, LayerReport, SmhipError, SmhipLibrary

logger = logging.getLogger(__name__)

_DTYPE_CODE = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16, torch.float32: _lib.F32}


def _raise_like_reference(e: SmhipError, layer_name: Optional[str] = None):
    """Map C-ABI status codes to the exceptions the reference raises."""
    if e.code == _lib.ERR_INF_IFFT:
        raise ValueError("Inf in ifft output") from e                      # functions.py:217
    if e.code == _lib.ERR_INF_MERGED:
        raise ValueError(f"Inf in merged tensor for {layer_name}") from e  # fast_fourier.py:274
    if e.code == _lib.ERR_SHAPE:
        raise NotImplementedError(e.message) from e
    if e.code == _lib.ERR_NONFINITE:
        # the reference spins forever here (no pair is ever found among NaN norms,
        # fast_fourier.py:171-254); a loud error instead - INTEGRATION.md "deviations"
        raise ValueError(f"Non-finite delta norm in {layer_name}: {e.message}") from e
    raise e


@dataclass
class BlendReport:
    cutoff_threshold: float = 0.0
    cull_threshold: float = 0.0
    dot: float = 0.0
    s00: float = 0.0
    s01: float = 0.0
    s11: float = 0.0
    n_slerp: int = 0
    t: float = 0.0
    cull_pct: float = 0.0

    @classmethod
    def from_c(cls, bi: BlendInfo) -> "BlendReport":
        return cls(bi.cutoff_threshold, bi.cull_threshold, bi.dot, bi.s00, bi.s01, bi.s11, int(bi.n_slerp),
                   bi.t, bi.cull_pct)


@dataclass
class LayerMergeReport:
    target_norm: float = 0.0
    delta_norms: List[float] = field(default_factory=list)
    steps: List[Tuple[int, int, str]] = field(default_factory=list)   # (x, y, branch)
    infos: List[BlendReport] = field(default_factory=list)
    nan_ifft: int = 0
    nan_final: int = 0
    merged_delta_norm: float = -1.0

    @property
    def branches(self) -> List[str]:
        return [s[2] for s in self.steps]


def _shape2d(t: torch.Tensor) -> Tuple[int, int]:
    if t.ndim == 1:
        return 1, t.shape[0]
    if t.ndim == 2:
        return t.shape[0], t.shape[1]
    # (merge_layer and the plain transforms take rank > 2 through _shape3d)
    raise NotImplementedError(f"tensor of rank {t.ndim} is not supported by this entry point of the HIP path")


def _shape3d(t: torch.Tensor) -> Tuple[int, int, int]:
    """(batch, rows, cols): the reference transforms the LAST TWO dims of an N-D tensor
    (functions.py:55-58); every leading dim is batch."""
    if t.ndim <= 2:
        r, c = _shape2d(t)
        return 1, r, c
    b = 1
    for d in t.shape[:-2]:
        b *= int(d)
    return b, int(t.shape[-2]), int(t.shape[-1])


class Engine:
    """One smhip context bound to one torch device."""

    def __init__(self, lib: Optional[SmhipLibrary] = None, device: Optional[torch.device] = None):
        if lib is None:
            lib = _lib.get_lib()
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device())
            if device.type != "cuda":
                raise RuntimeError("the HIP library needs a cuda (ROCm) torch device")
        self.lib = lib
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        index = self.device.index if self.device.type == "cuda" and self.device.index is not None else 0
        self.ctx = Context(lib, index)

    # -- plumbing ---------------------------------------------------------------
    def _stream(self) -> Optional[int]:
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _dev(self, t: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        if t.device != self.device:
            t = t.to(self.device)
        return t.contiguous()

    def _call(self, rc: int, layer_name: Optional[str] = None):
        try:
            self.ctx.check(rc)
        except SmhipError as e:
            _raise_like_reference(e, layer_name)

    # -- A4 / A8 -----------------------------------------------------------------
    def fft_transform(self, x: torch.Tensor) -> torch.Tensor:
        x = self._dev(x, torch.float32)
        b, r, c = _shape3d(x)
        out = torch.empty(x.shape + (2,), dtype=torch.float32, device=self.device)
        xs, os_ = x.reshape(b, -1), out.reshape(b, -1)
        for i in range(b):                        # slices of a rank > 2 tensor are independent transforms
            self._call(self.lib.dll.smhip_fft_transform(self.ctx.h, xs[i].data_ptr(), r, c, os_[i].data_ptr(), self._stream()))
        return torch.view_as_complex(out)

    def ifft_transform(self, spec: torch.Tensor) -> torch.Tensor:
        spec = self._dev(spec, torch.complex64)
        b, r, c = _shape3d(spec)
        sr = torch.view_as_real(spec)
        out = torch.empty(spec.shape, dtype=torch.float32, device=self.device)
        ss, os_ = sr.reshape(b, -1), out.reshape(b, -1)
        for i in range(b):
            self._call(self.lib.dll.smhip_ifft_transform(self.ctx.h, ss[i].data_ptr(), r, c, os_[i].data_ptr(), self._stream()))
        return out

    # -- A5 - A7 -------------------------------------------------------------------
    def interpolate_fft_components(self, f0, f1, t, t_sum=1.0, cutoff_pct=0.0, cull_pct=0.0, interp_imag=True):
        f0 = self._dev(f0, torch.complex64)
        f1 = self._dev(f1, torch.complex64)
        r, c = _shape2d(f0)
        out = torch.empty(f0.shape + (2,), dtype=torch.float32, device=self.device)
        info = BlendInfo()
        self._call(self.lib.dll.smhip_interpolate_fft_components(
            self.ctx.h, torch.view_as_real(f0).data_ptr(), torch.view_as_real(f1).data_ptr(), r, c,
            float(t), float(t_sum), float(cutoff_pct), float(cull_pct), 1 if interp_imag else 0,
            out.data_ptr(), C.byref(info), self._stream()))
        return torch.view_as_complex(out), BlendReport.from_c(info)

    def arithmetic_fft_components(self, f0, f1, t, agreement=True, do_imag=True):
        f0 = self._dev(f0, torch.complex64)
        f1 = self._dev(f1, torch.complex64)
        r, c = _shape2d(f0)
        out = torch.empty(f0.shape + (2,), dtype=torch.float32, device=self.device)
        self._call(self.lib.dll.smhip_arithmetic_fft_components(
            self.ctx.h, torch.view_as_real(f0).data_ptr(), torch.view_as_real(f1).data_ptr(), r, c,
            float(t), 1 if agreement else 0, 1 if do_imag else 0, out.data_ptr(), self._stream()))
        return torch.view_as_complex(out)

    # -- A9 / A10 ------------------------------------------------------------------
    def merge_tensors_fft2_slerp(self, v0, v1, t, b=0.1, t_sum=1.0, cutoff_pct=0.0, cull_pct=0.0):
        v0 = self._dev(v0, torch.float32)
        v1 = self._dev(v1, torch.float32)
        r, c = _shape2d(v0)
        out = torch.empty_like(v0)
        n0, n1, br = C.c_double(), C.c_double(), C.c_int()
        info = BlendInfo()
        self._call(self.lib.dll.smhip_merge_tensors_fft2_slerp(
            self.ctx.h, v0.data_ptr(), v1.data_ptr(), r, c, float(t), float(b), float(t_sum), float(cutoff_pct),
            float(cull_pct), out.data_ptr(), C.byref(n0), C.byref(n1), C.byref(br), C.byref(info), self._stream()))
        rep = BlendReport.from_c(info)
        rep.branch = _lib.BRANCH_NAMES.get(br.value, str(br.value))
        return out, n0.value, n1.value, rep

    def task_arithmetic_fft2(self, v0, v1, t, agreement=True):
        v0 = self._dev(v0, torch.float32)
        v1 = self._dev(v1, torch.float32)
        r, c = _shape2d(v0)
        out = torch.empty_like(v0)
        self._call(self.lib.dll.smhip_task_arithmetic_fft2(
            self.ctx.h, v0.data_ptr(), v1.data_ptr(), r, c, float(t), 1 if agreement else 0, out.data_ptr(), self._stream()))
        return out

    # -- torch.norm as ATen's CPU kernel computes it (norm_mode = reference_cpu) -----------------
    def reference_cpu_norm(self, x: torch.Tensor, base: Optional[torch.Tensor] = None) -> float:
        """``torch.norm(x - base)`` exactly as the reference's ``device="cpu"`` run gets it for a contiguous
        fp32 tensor (functions.py:85, fast_fourier.py:152,209-210): bit-identical, computed in parallel."""
        dtype = x.dtype if x.dtype in _DTYPE_CODE and (base is None or base.dtype == x.dtype) else torch.float32
        xs = self._dev(x, dtype)
        bs = self._dev(base, dtype) if base is not None else None
        if bs is not None and bs.shape != xs.shape:
            raise ValueError(f"shape mismatch: {tuple(xs.shape)} vs {tuple(bs.shape)}")
        out = C.c_float(0.0)
        self._call(self.lib.dll.smhip_reference_cpu_norm(self.ctx.h, xs.data_ptr(), bs.data_ptr() if bs is not None else None,
                                                         _DTYPE_CODE[dtype], xs.numel(), C.byref(out), self._stream()))
        return float(out.value)

    # -- A1 / A8 at function level ----------------------------------------------------------------
    def slerp(self, v0: torch.Tensor, v1: torch.Tensor, t: float) -> torch.Tensor:
        """reference functions.py:24-43 (quirk Q5 kept); the relative vector is normalised along the last dimension."""
        if v0.shape != v1.shape:
            raise ValueError(f"shape mismatch: {tuple(v0.shape)} vs {tuple(v1.shape)}")
        a, b = self._dev(v0, torch.float32), self._dev(v1, torch.float32)
        out = torch.empty_like(a)
        cols = a.shape[-1] if a.dim() >= 1 else 1
        rows = a.numel() // cols if cols else 0
        self._call(self.lib.dll.smhip_slerp(self.ctx.h, a.data_ptr(), b.data_ptr(), rows, cols if a.numel() else 0, float(t),
                                            out.data_ptr(), self._stream()))
        return out

    def exact_norm(self, x: torch.Tensor) -> float:
        """||x||_2, accumulated in fp64 on the device"""
        dtype = x.dtype if x.dtype in _DTYPE_CODE else torch.float32
        xs = self._dev(x, dtype)
        out = C.c_double(0.0)
        self._call(self.lib.dll.smhip_exact_norm(self.ctx.h, xs.data_ptr(), _DTYPE_CODE[dtype], xs.numel(), C.byref(out), self._stream()))
        return float(out.value)

    def div_scalar(self, x: torch.Tensor, s: float) -> torch.Tensor:
        """``x / s`` in x's dtype, as torch divides a tensor by a Python float (fp32 division, one rounding)"""
        dtype = x.dtype if x.dtype in _DTYPE_CODE else torch.float32
        xs = self._dev(x, dtype)
        out = torch.empty_like(xs)
        self._call(self.lib.dll.smhip_div_scalar(self.ctx.h, xs.data_ptr(), _DTYPE_CODE[dtype], xs.numel(), float(s), out.data_ptr(), self._stream()))
        return out

    def normalize_tensor(self, x: torch.Tensor, norm_mode: Optional[str] = None):
        """reference functions.py:75-88: (x / norm, norm), norm = ``x.norm().item()`` - in ``reference_cpu`` mode the value
        the reference's CPU run gets (ATen's biased fp32 kernel; a 16-bit tensor's norm is rounded to its dtype), else
        the exact one.  norm == 0: x comes back unchanged."""
        mode = DEFAULT_NORM_MODE if norm_mode is None else norm_mode
        if mode not in NORM_MODES:
            raise ValueError(f"norm_mode {mode!r}: one of {NORM_MODES}")
        if mode == "reference_cpu":
            norm = self.reference_cpu_norm(x)
            if x.dtype in (torch.bfloat16, torch.float16):
                norm = float(torch.tensor(norm, dtype=torch.float32).to(x.dtype))
        else:
            norm = self.exact_norm(x)
        return (self.div_scalar(x, norm) if norm != 0 else self._dev(x, x.dtype if x.dtype in _DTYPE_CODE else torch.float32)), norm

    # -- N3: AdditionMerge / TaskAdditionMerge ---------------------------------------------
    def addition_merge(self, finetunes: Sequence[torch.Tensor], base: torch.Tensor, sign_agreement: bool = False) -> torch.Tensor:
        """sum_i (finetune_i - base) in the tensors' dtype, optionally masked by the majority sign
        (reference addition.py:70-76 / taskaddition.py:69-79).  The base is not added back."""
        k = len(finetunes)
        if k < 1 or k > _lib.MAX_MODELS:
            raise ValueError(f"{k} models to merge: supported range is 1..{_lib.MAX_MODELS}")
        dtypes = {t.dtype for t in list(finetunes) + [base]}
        dtype = next(iter(dtypes)) if len(dtypes) == 1 else torch.promote_types(*dtypes) if len(dtypes) == 2 else torch.float32
        if dtype not in _DTYPE_CODE:
            dtype = torch.float32
        bs = self._dev(base, dtype)
        fts = [self._dev(t, dtype) for t in finetunes]
        for t in fts:
            if t.shape != bs.shape:
                raise ValueError(f"shape mismatch: {tuple(t.shape)} vs {tuple(bs.shape)}")
        ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in fts])
        out = torch.empty_like(bs)
        self._call(self.lib.dll.smhip_addition_merge(self.ctx.h, k, ptrs, bs.data_ptr(), _DTYPE_CODE[dtype], bs.numel(),
                                                     1 if sign_agreement else 0, out.data_ptr(), self._stream()))
        return out

    def correlate_pairs(self, tensors) -> torch.Tensor:
        """K x K matrix of mean column-wise cosine similarities (reference functions.py:304-314);
        `tensors`: a stacked tensor [K, ...] or a sequence of K equally shaped tensors."""
        ts = list(tensors.unbind(0)) if isinstance(tensors, torch.Tensor) else list(tensors)
        k = len(ts)
        if k < 2 or k > 8:
            raise ValueError("correlate_pairs: 2..8 tensors")
        dtype = ts[0].dtype if ts[0].dtype in _DTYPE_CODE and all(t.dtype == ts[0].dtype for t in ts) else torch.float32
        ts = [self._dev(t, dtype) for t in ts]
        rows = ts[0].shape[0] if ts[0].ndim >= 1 else 1
        cols = ts[0].numel() // max(rows, 1)
        ptrs = (C.c_void_p * k)(*[t.data_ptr() for t in ts])
        out = (C.c_float * (k * k))()
        self._call(self.lib.dll.smhip_correlate_pairs(self.ctx.h, k, ptrs, _DTYPE_CODE[dtype], rows, cols, out, self._stream()))
        return torch.tensor(list(out), dtype=torch.float32).reshape(k, k)

    # -- A1 - A13 fused ------------------------------------------------------------------
    def merge_layer(self, finetunes: Sequence[torch.Tensor], bases: Sequence[torch.Tensor], alphas: Sequence[float],
                    base_out: torch.Tensor, target_norm_offset: float = 1e-10, cull_start_pct: float = 0.20,
                    cutoff_pct: float = 0.08, t_sum: float = 1.0, want_delta: bool = False,
                    layer_name: str = "layer", b: float = 0.1, norm_mode: Optional[str] = None):
        k = len(finetunes)
        if k < 1 or k > _lib.MAX_MODELS:
            raise ValueError(f"{k} models to merge: supported range is 1..{_lib.MAX_MODELS}")
        # one input dtype per call (smhip_layer_desc.in_dtype).  Mixed dtypes are PROMOTED to
        # fp32, never demoted: the reference upcasts every tensor to fp32 before subtracting
        # (base.py:128-131), so an fp32 base next to bf16 finetunes must keep its low bits
        dtypes = {t.dtype for t in list(finetunes) + list(bases)}
        in_dtype = next(iter(dtypes)) if len(dtypes) == 1 else torch.float32
        if in_dtype not in _DTYPE_CODE:
            in_dtype = torch.float32
        keep = []           # keep device copies alive until the call returns
        desc = LayerDesc()
        desc.k = k
        seen: Dict[int, torch.Tensor] = {}
        for i in range(k):
            ft = self._dev(finetunes[i], in_dtype)
            bkey = id(bases[i])
            if bkey not in seen:
                seen[bkey] = self._dev(bases[i], in_dtype)
            bs = seen[bkey]
            if ft.shape != base_out.shape or bs.shape != base_out.shape:
                raise ValueError(f"shape mismatch in {layer_name}: {tuple(ft.shape)} / {tuple(bs.shape)} / {tuple(base_out.shape)}")
            keep += [ft, bs]
            desc.finetune[i] = ft.data_ptr()
            desc.base[i] = bs.data_ptr()
            desc.alpha[i] = float(alphas[i])
        bo_dtype = base_out.dtype if base_out.dtype in _DTYPE_CODE else torch.float32
        bo = seen.get(id(base_out))
        if bo is None or bo.dtype != bo_dtype:
            bo = self._dev(base_out, bo_dtype)
        keep.append(bo)
        nb, r, c = _shape3d(bo)
        desc.in_dtype = _DTYPE_CODE[in_dtype]
        desc.base_out = bo.data_ptr()
        desc.base_out_dtype = _DTYPE_CODE[bo_dtype]
        desc.rows, desc.cols = r, c
        desc.target_norm_offset = float(target_norm_offset)
        desc.cull_start_pct = float(cull_start_pct)
        desc.cutoff_pct = float(cutoff_pct)
        desc.t_sum = float(t_sum)
        desc.b = float(b)
        if norm_mode is None:
            norm_mode = DEFAULT_NORM_MODE            # one default everywhere (constants.py)
        if norm_mode not in NORM_MODES:
            raise ValueError(f"norm_mode {norm_mode!r}: 'exact' or 'reference_cpu'")
        desc.norm_mode = 1 if norm_mode == "reference_cpu" else 0
        desc.batch = nb
        out = torch.empty(bo.shape, dtype=torch.bfloat16, device=self.device)
        delta = torch.empty(bo.shape, dtype=torch.float32, device=self.device) if want_delta else None
        rep = LayerReport()
        self._call(self.lib.dll.smhip_merge_layer(self.ctx.h, C.byref(desc), out.data_ptr(),
                                                  delta.data_ptr() if delta is not None else None,
                                                  C.byref(rep), self._stream()), layer_name)
        report = LayerMergeReport(
            target_norm=rep.target_norm,
            delta_norms=[rep.delta_norm[i] for i in range(k)],
            steps=[(rep.step_x[i], rep.step_y[i], _lib.BRANCH_NAMES.get(rep.step_branch[i], "?")) for i in range(rep.n_steps)],
            infos=[BlendReport.from_c(rep.step_info[i]) for i in range(rep.n_steps)],
            nan_ifft=int(rep.nan_ifft), nan_final=int(rep.nan_final), merged_delta_norm=rep.merged_delta_norm)
        if report.nan_ifft or report.nan_final:
            logger.info(f"Warning: NaN replaced by 0 in {layer_name}: {report.nan_ifft} after ifft, {report.nan_final} after add-back")
        if want_delta:
            return out, report, delta
        return out, report


_engines: Dict[str, Engine] = {}


def resolve_device(device) -> torch.device:
    """'cuda', 'cuda:1', 'cpu' (ignored with a warning: this build computes on the GPU)."""
    d = torch.device(device) if device is not None else torch.device("cuda")
    if d.type != "cuda":
        logger.warning("device=%s requested; shardmerge_amd computes on the MI355X (HIP) only - using cuda", device)
        d = torch.device("cuda")
    if d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def get_engine(device=None) -> Engine:
    """Engine for a torch device; raises when the HIP library or a GPU is missing."""
    _lib.get_lib()                                   # fail loudly before touching torch.cuda
    if not torch.cuda.is_available():
        raise RuntimeError("no ROCm GPU visible: shardmerge_amd has no CPU path")
    d = resolve_device(device)
    key = str(d)
    if key not in _engines:
        _engines[key] = Engine(device=d)
    return _engines[key]
