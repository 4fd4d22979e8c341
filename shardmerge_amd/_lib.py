"""ctypes binding of the C ABI declared in include/shardmerge_hip.h.

The product path binds ``libshardmerge_hip.so`` (built in-tree by
``shardmerge_amd/csrc/Makefile``) and nothing else: there is no CPU fallback.
If the library is missing, ``get_lib()`` raises.  ``SmhipLibrary`` takes an
explicit path only so that the test-suite can bind the CPU work-group emulator
it builds under ``tests/emul`` with the same declarations.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

MAX_MODELS = 16
MAX_PAIRS = 32

OK, ERR_HIP, ERR_SHAPE, ERR_INF_IFFT, ERR_INF_MERGED, ERR_ARG, ERR_NOMEM, ERR_NONFINITE = range(8)
BF16, F16, F32 = 0, 1, 2
BRANCH_NAMES = {0: "add", 1: "arith", 2: "slerp", 3: "carry", 4: "early_v0", 5: "linear"}


class BlendInfo(C.Structure):
    _fields_ = [
        ("cutoff_threshold", C.c_double), ("cull_threshold", C.c_double),
        ("dot", C.c_double), ("s00", C.c_double), ("s01", C.c_double), ("s11", C.c_double),
        ("n_slerp", C.c_uint64),
        ("t", C.c_double), ("cull_pct", C.c_double),
    ]


class LayerDesc(C.Structure):
    _fields_ = [
        ("k", C.c_int),
        ("finetune", C.c_void_p * MAX_MODELS),
        ("base", C.c_void_p * MAX_MODELS),
        ("alpha", C.c_double * MAX_MODELS),
        ("in_dtype", C.c_int),
        ("base_out", C.c_void_p),
        ("base_out_dtype", C.c_int),
        ("rows", C.c_int), ("cols", C.c_int),
        ("target_norm_offset", C.c_double),
        ("cull_start_pct", C.c_double),
        ("cutoff_pct", C.c_double),
        ("t_sum", C.c_double),
        ("b", C.c_double),
        ("norm_mode", C.c_int),
        ("batch", C.c_int),
    ]


class LayerReport(C.Structure):
    _fields_ = [
        ("target_norm", C.c_double),
        ("delta_norm", C.c_double * MAX_MODELS),
        ("n_steps", C.c_int),
        ("step_x", C.c_int * MAX_PAIRS), ("step_y", C.c_int * MAX_PAIRS),
        ("step_branch", C.c_int * MAX_PAIRS),
        ("step_info", BlendInfo * MAX_PAIRS),
        ("nan_ifft", C.c_uint32), ("nan_final", C.c_uint32),
        ("merged_delta_norm", C.c_double),
    ]


class SmhipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"smhip error {code}: {message}")
        self.code = code
        self.message = message


class SmhipLibrary:
    """One loaded shared object exposing the smhip_* C ABI."""

    def __init__(self, path: os.PathLike):
        self.path = str(path)
        self.dll = C.CDLL(self.path)
        d = self.dll
        P, I, D, F = C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_float)
        d.smhip_version.restype = C.c_char_p
        d.smhip_create.argtypes = [I, C.POINTER(P)]
        d.smhip_destroy.argtypes = [P]
        d.smhip_destroy.restype = None
        d.smhip_last_error.argtypes = [P]
        d.smhip_last_error.restype = C.c_char_p
        d.smhip_reserve.argtypes = [P, I, I]
        d.smhip_workspace_bytes.argtypes = [P]
        d.smhip_workspace_bytes.restype = C.c_size_t
        d.smhip_length_supported.argtypes = [I]
        d.smhip_shape_supported.argtypes = [I, I]
        d.smhip_fft_transform.argtypes = [P, P, I, I, P, P]
        d.smhip_ifft_transform.argtypes = [P, P, I, I, P, P]
        d.smhip_interpolate_fft_components.argtypes = [P, P, P, I, I, D, D, D, D, I, P, C.POINTER(BlendInfo), P]
        d.smhip_arithmetic_fft_components.argtypes = [P, P, P, I, I, D, I, I, P, P]
        d.smhip_merge_tensors_fft2_slerp.argtypes = [P, P, P, I, I, D, D, D, D, D, P, C.POINTER(D), C.POINTER(D),
                                                     C.POINTER(I), C.POINTER(BlendInfo), P]
        d.smhip_task_arithmetic_fft2.argtypes = [P, P, P, I, I, D, I, P, P]
        d.smhip_merge_layer.argtypes = [P, C.POINTER(LayerDesc), P, P, C.POINTER(LayerReport), P]
        d.smhip_addition_merge.argtypes = [P, I, C.POINTER(C.c_void_p), P, I, C.c_size_t, I, P, P]
        d.smhip_correlate_pairs.argtypes = [P, I, C.POINTER(C.c_void_p), I, C.c_size_t, C.c_size_t, C.POINTER(C.c_float), P]
        d.smhip_reference_cpu_norm.argtypes = [P, P, P, I, C.c_size_t, C.POINTER(C.c_float), P]
        d.smhip_slerp.argtypes = [P, P, P, C.c_size_t, C.c_size_t, C.c_float, P, P]
        d.smhip_exact_norm.argtypes = [P, P, I, C.c_size_t, C.POINTER(D), P]
        d.smhip_div_scalar.argtypes = [P, P, I, C.c_size_t, C.c_float, P, P]
        d.smhip_debug_option.argtypes = [P, C.c_char_p, C.c_long]
        d.smhip_debug_query.argtypes = [P, C.c_char_p, C.POINTER(C.c_long)]
        d.smhip_profile_enable.argtypes = [P, I]
        d.smhip_profile_reset.argtypes = [P]
        d.smhip_profile_count.argtypes = [P]
        d.smhip_profile_get.argtypes = [P, I, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(D)]

    def version(self) -> str:
        return self.dll.smhip_version().decode()

    def length_supported(self, n: int) -> bool:
        return self.dll.smhip_length_supported(int(n)) == OK

    def shape_supported(self, rows: int, cols: int) -> bool:
        """what merge_layer takes: one length planned, the other planned or p * M (include/shardmerge_hip.h)"""
        return self.dll.smhip_shape_supported(int(rows), int(cols)) == OK


class Context:
    """A smhip_ctx: one per device / caller thread.  Owns the device workspace."""

    def __init__(self, lib: SmhipLibrary, device: int = 0):
        self.lib = lib
        self.device = device
        h = C.c_void_p()
        rc = lib.dll.smhip_create(int(device), C.byref(h))
        if rc != OK or not h:
            raise SmhipError(rc, f"smhip_create(device={device}) failed")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.dll.smhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        if rc != OK:
            raise SmhipError(rc, self.lib.dll.smhip_last_error(self.h).decode())

    def workspace_bytes(self) -> int:
        return int(self.lib.dll.smhip_workspace_bytes(self.h))

    def debug_option(self, key: str, value: int):
        self.check(self.lib.dll.smhip_debug_option(self.h, key.encode(), int(value)))

    def debug_query(self, key: str) -> int:
        out = C.c_long(0)
        self.check(self.lib.dll.smhip_debug_query(self.h, key.encode(), C.byref(out)))
        return int(out.value)

    def profile(self, on: bool):
        self.check(self.lib.dll.smhip_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self.check(self.lib.dll.smhip_profile_reset(self.h))

    def profile_table(self):
        out = {}
        for i in range(self.lib.dll.smhip_profile_count(self.h)):
            name, n, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            self.check(self.lib.dll.smhip_profile_get(self.h, i, C.byref(name), C.byref(n), C.byref(ms)))
            out[name.value.decode()] = (int(n.value), float(ms.value))
        return out


# SHARDMERGE_HIP_LIB selects another build of the same HIP library (tuning experiments)
_HIP_LIB_PATH = Path(os.environ.get("SHARDMERGE_HIP_LIB") or Path(__file__).resolve().parent / "libshardmerge_hip.so")
_lib: Optional[SmhipLibrary] = None


def hip_library_path() -> Path:
    return _HIP_LIB_PATH


def get_lib() -> SmhipLibrary:
    """The HIP library.  Raises if it has not been built: no fallback exists."""
    global _lib
    if _lib is None:
        if not _HIP_LIB_PATH.exists():
            raise RuntimeError(
                f"{_HIP_LIB_PATH} is missing: build it with `make -C shardmerge_amd/csrc` "
                "(or __graft_entry__.build()); shardmerge_amd has no CPU fallback")
        _lib = SmhipLibrary(_HIP_LIB_PATH)
    return _lib
