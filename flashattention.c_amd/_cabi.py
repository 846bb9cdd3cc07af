"""ctypes view of include/flashattn_amd.h (the C ABI is the product boundary; this file only binds it)."""
from __future__ import annotations

import ctypes
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libflashattn_amd.so")

FA_OK = 0
FA_DTYPE_F32, FA_DTYPE_BF16, FA_DTYPE_BF16_OUT_F32 = 0, 1, 2
FA_KERNEL_AUTO, FA_KERNEL_NAIVE, FA_KERNEL_MFMA, FA_KERNEL_SPLIT, FA_KERNEL_PB2 = 0, 1, 2, 3, 6   # (4, 5: retired fp16-P kernels)

# every symbol include/flashattn_amd.h declares
EXPORTED_SYMBOLS = (
    "fa_forward", "fa_forward_ex", "fa_workspace_bytes", "fa_forward_ws", "fa_forward_sharded", "fa_forward_sharded_ex", "fa_forward_packed_qkv", "fa_time_forward", "fa_time_forward_graph",
    "fa_last_forward_route", "fa_get_stats", "fa_read_device_counters", "fa_last_error", "fa_device_count", "fa_version", "fa_kernel_name", "fa_kernel_name_for",
)


class FaStats(ctypes.Structure):
    """struct fa_stats of include/flashattn_amd.h (ABI 6: carries its own size, fields are only ever appended)"""
    _fields_ = [(n, ctypes.c_uint64) for n in ("struct_bytes", "forwards", "scratch_replans")]


class ExtensionMissing(RuntimeError):
    """libflashattn_amd.so is not built -- there is deliberately NO fallback path."""


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissing(
            f"{LIB_PATH} not found. Build it with `python flashattention.c_amd/build.py` "
            "(or __graft_entry__.build()). There is no CPU/PyTorch fallback for this operator.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    L.fa_forward.argtypes = [vp, vp, vp, vp, i64, i64, i32, f32, i32, i32, vp]
    L.fa_forward.restype = ctypes.c_int
    L.fa_forward_ex.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, f32, i32, i32, i32, vp]
    L.fa_forward_ex.restype = ctypes.c_int
    L.fa_workspace_bytes.argtypes = [i64, i64, i32, i32, i32, i32]
    L.fa_workspace_bytes.restype = ctypes.c_size_t
    L.fa_forward_ws.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, f32, i32, i32, i32, vp, ctypes.c_size_t, vp]
    L.fa_forward_ws.restype = ctypes.c_int
    L.fa_forward_sharded.argtypes = [i32, ctypes.POINTER(i32), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                     ctypes.POINTER(vp), ctypes.POINTER(i64), i64, i32, f32, i32, i32, ctypes.POINTER(vp)]
    L.fa_forward_sharded.restype = ctypes.c_int
    L.fa_forward_sharded_ex.argtypes = [i32, ctypes.POINTER(i32), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                        ctypes.POINTER(vp), ctypes.POINTER(i64), i64, i32, f32, i32, i32, i32, ctypes.POINTER(vp),
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(vp)]
    L.fa_forward_sharded_ex.restype = ctypes.c_int
    L.fa_forward_packed_qkv.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.fa_forward_packed_qkv.restype = ctypes.c_int
    L.fa_time_forward.argtypes = [vp, vp, vp, vp, i64, i64, i32, f32, i32, i32, i32, vp, i32, i32, ctypes.POINTER(f32)]
    L.fa_time_forward.restype = ctypes.c_int
    L.fa_time_forward_graph.argtypes = [vp, vp, vp, vp, i64, i64, i32, f32, i32, i32, i32, i32, i32, ctypes.POINTER(f32)]
    L.fa_time_forward_graph.restype = ctypes.c_int
    L.fa_last_forward_route.argtypes = [vp, ctypes.POINTER(i32)]
    L.fa_last_forward_route.restype = ctypes.c_int
    L.fa_get_stats.argtypes = [ctypes.POINTER(FaStats), ctypes.c_size_t]
    L.fa_get_stats.restype = ctypes.c_int
    L.fa_read_device_counters.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.fa_read_device_counters.restype = ctypes.c_int
    L.fa_last_error.argtypes = []
    L.fa_last_error.restype = ctypes.c_char_p
    L.fa_device_count.argtypes = []
    L.fa_device_count.restype = ctypes.c_int
    L.fa_version.argtypes = []
    L.fa_version.restype = ctypes.c_char_p
    L.fa_kernel_name.argtypes = [i32, i32, i32]
    L.fa_kernel_name.restype = ctypes.c_char_p
    L.fa_kernel_name_for.argtypes = [i32, i32, i32, i64, i64]
    L.fa_kernel_name_for.restype = ctypes.c_char_p
    _lib = L
    return L


class FlashAttnError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"flashattn_amd error {code}: {message}")
        self.code = code


def check(code: int) -> None:
    if code != FA_OK:
        raise FlashAttnError(code, lib().fa_last_error().decode("utf-8", "replace"))
