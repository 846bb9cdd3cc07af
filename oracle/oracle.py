"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python face of the CPU oracle for the flash-attention forward path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module;
``flashattention.c_amd`` (the product) never does.

Two independent restatements live here:

* ``attention_numpy``      numpy fp64, the three-line formula of the reference's own Python oracle
                           (``/root/reference/bench_flashattention.py:36-48``): S = q k^T (* scale),
                           softmax over keys (causal: keys > query row masked to -inf), O = A v.
* ``liboracle.so``         plain-C restatement (``oracle/attention_oracle.c``), OpenMP, used where
                           numpy's (BH, N, N) score matrix would not fit or would be slow.

Parity pinning: ``oracle/make_golden.py`` executed the reference's Python oracle functions in the
build container and committed their outputs under ``tests/golden/``; ``tests/test_oracle.py``
holds both restatements to those vectors, and to ``oracle/_ref`` (the reference's C CPU
attention compiled from the reference tree) for the packed-QKV layout.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_llmc_attention_cpu.so")
_lib: Optional[ctypes.CDLL] = None
_ref: Optional[ctypes.CDLL] = None


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so (and oracle/_ref when the reference tree is present)."""
    src = os.path.join(_HERE, "attention_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF_PATH)):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        dp = ctypes.POINTER(ctypes.c_double)
        i64 = ctypes.c_int64
        L.oracle_attention_f64.argtypes = [fp, fp, fp, dp, dp, i64, i64, i64, ctypes.c_double, ctypes.c_int]
        L.oracle_attention_f64.restype = None
        L.oracle_attention_f32.argtypes = [fp, fp, fp, fp, fp, i64, i64, i64, ctypes.c_float, ctypes.c_int]
        L.oracle_attention_f32.restype = None
        L.oracle_flash_tiled_f32.argtypes = [fp, fp, fp, fp, i64, i64, i64, ctypes.c_float, ctypes.c_int]
        L.oracle_flash_tiled_f32.restype = None
        L.oracle_attention_packed_f32.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_attention_packed_f32.restype = None
        L.oracle_num_threads.argtypes = []
        L.oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def have_reference_build() -> bool:
    return os.path.exists(_REF_PATH)


def ref_lib() -> ctypes.CDLL:
    """The reference's own attention_forward_cpu, compiled into oracle/_ref (see oracle/Makefile)."""
    global _ref
    if _ref is None:
        L = ctypes.CDLL(_REF_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        L.attention_forward_cpu.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.attention_forward_cpu.restype = None
        _ref = L
    return _ref


def _f32c(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def num_threads() -> int:
    return int(lib().oracle_num_threads())


# ------------------------------------------------------------------------------------------------
# restatement 1: numpy fp64 (small shapes only -- materialises the (BH, N, N) score matrix)
# ------------------------------------------------------------------------------------------------
def attention_numpy(q, k, v, causal: bool = False, scale: float = 1.0, return_lse: bool = False):
    """bench_flashattention.py:36-40 (unmasked) / :42-48 (causal) in numpy fp64."""
    q64, k64, v64 = (np.asarray(t, dtype=np.float64) for t in (q, k, v))
    s = np.einsum("bnd,bmd->bnm", q64, k64) * float(scale)
    if causal:
        n = s.shape[-1]
        s = np.where(np.tril(np.ones((n, n), dtype=bool))[None], s, -np.inf)
    m = s.max(axis=-1, keepdims=True)
    p = np.exp(s - m)
    l = p.sum(axis=-1, keepdims=True)
    o = np.einsum("bnm,bmd->bnd", p / l, v64)
    if return_lse:
        return o, (m + np.log(l))[..., 0]
    return o


# ------------------------------------------------------------------------------------------------
# restatement 2: C (oracle/attention_oracle.c)
# ------------------------------------------------------------------------------------------------
def attention_f64(q, k, v, causal: bool = False, scale: float = 1.0, return_lse: bool = False):
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    bh, n, d = q.shape
    o = np.empty((bh, n, d), dtype=np.float64)
    lse = np.empty((bh, n), dtype=np.float64)
    lib().oracle_attention_f64(_ptr(q, ctypes.c_float), _ptr(k, ctypes.c_float), _ptr(v, ctypes.c_float),
                               _ptr(o, ctypes.c_double), _ptr(lse, ctypes.c_double),
                               bh, n, d, float(scale), int(bool(causal)))
    return (o, lse) if return_lse else o


def attention_f32(q, k, v, causal: bool = False, scale: float = 1.0) -> np.ndarray:
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    bh, n, d = q.shape
    o = np.empty((bh, n, d), dtype=np.float32)
    lib().oracle_attention_f32(_ptr(q, ctypes.c_float), _ptr(k, ctypes.c_float), _ptr(v, ctypes.c_float),
                               _ptr(o, ctypes.c_float), None, bh, n, d, float(scale), int(bool(causal)))
    return o


def flash_tiled_f32(q, k, v, causal: bool = False, scale: float = 1.0) -> np.ndarray:
    """The CUDA kernel's tile recurrence (src/flashattention.cu:214-354) in fp32 on the CPU."""
    q, k, v = _f32c(q), _f32c(k), _f32c(v)
    bh, n, d = q.shape
    o = np.empty((bh, n, d), dtype=np.float32)
    lib().oracle_flash_tiled_f32(_ptr(q, ctypes.c_float), _ptr(k, ctypes.c_float), _ptr(v, ctypes.c_float),
                                 _ptr(o, ctypes.c_float), bh, n, d, float(scale), int(bool(causal)))
    return o


def attention_packed_f32(inp, n_head: int) -> np.ndarray:
    """llm.c layout: inp (B, T, 3C) -> out (B, T, C); causal; scale 1/sqrt(C/NH)."""
    inp = _f32c(inp)
    b, t, c3 = inp.shape
    c = c3 // 3
    out = np.empty((b, t, c), dtype=np.float32)
    lib().oracle_attention_packed_f32(_ptr(inp, ctypes.c_float), _ptr(out, ctypes.c_float), b, t, c, int(n_head))
    return out


def reference_attention_packed_f32(inp, n_head: int) -> np.ndarray:
    """The reference's attention_forward_cpu itself (oracle/_ref); build-container / shipped .so only."""
    inp = _f32c(inp)
    b, t, c3 = inp.shape
    c = c3 // 3
    out = np.empty((b, t, c), dtype=np.float32)
    preatt = np.empty((b, n_head, t, t), dtype=np.float32)
    att = np.empty((b, n_head, t, t), dtype=np.float32)
    fp = ctypes.c_float
    ref_lib().attention_forward_cpu(_ptr(out, fp), _ptr(preatt, fp), _ptr(att, fp), _ptr(inp, fp), b, t, c, int(n_head))
    return out


# ------------------------------------------------------------------------------------------------
# layout + dtype helpers shared by the tests
# ------------------------------------------------------------------------------------------------
def split_packed_qkv(inp: np.ndarray, n_head: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(B, T, 3C) -> three (B*NH, T, hs) tensors; restates permute_kernel's index map
    (src/llm.c/attention_forward.cu:519-547)."""
    b, t, c3 = inp.shape
    c = c3 // 3
    hs = c // n_head
    x = inp.reshape(b, t, 3, n_head, hs).transpose(2, 0, 3, 1, 4)  # (3, B, NH, T, hs)
    q, k, v = (np.ascontiguousarray(x[i]).reshape(b * n_head, t, hs) for i in range(3))
    return q, k, v


def merge_heads(o: np.ndarray, batch: int, n_head: int) -> np.ndarray:
    """(B*NH, T, hs) -> (B, T, C); restates unpermute_kernel (src/llm.c/attention_forward.cu:549-565)."""
    bh, t, hs = o.shape
    return np.ascontiguousarray(o.reshape(batch, n_head, t, hs).transpose(0, 2, 1, 3)).reshape(batch, t, n_head * hs)


def round_to_bf16(a: np.ndarray) -> np.ndarray:
    """fp32 -> nearest-even bf16 -> fp32 (values representable in bf16)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return rounded.astype(np.uint32).view(np.float32).reshape(a.shape)


def bf16_bits(a: np.ndarray) -> np.ndarray:
    """fp32 (bf16-representable or not) -> uint16 bf16 bit patterns, round-to-nearest-even."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).reshape(a.shape)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(b.shape)
