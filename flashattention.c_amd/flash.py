"""Host-side mirror of the reference's operator surface.

The reference exposes exactly one call (``/root/reference/src/main.cpp:3-6``, used at
``/root/reference/bench_flashattention.py:10,70``)::

    minimal_flash = load(name='flash', sources=[...])
    out = minimal_flash.forward(q, k, v, masking)        # q, k, v: (batch*heads, seq, head_dim) on the GPU

This module provides the same call with the same positional meaning, backed by the C ABI in
``include/flashattn_amd.h`` (hand-written HIP for gfx950).  PyTorch is used for device memory and the stream
handle only.  Defaults reproduce the reference: ``scale = 1.0`` (``flashattention.cu:593,600``), fp32 in -> fp32 out.
Differences, all deliberate (DESIGN.md "boundary"): the call is asynchronous on the current stream (the reference
ends with ``cudaDeviceSynchronize``), invalid input raises ``ValueError``/``TypeError`` instead of tripping a device
``assert`` (``flashattention.cu:606``), every head dim up to 256 runs (32/64/128 on every kernel family, the other multiples of 32
on the exact fp32 kernel -- fp32 or bf16 tensors --, the rest on the rung-0 kernel; the reference needs ``#define d`` edited, ``:15``), any sequence length is exact
(the reference needs N % 32 == 0), and bf16 tensors are accepted (bf16 MFMA path).
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch

from . import _cabi

__all__ = ["forward", "forward_packed_qkv", "load", "time_forward", "last_forward_route", "workspace_bytes", "stats", "SUPPORTED_HEAD_DIMS"]

SUPPORTED_HEAD_DIMS = (32, 64, 128)   # ... by every kernel family; ``kernel="auto"`` takes any head dim up to MAX_HEAD_DIM
MAX_HEAD_DIM = 256
_DTYPES = {torch.float32: _cabi.FA_DTYPE_F32, torch.bfloat16: _cabi.FA_DTYPE_BF16}
_KERNELS = {"auto": _cabi.FA_KERNEL_AUTO, "naive": _cabi.FA_KERNEL_NAIVE, "mfma": _cabi.FA_KERNEL_MFMA,
            "exact": _cabi.FA_KERNEL_MFMA, "split": _cabi.FA_KERNEL_SPLIT, "pb2": _cabi.FA_KERNEL_PB2}


def _kernel_id(kernel: Union[str, int]) -> int:
    if isinstance(kernel, int):
        return kernel
    name, _, variant = kernel.partition(":")
    if name not in _KERNELS:
        raise ValueError(f"unknown kernel {kernel!r}; choose from {sorted(_KERNELS)}")
    return _KERNELS[name] | (int(variant) << 8 if variant else 0)


def _check_qkv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> Tuple[int, int, int]:
    for name, t in (("q", q), ("k", k), ("v", v)):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a torch.Tensor")
        if t.dim() != 3:
            raise ValueError(f"{name} must be 3-D (batch*heads, seq_len, head_dim), got shape {tuple(t.shape)}")
    if not (q.shape == k.shape == v.shape):
        raise ValueError(f"q, k, v must have identical shapes, got {tuple(q.shape)}, {tuple(k.shape)}, {tuple(v.shape)}")
    if not (q.dtype == k.dtype == v.dtype):
        raise TypeError("q, k, v must share one dtype")
    if q.dtype not in _DTYPES:
        raise TypeError(f"dtype {q.dtype} not supported (float32 or bfloat16)")
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise ValueError("q, k, v must live on a GPU (there is no CPU implementation of this operator)")
    if not (q.device == k.device == v.device):
        raise ValueError("q, k, v must be on the same device")
    bh, n, d = q.shape
    if bh < 1 or n < 1:
        raise ValueError("empty batch or sequence")
    return bh, n, d


def forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = False, *,
            scale: float = 1.0, return_lse: bool = False, kernel: Union[str, int] = "auto",
            out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None,
            workspace: Optional[torch.Tensor] = None):
    """``O = softmax(scale * q k^T [+ causal mask]) v`` per (batch*head); drop-in for ``flash.forward(q, k, v, causal)``.

    Returns a new tensor shaped like ``q`` (the reference allocates with ``torch::zeros``; here ``torch.empty`` is
    enough because every element is written).  ``return_lse=True`` additionally returns the (BH, N) fp32 row
    log-sum-exp -- the quantity the reference's unused ``O_l`` buffer was reserved for.

    Kernel choice and accuracy: ``include/flashattn_amd.h`` sections 1-2 are the contract.  fp32 tensors: ``kernel="auto"`` computes on
    the 16-bit matrix pipes with split operands (Q.K^T: two-term fp16 splits of Q' and of the keys centred on a reference key; P.V: two-term
    bf16 splits of P and of the centred values): ``|O - O64| <= max(1e-3, E_ref) + 3 * 2^-17 * max|v - vbar|`` on every input, ``E_ref`` =
    what the reference's own fp32 FMA chain leaves on that input; <= 1e-4 on unit-variance data; a workgroup whose operands leave what the
    16-bit terms hold redoes its rows in fp32 arithmetic inside the same launch.  ``"split"`` is the same without that guard, ``"exact"``
    (= ``"mfma"``) computes in fp32 arithmetic (head dims 32 .. 256 in steps of 32).  bf16 tensors: ``out_dtype=torch.float32`` stores the
    fp32 accumulator and, under ``"auto"``, selects the accurate P -- bf16 hi + bf16 lo terms in one launch (``"pb2"``: 2e-5 on
    B=2 H=8 d=64 N=8192); a bf16 output keeps the fastest kernels (bf16 P, ``"mfma"``: 1.5e-2 there, 3e-4 at 1/sqrt(d)).  Head dims
    outside 32 / 64 / 128: see the module docstring.  ``out`` must not overlap q, k or v.

    The call goes through ``fa_forward_ws``: scratch (the partials of a key-split launch), when the call can use any, is a ``torch.empty``
    byte tensor from torch's caching allocator on the current stream -- the C ABI itself allocates nothing, which also makes every kernel
    family legal under ``torch.cuda.graph`` capture.  ``workspace`` may pass a preallocated ``torch.uint8`` tensor of at least
    ``workspace_bytes(...)`` bytes instead.
    """
    bh, n, d = _check_qkv(q, k, v)
    kid = _kernel_id(kernel)
    if d > MAX_HEAD_DIM:
        raise ValueError(f"head_dim {d} not supported (1 .. {MAX_HEAD_DIM})")
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else q.dtype
    dt = _DTYPES[q.dtype]
    if out_dtype != q.dtype:
        if not (q.dtype == torch.bfloat16 and out_dtype == torch.float32):
            raise TypeError(f"out_dtype {out_dtype} not supported for {q.dtype} inputs")
        dt = _cabi.FA_DTYPE_BF16_OUT_F32
    if out is None:
        out = torch.empty(q.shape, dtype=out_dtype, device=q.device)
    elif out.shape != q.shape or out.dtype != out_dtype or out.device != q.device or not out.is_contiguous():
        raise ValueError("out must be a contiguous tensor shaped like q with dtype out_dtype")
    lse = torch.empty((bh, n), dtype=torch.float32, device=q.device) if return_lse else None
    L = _cabi.lib()
    with torch.cuda.device(q.device):
        need = int(L.fa_workspace_bytes(bh, n, d, int(bool(causal)), dt, kid))
        if workspace is None:
            # 256-byte aligned by the caching allocator (its blocks are 512-byte multiples)
            workspace = torch.empty(need, dtype=torch.uint8, device=q.device) if need else None
        elif workspace.dtype != torch.uint8 or workspace.device != q.device or not workspace.is_contiguous() or workspace.numel() < need:
            raise ValueError(f"workspace must be a contiguous torch.uint8 tensor of at least {need} bytes on {q.device}")
        stream = torch.cuda.current_stream().cuda_stream
        rc = L.fa_forward_ws(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                             lse.data_ptr() if lse is not None else None,
                             bh, n, d, float(scale), int(bool(causal)), dt, kid,
                             workspace.data_ptr() if workspace is not None else None,
                             workspace.numel() if workspace is not None else 0, ctypes.c_void_p(stream))
    _cabi.check(rc)
    return (out, lse) if return_lse else out


def workspace_bytes(bh: int, n: int, d: int, causal: bool = False, *, dtype: torch.dtype = torch.float32,
                    out_dtype: Optional[torch.dtype] = None, kernel: Union[str, int] = "auto") -> int:
    """Bytes of scratch ``forward`` needs for this call (``fa_workspace_bytes``); 0 for most shapes."""
    dt = _DTYPES[dtype]
    if out_dtype is not None and out_dtype != dtype:
        dt = _cabi.FA_DTYPE_BF16_OUT_F32
    return int(_cabi.lib().fa_workspace_bytes(int(bh), int(n), int(d), int(bool(causal)), dt, _kernel_id(kernel)))


def forward_packed_qkv(inp: torch.Tensor, n_head: int) -> torch.Tensor:
    """llm.c layout: ``inp`` (B, T, 3C) fp32 -> (B, T, C); causal, scale 1/sqrt(C/n_head).

    Replaces ``attention_forward6`` (/root/reference/src/llm.c/attention_forward.cu:1106-1179) without its
    permute / unpermute kernels or temporaries.
    """
    if inp.dim() != 3 or inp.shape[-1] % 3 != 0:
        raise ValueError(f"inp must be (B, T, 3C), got {tuple(inp.shape)}")
    if inp.dtype != torch.float32 or not inp.is_cuda:
        raise TypeError("inp must be a float32 GPU tensor")
    b, t, c3 = inp.shape
    c = c3 // 3
    if c % n_head != 0:
        raise ValueError("C must be divisible by n_head")
    inp = inp.contiguous()
    out = torch.empty((b, t, c), dtype=torch.float32, device=inp.device)
    with torch.cuda.device(inp.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = _cabi.lib().fa_forward_packed_qkv(inp.data_ptr(), out.data_ptr(), b, t, c, int(n_head), ctypes.c_void_p(stream))
    _cabi.check(rc)
    return out


def time_forward(q, k, v, causal: bool = False, *, scale: float = 1.0, kernel: Union[str, int] = "auto",
                 warmup: int = 3, iters: int = 20, out: Optional[torch.Tensor] = None, graph: bool = False) -> float:
    """Mean milliseconds per forward, HIP events recorded on the launch stream inside the C ABI (fa_time_forward).
    ``graph=True``: the ``iters`` launches are captured into one hipGraph and one replay is timed (fa_time_forward_graph)."""
    bh, n, d = _check_qkv(q, k, v)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    if out is None:
        out = torch.empty_like(q)
    dt = _DTYPES[q.dtype]
    if out.dtype != q.dtype:
        if not (q.dtype == torch.bfloat16 and out.dtype == torch.float32):
            raise TypeError(f"out dtype {out.dtype} not supported for {q.dtype} inputs")
        dt = _cabi.FA_DTYPE_BF16_OUT_F32
    ms = ctypes.c_float(0.0)
    with torch.cuda.device(q.device):
        if graph:
            torch.cuda.current_stream().synchronize()
            rc = _cabi.lib().fa_time_forward_graph(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), bh, n, d, float(scale),
                                                   int(bool(causal)), dt, _kernel_id(kernel), int(warmup), int(iters),
                                                   ctypes.byref(ms))
            _cabi.check(rc)
            return float(ms.value)
        stream = torch.cuda.current_stream().cuda_stream
        rc = _cabi.lib().fa_time_forward(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), bh, n, d, float(scale),
                                         int(bool(causal)), dt, _kernel_id(kernel), ctypes.c_void_p(stream),
                                         int(warmup), int(iters), ctypes.byref(ms))
    _cabi.check(rc)
    return float(ms.value)


def last_forward_route(stream: Optional[torch.cuda.Stream] = None) -> int:
    """Which arithmetic produced this thread's most recent forward (blocking; diagnostics): 0 = nothing to report (every bf16 path,
    explicit kernels, other head dims, forwards enqueued under graph capture), 1 = fp32 tensors under ``"auto"``: split products throughout,
    2 = the range guard fired (operands outside what fp16 terms hold, or a NaN) and at least one workgroup redid its rows in exact fp32
    arithmetic (inside the same launch)."""
    r = ctypes.c_int32(0)
    s = (stream or torch.cuda.current_stream()).cuda_stream
    _cabi.check(_cabi.lib().fa_last_forward_route(ctypes.c_void_p(s), ctypes.byref(r)))
    return int(r.value)


def stats() -> dict:
    """Host counters (``fa_get_stats``: forwards, re-plans without scratch) merged with the two counters the kernels bump on their slow
    paths (``fa_read_device_counters``, BLOCKING): ``tiles_redone`` (optimistic attempt failed, tile recomputed: ~2x) and
    ``workgroups_fp32`` (fp32 ``"auto"`` workgroups redone in fp32 arithmetic: ~3x)."""
    st = _cabi.FaStats()
    _cabi.check(_cabi.lib().fa_get_stats(ctypes.byref(st), ctypes.sizeof(st)))
    out = {n: int(getattr(st, n)) for n, _ in _cabi.FaStats._fields_}
    t, w = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _cabi.check(_cabi.lib().fa_read_device_counters(ctypes.byref(t), ctypes.byref(w)))
    out["tiles_redone"], out["workgroups_fp32"] = int(t.value), int(w.value)
    return out


def load(name: str = "flash", sources=None, extra_cuda_cflags=None, **_ignored):
    """Stand-in for ``torch.utils.cpp_extension.load(name='flash', sources=[...])`` as called at
    /root/reference/bench_flashattention.py:10: returns an object whose ``.forward(q, k, v, causal)`` is this operator.
    ``sources`` / flags are accepted and ignored -- the library is prebuilt by ``build.py``."""
    _cabi.lib()  # fail loudly now if the extension is missing
    return SimpleNamespace(forward=forward, forward_packed_qkv=forward_packed_qkv, __name__=name)
