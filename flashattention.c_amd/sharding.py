"""Batch*head sharding across the GPUs of one node -- no collective on the data path.

Every (batch*head) slab is independent (the reference uses ``blockIdx.x`` only to pick the slab,
/root/reference/src/flashattention.cu:144), so multi-GPU is a contiguous split of dim 0 of the (BH, N, d) tensors.
Two launch styles are supported:

* one process per GPU (``torchrun``): each rank calls ``shard_range(bh, world, rank)`` and runs ``flash.forward`` on
  its slice; bench.py does this and only uses ``torch.distributed`` for the barrier / max-over-ranks timing.
* one process, several devices: ``forward_sharded`` hands per-device pointers to ``fa_forward_sharded``.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import torch

from . import _cabi


def shard_range(bh: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slab range of `rank`; the first bh % world ranks get one extra slab."""
    if world < 1 or not (0 <= rank < world) or bh < 0:
        raise ValueError(f"bad shard request bh={bh} world={world} rank={rank}")
    base, rem = divmod(bh, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_sizes(bh: int, world: int) -> List[int]:
    return [e - b for b, e in (shard_range(bh, world, r) for r in range(world))]


def forward_sharded(qs: Sequence[torch.Tensor], ks: Sequence[torch.Tensor], vs: Sequence[torch.Tensor],
                    causal: bool = False, *, scale: float = 1.0, return_lse: bool = False, kernel="auto",
                    out_dtype=None):
    """Single-process multi-device forward: shard i lives on ``qs[i].device``; returns the per-device outputs (and, with
    ``return_lse``, the per-device (bh_i, N) log-sum-exps).  Goes through ``fa_forward_sharded_ex``: every shard gets a ``torch.empty``
    workspace of ``fa_workspace_bytes`` on its own device (nothing is allocated inside the C ABI), an explicit ``kernel`` and
    ``out_dtype=torch.float32`` for bf16 shards (the accurate path) work as in ``forward``."""
    from .flash import _kernel_id
    n_sh = len(qs)
    if not (n_sh == len(ks) == len(vs)) or n_sh < 1:
        raise ValueError("qs, ks, vs must be equally long, non-empty lists")
    n, d, dt = qs[0].shape[1], qs[0].shape[2], qs[0].dtype
    odt = dt if out_dtype is None else out_dtype
    if odt != dt and not (dt == torch.bfloat16 and odt == torch.float32):
        raise TypeError(f"out_dtype {odt} not supported for {dt} inputs")
    dtype_id = {torch.float32: _cabi.FA_DTYPE_F32, torch.bfloat16: _cabi.FA_DTYPE_BF16}[dt] if odt == dt else _cabi.FA_DTYPE_BF16_OUT_F32
    kid = _kernel_id(kernel)
    L = _cabi.lib()
    outs, lses, wss, streams, devs = [], [], [], [], []
    for q, k, v in zip(qs, ks, vs):
        if not (q.is_cuda and q.shape == k.shape == v.shape and q.dtype == k.dtype == v.dtype == dt):
            raise ValueError("every shard needs matching GPU tensors")
        if not (q.device == k.device == v.device):
            raise ValueError("q, k, v of one shard must live on the same device")
        if q.shape[1] != n or q.shape[2] != d:
            raise ValueError("all shards must share seq_len and head_dim")
        # the kernel writes a dense row-major (bh, n, d) shard whatever the strides of q are
        outs.append(torch.empty(q.shape, dtype=odt, device=q.device))
        lses.append(torch.empty(q.shape[:2], dtype=torch.float32, device=q.device) if return_lse else None)
        need = int(L.fa_workspace_bytes(q.shape[0], n, d, int(bool(causal)), dtype_id, kid)) if q.shape[0] else 0
        with torch.cuda.device(q.device):
            wss.append(torch.empty(need, dtype=torch.uint8, device=q.device) if need else None)
        devs.append(q.device.index)
        streams.append(torch.cuda.current_stream(q.device).cuda_stream)
    # contiguous copies (if any were needed) stay referenced until the launches are enqueued; torch's caching allocator keeps a
    # block alive for work already queued on the stream it was allocated on
    qc, kc, vc = ([t.contiguous() for t in ts] for ts in (qs, ks, vs))
    vp = ctypes.c_void_p
    arr = lambda ts: (vp * n_sh)(*[t.data_ptr() if (t is not None and t.numel()) else None for t in ts])  # noqa: E731
    rc = L.fa_forward_sharded_ex(
        n_sh, (ctypes.c_int32 * n_sh)(*devs), arr(qc), arr(kc), arr(vc), arr(outs), arr(lses) if return_lse else None,
        (ctypes.c_int64 * n_sh)(*[q.shape[0] for q in qs]), n, d, float(scale), int(bool(causal)), dtype_id, kid,
        arr(wss), (ctypes.c_size_t * n_sh)(*[w.numel() if w is not None else 0 for w in wss]), (vp * n_sh)(*streams))
    del qc, kc, vc
    _cabi.check(rc)
    return (outs, lses) if return_lse else outs
