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
                    causal: bool = False, *, scale: float = 1.0) -> List[torch.Tensor]:
    """Single-process multi-device forward: shard i lives on ``qs[i].device``; returns the per-device outputs."""
    n_sh = len(qs)
    if not (n_sh == len(ks) == len(vs)) or n_sh < 1:
        raise ValueError("qs, ks, vs must be equally long, non-empty lists")
    n, d, dt = qs[0].shape[1], qs[0].shape[2], qs[0].dtype
    outs, streams, devs = [], [], []
    for q, k, v in zip(qs, ks, vs):
        if not (q.is_cuda and q.shape == k.shape == v.shape and q.dtype == k.dtype == v.dtype == dt):
            raise ValueError("every shard needs matching GPU tensors")
        if not (q.device == k.device == v.device):
            raise ValueError("q, k, v of one shard must live on the same device")
        if q.shape[1] != n or q.shape[2] != d:
            raise ValueError("all shards must share seq_len and head_dim")
        # the kernel writes a dense row-major (bh, n, d) shard whatever the strides of q are
        outs.append(torch.empty(q.shape, dtype=q.dtype, device=q.device))
        devs.append(q.device.index)
        streams.append(torch.cuda.current_stream(q.device).cuda_stream)
    # contiguous copies (if any were needed) stay referenced until the launches are enqueued; torch's caching allocator keeps a
    # block alive for work already queued on the stream it was allocated on
    qc, kc, vc = ([t.contiguous() for t in ts] for ts in (qs, ks, vs))
    vp = ctypes.c_void_p
    arr = lambda ts: (vp * n_sh)(*[t.data_ptr() if t.shape[0] else None for t in ts])  # noqa: E731
    rc = _cabi.lib().fa_forward_sharded(
        n_sh, (ctypes.c_int32 * n_sh)(*devs), arr(qc), arr(kc), arr(vc), arr(outs), (ctypes.c_int64 * n_sh)(*[q.shape[0] for q in qs]),
        n, d, float(scale), int(bool(causal)), {torch.float32: 0, torch.bfloat16: 1}[dt], (vp * n_sh)(*streams))
    del qc, kc, vc
    _cabi.check(rc)
    return outs
