"""flashattention.c_amd -- MI355X-native fused flash-attention forward behind the reference's ``forward(q, k, v, causal)``.

Layout: ``csrc/`` hand-written HIP kernels + the C ABI (``include/flashattn_amd.h``); ``flash.py`` the host-side mirror
of the reference's Python surface; ``sharding.py`` batch*head sharding; ``build.py`` the hipcc build.
Import name: ``flashattention_c_amd`` (the directory name contains a dot; ``flashattention_c_amd.py`` at the repo root
aliases it).
"""
from . import _cabi  # noqa: F401
from .flash import SUPPORTED_HEAD_DIMS, forward, forward_packed_qkv, last_forward_route, load, stats, time_forward, workspace_bytes  # noqa: F401
from .sharding import forward_sharded, shard_range, shard_sizes  # noqa: F401

__all__ = ["forward", "forward_packed_qkv", "load", "time_forward", "last_forward_route", "forward_sharded", "shard_range", "shard_sizes", "workspace_bytes",
           "stats", "SUPPORTED_HEAD_DIMS"]
