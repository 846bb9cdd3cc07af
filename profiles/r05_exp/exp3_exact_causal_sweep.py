"""Round 5, experiment 3 (GPU): causal launches of the exact fp32 kernel, one tile per workgroup (mfma:1) against paired tiles (mfma:2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import flashattention_c_amd as fa
dev = torch.device("cuda", 0)
def t(q, k, v, kernel, iters=6):
    fa.time_forward(q, k, v, True, kernel=kernel, warmup=3, iters=2)
    return min(fa.time_forward(q, k, v, True, kernel=kernel, warmup=1, iters=iters) for _ in range(3))
for d in (64, 128, 32):
    for n in (512, 1024, 1500, 2048, 3000, 4096, 8192):
        for bh in (4, 8, 12, 16, 24, 32, 40, 64, 96, 128, 256):
            if bh * n > 1 << 21 or bh * n < 1 << 14: continue
            q, k, v = (torch.randn(bh, n, d, device=dev) for _ in range(3))
            a, b = t(q, k, v, "mfma:1"), t(q, k, v, "mfma:2")
            tiles = bh * ((n + 127) // 128); pairs = bh * (((n + 127) // 128 + 1) // 2)
            print(f"d={d:3d} n={n:5d} bh={bh:3d} tiles={tiles:5d} pairs={pairs:5d}  one {a:.4f}  paired {b:.4f}  paired/one {b / a:.3f}", flush=True)
