"""Round 5, experiment 2 (GPU): the exact fp32 kernel -- paired causal tiles (mfma:2) against one tile per workgroup (mfma:1), key shares
for idle grids (mfma = the plan's choice) against the unsplit launch (mfma:1); correctness of each against the naive kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import flashattention_c_amd as fa
dev = torch.device("cuda", 0)
def t(q, k, v, causal, kernel, iters=10):
    fa.time_forward(q, k, v, causal, kernel=kernel, warmup=5, iters=3)
    return min(fa.time_forward(q, k, v, causal, kernel=kernel, warmup=2, iters=iters) for _ in range(3))
for (bh, n, d) in ((16, 8192, 64), (128, 1024, 64), (16, 8192, 128), (16, 8192, 32), (8, 8192, 64), (32, 4096, 64), (64, 2048, 64), (12, 8192, 64), (4, 8192, 64), (2, 8192, 64), (1, 8192, 64), (1, 16384, 64), (1, 4096, 128), (3, 5000, 64), (40, 1500, 64)):
    q, k, v = (torch.randn(bh, n, d, device=dev) for _ in range(3))
    for causal in (False, True):
        ref, lref = fa.forward(q, k, v, causal, kernel="naive", return_lse=True)
        line = f"bh={bh:3d} n={n:5d} d={d:3d} causal={int(causal)}:"
        for kern in (["mfma", "mfma:1"] + (["mfma:2"] if causal else [])):
            o, l = fa.forward(q, k, v, causal, kernel=kern, return_lse=True)
            err = float((o - ref).abs().max()); lerr = float((l - lref).abs().max())
            ms = t(q, k, v, causal, kern)
            fl = (2 if causal else 4) * bh * n * n * d
            line += f"  {kern} {ms:.4f} ms ({fl / ms / 1e9 / 157.3:.3f}) err {err:.1e}/{lerr:.1e} ws {fa.workspace_bytes(bh, n, d, causal, kernel=kern)}"
        print(line, flush=True)
