"""Round 5, experiment 4 (GPU): causal key-share launches (small BH, long rows) in the three families that have them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import flashattention_c_amd as fa
dev = torch.device("cuda", 0)
def t(q, k, v, kernel, out=None, iters=30):
    fa.time_forward(q, k, v, True, kernel=kernel, warmup=20, iters=5, out=out)
    return min(fa.time_forward(q, k, v, True, kernel=kernel, warmup=3, iters=iters, out=out) for _ in range(3))
for d in (64, 128):
    for (bh, n) in ((1, 8192), (2, 8192), (4, 8192), (8, 8192), (1, 16384), (3, 5000)):
        q, k, v = (torch.randn(bh, n, d, device=dev) for _ in range(3))
        ref = fa.forward(q, k, v, True, kernel="naive")
        qb, kb, vb = (x.bfloat16() for x in (q, k, v))
        refb = fa.forward(qb.float(), kb.float(), vb.float(), True, kernel="naive")
        o32 = torch.empty_like(q)
        line = f"d={d:3d} bh={bh} n={n:5d} causal:"
        for name, tens, kern, out, r in (("bf16", (qb, kb, vb), "auto", None, refb), ("bf16->f32 (pb2)", (qb, kb, vb), "auto", o32, refb), ("fp32 auto", (q, k, v), "auto", None, ref), ("fp32 exact", (q, k, v), "exact", None, ref)):
            o = fa.forward(*tens, True, kernel=kern, out_dtype=(torch.float32 if out is not None else None))
            err = float((o.float() - r).abs().max())
            ms = t(*tens, kern, out=out)
            line += f"  {name} {ms:.4f} ms (err {err:.1e})"
        print(line, flush=True)
