import torch, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import flashattention_c_amd as fa
dev = torch.device('cuda:0')
for (bh, n, d) in ((16, 8192, 64), (128, 1024, 64)):
    q, k, v = (torch.randn(bh, n, d, device=dev) for _ in range(3))
    ones = torch.ones_like(v); zeros = torch.zeros_like(v)
    for name, vv in (("N(0,1)", v), ("ones", ones), ("zeros", zeros), ("tiny 2^-60", v * 2.0 ** -60)):
        ms = [fa.time_forward(q, k, vv, c, warmup=10, iters=20) for c in (False, True)]
        print(f"fp32 {bh}x{n}x{d} V = {name:10s}: {ms[0]:.4f} ms   causal {ms[1]:.4f} ms")
for (bh, n, d) in ((16, 8192, 64), (128, 8192, 64), (16, 8192, 128)):
    q, k, v = (torch.randn(bh, n, d, device=dev, dtype=torch.bfloat16) for _ in range(3))
    for name, vv in (("N(0,1)", v), ("zeros", torch.zeros_like(v)), ("tiny 2^-60", (v.float() * 2.0 ** -60).to(torch.bfloat16))):
        ms = [fa.time_forward(q, k, vv, c, warmup=10, iters=20) for c in (False, True)]
        print(f"bf16 {bh}x{n}x{d} V = {name:10s}: {ms[0]:.4f} ms   causal {ms[1]:.4f} ms")
