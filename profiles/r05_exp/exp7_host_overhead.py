import time, torch, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import flashattention_c_amd as fa
dev = torch.device('cuda:0')
for (bh, n, d, dt) in ((16, 1024, 32, torch.float32), (16, 1024, 32, torch.bfloat16), (128, 1024, 64, torch.float32)):
    q, k, v = (torch.randn(bh, n, d, device=dev, dtype=dt) for _ in range(3))
    out = torch.empty_like(q)
    for _ in range(50): fa.forward(q, k, v, False, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 2000
    for _ in range(N): fa.forward(q, k, v, False, out=out)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    kms = fa.time_forward(q, k, v, False, warmup=20, iters=200, out=out)
    t0 = time.perf_counter()
    for _ in range(N): fa.forward(q, k, v, False)
    torch.cuda.synchronize()
    t_alloc = time.perf_counter() - t0
    print(f"bh={bh} n={n} d={d} {dt}: python issue {t_issue / N * 1e6:.1f} us/call, wall {t_all / N * 1e6:.1f} us/call, kernel {kms * 1e3:.1f} us; without out= {t_alloc / N * 1e6:.1f} us/call")

# MI355X box, round 5 (final binary):
#   bh=16 n=1024 d=32 fp32:  python issue 11.8 us/call, wall 24.7 us/call, kernel 24.5 us
#   bh=16 n=1024 d=32 bf16:  python issue 10.7 us/call, wall 12.0 us/call, kernel 12.0 us
#   bh=128 n=1024 d=64 fp32: python issue 22.6 us/call (queue full), wall 103.5 us/call, kernel 103.0 us
# -> the ctypes mirror enqueues a forward in ~11 us: the smallest BASELINE config (c1, 12 us in bf16) still runs kernel-bound.
