#!/usr/bin/env python3
"""Counterpart of the reference's bench harness (/root/reference/bench_flashattention.py:1-80) for MI355X.

Same workflow, same CLI (``--batch_size --seq_len --masking``), same inputs (``randn(batch*n_head, seq_len, head_dim)``,
n_head = 8, head_dim = 64 by default), same verdict line.  What the reference does with
``torch.autograd.profiler.profile(use_cuda=True)`` on one un-warmed call (``:62-72``) is done here with HIP events on the
launch stream over warmed, repeated calls, and the comparison column "PyTorch" is both the reference's manual
``matmul -> softmax -> matmul`` (``:36-48``) and PyTorch-ROCm's fused SDPA, on the GPU.

    python flashattention.c_amd/harness/bench_flashattention.py --batch_size 2 --seq_len 8192 [--masking] [--dtype bf16]
    python flashattention.c_amd/harness/bench_flashattention.py --device cpu --batch_size 2 --seq_len 1024 --head_dim 32

``--device cpu`` is config c1 of BASELINE.json (plumbing without a GPU): it only times PyTorch's CPU SDPA and the manual
oracle -- the flash operator itself has no CPU implementation and is skipped with a message.
The correctness line uses a STATED tolerance: 1e-3 for fp32 (the reference accepts 1e-1, ``:74``), per-path for bf16.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

import torch
from torch.nn import functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def manual_attention(q, k, v, masking: bool, scale: float):
    """The reference's oracle: softmax(q k^T [* scale]) v, causal via masked_fill(-inf) (bench_flashattention.py:36-48)."""
    s = torch.matmul(q, k.transpose(-2, -1)) * scale
    if masking:
        n = s.shape[-1]
        s = s.masked_fill(~torch.ones(n, n, dtype=torch.bool, device=s.device).tril(), float("-inf"))
    return torch.matmul(F.softmax(s, dim=-1), v)


def time_gpu(fn, warmup: int, iters: int) -> float:
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def time_cpu(fn, iters: int = 3) -> float:
    fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=1, help="Batch size")
    ap.add_argument("--seq_len", type=int, default=8192, help="Sequence length")
    ap.add_argument("--masking", action="store_true", help="Causal masking (the reference's type=bool flag is always-true when given)")
    ap.add_argument("--n_head", type=int, default=8)
    ap.add_argument("--head_dim", type=int, default=64, choices=(32, 64, 128))
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    ap.add_argument("--scale", type=float, default=1.0, help="softmax scale; the reference uses 1.0 (1/sqrt(d) commented out)")
    ap.add_argument("--device", choices=("cuda", "cpu"), default="cuda")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--print_tensors", action="store_true", help="print the result and the oracle tensors like the reference script does (:76-77)")
    args = ap.parse_args()
    print(args)
    bh, n, d = args.batch_size * args.n_head, args.seq_len, args.head_dim
    print(f"Using {args.batch_size} batch size, {args.n_head} heads, {n} sequence length, {d} head embedding size, "
          f"{'with' if args.masking else 'without'} causal masking, dtype {args.dtype}, scale {args.scale:g}")
    flop = (2.0 if args.masking else 4.0) * bh * n * n * d

    torch.manual_seed(args.seed)
    q, k, v = (torch.randn(bh, n, d) for _ in range(3))

    def row(name, ms):
        print(f"  {name:<44s} {ms:10.3f} ms   {flop / ms / 1e9:10.2f} TFLOP/s")

    if args.device == "cpu":
        print(f"=== CPU baseline: {torch.get_num_threads()} threads, {os.cpu_count()} host CPUs ===")
        row("PyTorch CPU SDPA (fp32)", time_cpu(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=args.masking, scale=args.scale)))
        if bh * n * n * 4 < 8e9:
            row("manual matmul->softmax->matmul (fp32)", time_cpu(lambda: manual_attention(q, k, v, args.masking, args.scale)))
            ok = torch.allclose(manual_attention(q, k, v, args.masking, args.scale),
                                F.scaled_dot_product_attention(q, k, v, is_causal=args.masking, scale=args.scale), rtol=0, atol=1e-3)
            print("[Correctness] oracle vs SDPA sanity check: " + ("PASSED" if ok else "FAILED"))
        print("flash operator: no CPU implementation (GPU-only by design) -- skipped")
        return 0

    if not torch.cuda.is_available():
        raise SystemExit("no GPU visible; use --device cpu for the plumbing run")
    import flashattention_c_amd as fa
    minimal_flash = fa.load(name="flash", sources=["src/main.cpp", "src/flashattention.cu"], extra_cuda_cflags=["-O3"])

    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    qd, kd, vd = (t.to(tdt).cuda() for t in (q, k, v))

    print("=== profiling manual attention (the reference's 'Pytorch' column) ===")
    manual_result = None
    if bh * n * n * 4 * 2 < 40e9:  # the (BH, N, N) score matrix is materialised twice
        manual_result = manual_attention(qd.float(), kd.float(), vd.float(), args.masking, args.scale)
        row("manual matmul->softmax->matmul (fp32, GPU)", time_gpu(lambda: manual_attention(qd.float(), kd.float(), vd.float(), args.masking, args.scale), 1, 3))
    else:
        print("  skipped: score matrix would not fit")
    row(f"PyTorch-ROCm SDPA ({args.dtype}) [*]",
        time_gpu(lambda: F.scaled_dot_product_attention(qd, kd, vd, is_causal=args.masking, scale=args.scale), args.warmup, args.iters))
    print("  [*] the comparison column of the reference script, NOT a tuned competitor: on this image (torch 2.10 + ROCm 7.0 wheels) SDPA times like\n"
          "      its math backend (scores materialised: 30-50 TFLOP/s at the README shapes) -- the ratio to it says nothing about kernel quality;\n"
          "      the roofline fraction below does.")

    print("=== profiling minimal flash attention ===")
    out = torch.empty_like(qd)
    result = minimal_flash.forward(qd, kd, vd, args.masking, scale=args.scale, out=out)
    ms = fa.time_forward(qd, kd, vd, args.masking, scale=args.scale, warmup=args.warmup, iters=args.iters, out=out)
    row(f"flashattention_c_amd.forward ({args.dtype}, HIP, gfx950)", ms)
    if args.dtype == "bf16":
        print(f"  -> {flop / ms / 1e9 / 2500.0 * 100:.1f} % of the dense bf16 MFMA peak (2500 TFLOP/s)")
    else:
        # fp32 tensors: three bf16 MFMA products per contraction (hi/lo splits) -- 3x the algorithmic flop on the bf16 pipe
        print(f"  -> split products: {3 * flop / ms / 1e9 / 2500.0 * 100:.1f} % of the dense bf16 MFMA peak at 3x the algorithmic flop")
        ms_exact = fa.time_forward(qd, kd, vd, args.masking, scale=args.scale, kernel="exact", warmup=args.warmup, iters=args.iters)
        row("  same op in exact fp32 arithmetic (kernel=\"exact\")", ms_exact)
        print(f"  -> {flop / ms_exact / 1e9 / 157.3 * 100:.1f} % of the dense f32 MFMA peak (157.3 TFLOP/s)")

    ref = manual_result if manual_result is not None else F.scaled_dot_product_attention(
        qd.float(), kd.float(), vd.float(), is_causal=args.masking, scale=args.scale)
    tol = 1e-3 if args.dtype == "f32" else 2.5e-2
    err = (result.float() - ref).abs().max().item()
    if err < tol:
        print(f"[Correctness] attn values sanity check: PASSED (max abs err {err:.2e} < {tol:g})")
    else:
        print(f"[Correctness] attn values sanity check: FAILED (max abs err {err:.2e} >= {tol:g})")
    if args.dtype == "bf16":
        # the accurate bf16 path: same tensors, fp32 output -> FA_KERNEL_AUTO carries P as two bf16 terms (one launch) -- held to the fp32 bar
        o32 = torch.empty(qd.shape, dtype=torch.float32, device=qd.device)
        acc = minimal_flash.forward(qd, kd, vd, args.masking, scale=args.scale, out=o32)
        ms_acc = fa.time_forward(qd, kd, vd, args.masking, scale=args.scale, warmup=args.warmup, iters=args.iters, out=o32)
        row("flashattention_c_amd.forward (bf16 in, fp32 out: accurate P)", ms_acc)
        err_acc = (acc - ref).abs().max().item()
        verdict = "PASSED" if err_acc < 1e-3 else "FAILED"
        print(f"[Correctness] accurate bf16 path: {verdict} (max abs err {err_acc:.2e} vs the fp32 bar 0.001)")
    if args.print_tensors:
        print(result.cpu())
        print(ref.cpu())
    return 0 if err < tol else 1


if __name__ == "__main__":
    sys.exit(main())
