#!/usr/bin/env python3
"""bench.py -- the fused flash-attention forward on MI355X: fwd ms, achieved TFLOP/s, fraction of the MFMA roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c3|c2|c5] [--causal]

Contract (one JSON line on stdout from rank 0):
  * a "step" is one forward over one rank's (batch*heads, N, d) shard of synthetic randn data already resident in
    HBM; W untimed steps, then exactly K timed steps bracketed by barrier + torch.cuda.synchronize() on both sides;
    the time is the MAX over ranks; `value` = algorithmic FLOP of all ranks / that time, in TFLOP/s.
  * N = 1 workload = the configuration BASELINE.json's metric is quoted on: B=2 H=8 d=64 N=8192, bf16 MFMA path
    (config "c4").  N > 1: every rank runs that same shard (weak scaling; the batch*head axis shards
    with NO collective on the data path -- each slab is independent, /root/reference/src/flashattention.cu:144), and
    `extra.c5` carries BASELINE config 5, B=64 H=16 (1024 slabs) split across the ranks (strong scaling), timed with the same
    protocol.  `--workload c5` makes that the headline value instead.
  * `python bench.py --gpus N` with N > 1 works as typed: when it is not already running under torch.distributed.run it
    starts `python -m torch.distributed.run --nproc-per-node N ... bench.py` as a CHILD process (before anything here has
    touched a GPU) and relays rank 0's line.  Under torch.distributed.run (RANK / WORLD_SIZE set) it is one rank.
  * "roofline": dominant kernel's algorithmic FLOP per launch / its mean launch duration measured with HIP events on
    the launch stream (fa_time_forward in the C ABI) against the dense MFMA peak of the pipe that kernel computes on.
  * "cpu_baseline": PyTorch's CPU SDPA (fp32, scale 1.0 -- the reference bench's own oracle expression,
    bench_flashattention.py:36-40, as BASELINE.md section 3 prescribes) on this box's host cores, on a bounded sample of the
    same workload (rank 0, N = 1 only) -- a reported baseline, not the optimisation target.  The C oracle's timing is
    in `extra.cpu_oracle_port`.
The product path has no fallback: if the HIP library is missing this script fails.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}

WORKLOADS = {
    # name: (B, H, d, N, dtype, scaling)
    "c2": (8, 16, 64, 1024, "f32", "weak"),
    "c3": (2, 8, 64, 8192, "f32", "weak"),
    "c4": (2, 8, 64, 8192, "bf16", "weak"),
    "c5": (64, 16, 64, 8192, "bf16", "strong"),
}


def fwd_flop(bh: int, n: int, d: int, causal: bool) -> float:
    """Algorithmic FLOP of one forward: two GEMMs of 2*N*N*d per slab (softmax not counted), half of it when causal."""
    return (2.0 if causal else 4.0) * bh * float(n) * float(n) * d


def algorithmic_bytes(bh: int, n: int, d: int, elem: int) -> float:
    """HBM minimum: read Q, K, V once, write O once."""
    return 4.0 * bh * n * d * elem


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def under_launcher() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` typed directly: run the N ranks as a child torch.distributed.run job and relay its output.
    Nothing in this (parent) process has touched a GPU or may touch one afterwards."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    r = subprocess.run(cmd, env=env)
    return r.returncode


def init_dist(world: int, backend: str):
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend=backend)
    return dist


def timed_region(step_fn, steps: int, warmup: int, sync_fn, world: int, dist=None, device=None) -> float:
    """W untimed steps, then K timed steps bracketed by barrier + sync on both sides; returns MAX-over-ranks seconds."""
    import torch
    for _ in range(warmup):
        step_fn()
    sync_fn()
    if world > 1:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if world > 1:
        dist.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                       # each rank's own time: a straggler GPU is visible in extra.per_rank_ms
        PER_RANK_S[:] = [float(x.item()) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    else:
        PER_RANK_S[:] = [dt]
    return dt


PER_RANK_S = []   # seconds of the last timed_region on every rank (filled on all ranks)
C4_ONE_GPU_TFLOPS = 1245.0   # the c4 headline on one MI355X with round 6's kernels: 1207 - 1275 over its last five boxes (the driver's records of rounds 1 - 5,
                             # BENCH_r01 .. r05: 1164, 1212, 1220, 1171, 1192 -- mean 1190 -- times the 4.5 % the round's A/B runs add up to)
C5_ONE_GPU_MS = 14.3   # all 1024 slabs of config 5 on one MI355X: 14.16 - 14.48 ms over round 6's collections (rounds 3 - 5: 14.57 - 14.77), +- 4 % between boxes


def make_inputs(bh: int, n: int, d: int, dtype: str, device, seed: int):
    import torch
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    if getattr(device, "type", "cpu") == "cuda" and bh * n * d > (1 << 27):
        # big shards (c5: 1 GiB per tensor) are generated on the device; same distribution, per-rank seed
        g = torch.Generator(device=device).manual_seed(seed)
        return [torch.randn(bh, n, d, generator=g, device=device, dtype=torch.float32).to(tdt) for _ in range(3)]
    g = torch.Generator(device="cpu").manual_seed(seed)
    # generated on the host in fp32 (as the reference bench does, bench_flashattention.py:31-33), then moved
    return [torch.randn(bh, n, d, generator=g, dtype=torch.float32).to(tdt).to(device) for _ in range(3)]


def cpu_oracle_port(total_heads: int, n: int, d: int, causal: bool, scale: float):
    """The C oracle (oracle/attention_oracle.c, fp64 accumulate, OpenMP) on a bounded sample: whole heads of the same N and d."""
    import numpy as np
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    cores = orc.num_threads()
    q, k, v = (rng.standard_normal((1, n, d)).astype(np.float32) for _ in range(3))
    t0 = time.perf_counter()
    orc.attention_f64(q, k, v, causal=causal, scale=scale)   # one full head: sizes the sample (and warms the threads)
    dt = time.perf_counter() - t0
    heads = int(max(1, min(total_heads, 10.0 // max(dt, 1e-3))))
    if heads > 1:
        q, k, v = (rng.standard_normal((heads, n, d)).astype(np.float32) for _ in range(3))
        t0 = time.perf_counter()
        orc.attention_f64(q, k, v, causal=causal, scale=scale)
        dt = time.perf_counter() - t0
    tf = fwd_flop(heads, n, d, causal) / dt / 1e12
    return {"value": round(tf, 5), "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": f"{heads} of {total_heads} heads at N={n} d={d}, fp64-accumulate C oracle (OpenMP), {dt:.1f} s",
            "ms_per_full_workload_est": round(dt / heads * total_heads * 1e3, 1)}


def cpu_sdpa_baseline(total_heads: int, n: int, d: int, causal: bool, scale: float, budget_s: float = 20.0):
    """PyTorch CPU SDPA, fp32: the reference bench's oracle expression (softmax(q k^T) v, scale 1.0 --
    bench_flashattention.py:36-40; identical to F.scaled_dot_product_attention(scale=1.0) on CPU)."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(1, n, d, generator=g) for _ in range(3))
    t0 = time.perf_counter()
    F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale)   # one head: sizes the sample, warms the thread pool
    dt1 = time.perf_counter() - t0
    heads = int(max(1, min(total_heads, (budget_s / 4.0) // max(dt1, 1e-4))))
    q, k, v = (torch.randn(heads, n, d, generator=g) for _ in range(3))
    F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[1]
    return {"value": round(fwd_flop(heads, n, d, causal) / med / 1e12, 5), "unit": "TFLOP/s", "cores": torch.get_num_threads(),
            "kind": "reference",
            "sample": f"torch CPU scaled_dot_product_attention fp32 (= the reference bench's oracle expression, bench_flashattention.py:36-40), "
                      f"{heads} of {total_heads} heads at N={n} d={d}, median of 3 after 1 warm-up, {med * 1e3:.0f} ms per call",
            "host_cpus": os.cpu_count(), "ms_per_full_workload_est": round(med / heads * total_heads * 1e3, 1)}


def lib_sha256() -> str:
    """sha256 of the library this process loads: ties a committed PMC traffic figure to the binary it was measured on."""
    import hashlib
    from flashattention_c_amd import _cabi
    h = hashlib.sha256()
    with open(_cabi.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def time_stats(fa, tensors, reps: int = 5, **kw):
    """`reps` repetitions of fa.time_forward's iters loop (HIP events on the launch stream inside the C ABI): min / median / p90 of
    the per-repetition means, so that a slow outlier repetition is visible instead of averaged in (the reference's
    benchmark_kernel, src/llm.c/common.h:108-124, reports one mean)."""
    warm = kw.pop("warmup", 3)
    ms = []
    for r in range(reps):
        ms.append(fa.time_forward(*tensors, warmup=warm if r == 0 else 1, **kw))
    ms.sort()
    p90 = ms[min(len(ms) - 1, int(round(0.9 * (len(ms) - 1))))]
    return {"min": round(ms[0], 4), "median": round(ms[len(ms) // 2], 4), "p90": round(p90, 4), "reps": reps, "iters_per_rep": kw.get("iters", 20)}


def load_pmc_traffic():
    """HBM bytes per launch from the committed PMC profile of this same command (profiles/), if present."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                return json.load(f)
        except Exception:
            return None
    return None


def kernel_peak(kernel_name: str, dtype: str):
    """Dense peak of the matrix pipe the named kernel computes on, and how its executed FLOP relate to the algorithmic ones."""
    if kernel_name == "fa_fwd_f32_kernel":
        return PEAK_TFLOPS["f32"], "fp32 MFMA (v_mfma_f32_32x32x2_f32), 1x the algorithmic FLOP"
    if kernel_name == "fa_fwd_f32_split_kernel" and dtype == "f32":
        return PEAK_TFLOPS["bf16"] / 3.0, "16-bit MFMA dense peak / 3: three products per contraction (fp16 hi/lo terms for Q.K^T, bf16 hi/lo terms for P.V)"
    if kernel_name == "fa_fwd_f32_split_kernel":
        return PEAK_TFLOPS["bf16"] / 2.0, "bf16 MFMA peak / 2: two bf16 products per contraction (hi/lo terms of P and Q')"
    return PEAK_TFLOPS["bf16"], "bf16 / fp16 MFMA dense peak, 1x the algorithmic FLOP"


# ---- the bench checks what it timed (the reference validates the very tensor it just timed: bench_flashattention.py:69-79) -------------
# Tolerances (max-abs against the fp32 reference of the same op on identical inputs).  The bf16-P figures are the regression thresholds
# tests/test_gpu_parity.py uses for seeded N(0, 1) data -- typical values with head room, not bounds: the bound for ANY data is
# (2^-8 + 2^-10) * max_row sum_j w_j |v_j - O| (+ 2^-8 |O| for a bf16 output), tests/adversarial.py: p_rounding_bound, ~1.7e-2 / ~3.3e-2 for
# this data; tests/test_gpu_adversarial.py asserts it on constructed worst cases and on these very shapes.
TOLERANCE = {
    "bf16_p_bf16_out": 2.5e-2,   # bf16 P (8 significant bits) + the output's own bf16 rounding, unscaled unit-variance logits
    "bf16_p_f32_out": 1.2e-2,
    "accurate": 1e-3,            # the north star's bar (BASELINE.json): fp32 output, P as two bf16 terms
    "f32": 1e-3,                 # fp32 tensors, either arithmetic
}
VALIDATION_FAILURES = []


def validate_output(fa, q, k, v, out, causal, scale, tol, what, slabs=None, oracle_slab=None):
    """Compare the `out` the timed launches wrote with (a) the rung-0 kernel (fa_naive_f32_kernel: one wave per query row, plain fp32
    loops -- an independent implementation) ON THE DEVICE for `slabs`, and (b) -- `oracle_slab`, rank 0's cpu_baseline leg only -- the
    fp64 CPU oracle for one slab.  Outside every timed region.  Returns the entry for the JSON line; records a failure when outside tol."""
    import torch
    bh = q.shape[0]
    slabs = [0, bh - 1] if slabs is None else slabs
    worst = 0.0
    for i in sorted(set(slabs)):
        qi, ki, vi = (t[i:i + 1].float().contiguous() for t in (q, k, v))
        ref = fa.forward(qi, ki, vi, causal, scale=scale, kernel="naive")
        err = (out[i:i + 1].float() - ref).abs().max().item()
        worst = max(worst, err if err == err else float("inf"))
    ent = {"max_abs_err": float(f"{worst:.3e}"), "tolerance": tol,
           "checked": f"the output the timed launches wrote, slabs {sorted(set(slabs))} of {bh} against the rung-0 fp32 kernel on the device"}
    if oracle_slab is not None:
        from oracle import oracle as orc   # the checker, never the thing measured
        i = oracle_slab
        qi, ki, vi = (t[i:i + 1].float().cpu().numpy() for t in (q, k, v))
        ref64 = orc.attention_f64(qi, ki, vi, causal=causal, scale=scale)
        e64 = float(abs(out[i:i + 1].float().cpu().numpy().astype("float64") - ref64).max())
        ent["max_abs_err_vs_fp64_oracle"] = float(f"{e64:.3e}")
        ent["checked"] += f"; slab {i} against the fp64 CPU oracle"
        worst = max(worst, e64 if e64 == e64 else float("inf"))
    ent["ok"] = bool(worst <= tol)
    if not ent["ok"]:
        VALIDATION_FAILURES.append(f"{what}: max-abs error {worst:.3e} > tolerance {tol:g}")
    return ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c4")
    ap.add_argument("--causal", action="store_true")
    ap.add_argument("--scale", type=float, default=1.0, help="softmax scale; the reference hard-wires 1.0")
    ap.add_argument("--prewarm-ms", type=float, default=200.0,
                    help="untimed device warm-up before the W warm-up steps: an idle MI355X needs ~100 ms of load before its "
                         "clocks settle (the first ~100 launches of a 0.3 ms kernel run ~10 %% slow)")
    ap.add_argument("--accurate", action="store_true",
                    help="bf16 workloads: ask for the fp32 accumulator as output -- FA_KERNEL_AUTO then carries P as two bf16 terms (the accurate path, one launch); "
                         "used by profiles/collect.sh to profile that kernel chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--cpu-stub", action="store_true",
                    help="TEST HOOK (tests/test_distributed_gloo.py): exercise launcher, sharding and timing protocol on CPU ranks with a "
                         "sleep instead of the kernel; the line it prints says so and carries no measurement")
    ap.add_argument("--same-device", action="store_true",
                    help="TEST HOOK (tests/test_gpu_dist.py): every rank uses cuda:0 -- two ranks on a one-GPU box (with --backend gloo) exercise "
                         "the world > 1 code of this script; the line says so")
    args = ap.parse_args()

    if args.gpus > 1 and not under_launcher():
        # before anything touches a GPU: the ranks are children, this process only relays (never exec from a GPU process)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    rank, world, local = dist_env()
    args.gpus = world
    if args.cpu_stub:
        return stub_main(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the flash-attention forward has no CPU path")
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = init_dist(world, args.backend) if world > 1 else None

    import flashattention_c_amd as fa  # raises if libflashattn_amd.so is missing -- no fallback
    from flashattention_c_amd import _cabi

    B, H, d, n, dtype, scaling = WORKLOADS[args.workload]
    total_bh = B * H
    if scaling == "strong":
        b0, b1 = fa.shard_range(total_bh, world, rank)
        bh = b1 - b0
        global_bh = total_bh
    else:
        bh = total_bh
        global_bh = total_bh * world
    causal = bool(args.causal)
    q, k, v = make_inputs(bh, n, d, dtype, device, seed=rank)
    out = torch.empty(q.shape, dtype=torch.float32, device=device) if (args.accurate and dtype == "bf16") else torch.empty_like(q)

    def step():
        fa.forward(q, k, v, causal, scale=args.scale, out=out)

    def prewarm(fn):
        if args.prewarm_ms > 0:  # untimed: bring the device out of its idle power state
            t_end = time.perf_counter() + args.prewarm_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()

    prewarm(step)
    slow0 = fa.stats()   # (after the warm-up forwards: the kernels' slow-path counters exist from the first forward outside a capture)
    # The line's two clocks -- K back-to-back forwards from Python between barriers (ms_per_step -> `value`) and the C ABI's event-timed
    # loops on the launch stream (kernel_ms -> `roofline`) -- are taken in ALTERNATION inside one warm state, and when they disagree by more
    # than 2 % (a clock ramp, a noisy neighbour on the box: round 4's line read 4.2 % between them) the pair is measured again, in this
    # process, at most three times.  Every attempt is a complete timed region of exactly K steps; the LAST attempt is the one reported,
    # all of them are listed in extra.timing_attempts.
    kiters = max(10, min(args.steps, 50))
    attempts = []
    kstats = None
    for attempt in range(3):
        dt = timed_region(step, args.steps, args.warmup if attempt == 0 else 2, torch.cuda.synchronize, world, dist, device)
        per_rank_s = list(PER_RANK_S)
        agree = torch.zeros(1, dtype=torch.int32, device=device)
        if rank == 0:
            kstats = time_stats(fa, (q, k, v), reps=5, causal=causal, scale=args.scale, warmup=3, iters=kiters, out=out)
            ratio = kstats["median"] / (dt / args.steps * 1e3)
            attempts.append({"ms_per_step": round(dt / args.steps * 1e3, 4), "kernel_ms": kstats["median"], "kernel_ms_over_ms_per_step": round(ratio, 4)})
            agree[0] = 1 if abs(ratio - 1.0) <= 0.02 else 0
        if world > 1:
            dist.broadcast(agree, src=0)
        if int(agree.item()) == 1:
            break
    ms_per_step = dt / args.steps * 1e3
    per_rank_ms = [round(x / args.steps * 1e3, 4) for x in per_rank_s]
    value = fwd_flop(global_bh, n, d, causal) * args.steps / dt / 1e12
    route = fa.last_forward_route()   # 0: single launch; 1 / 2: primary / fallback kernel of a conditional chain (fp32 guard)
    # what the timed launches wrote, checked before anything else runs (rank 0; every rank's shard has the same distribution)
    head_tol = TOLERANCE["f32"] if dtype == "f32" else TOLERANCE["accurate"] if args.accurate else TOLERANCE["bf16_p_bf16_out"]
    head_check = None
    if rank == 0:
        head_check = validate_output(fa, q, k, v, out, causal, args.scale, head_tol, f"headline ({args.workload})",
                                     oracle_slab=0 if (world == 1 and not args.no_cpu_baseline) else None)

    extras = {"per_rank_ms": per_rank_ms} if world > 1 else {}
    if rank == 0:
        extras["timing_attempts"] = attempts
        # what the kernels themselves counted while both clocks ran (fa_read_device_counters): a timed launch that took a slow path -- a tile
        # redone behind a failed optimistic attempt, a workgroup redone in fp32 arithmetic -- would show here
        torch.cuda.synchronize()
        slow1 = fa.stats()
        extras["slow_paths_during_timing"] = {key: slow1[key] - slow0[key] for key in ("tiles_redone", "workgroups_fp32")}
    if world > 1:   # what the collective layer saw, not what the environment said
        extras["ranks_seen"] = int(dist.get_world_size())
        extras["dist_backend"] = str(dist.get_backend())
    if world > 1 and args.workload == "c4" and not causal and not args.accurate:
        # weak scaling, no collective: N ranks should read N x the one-GPU value (1130 - 1220 TFLOP/s over three rounds' boxes)
        extras["weak_scaling"] = {"predicted_value": round(world * C4_ONE_GPU_TFLOPS, 1),
                                  "efficiency_vs_prediction": round(value / (world * C4_ONE_GPU_TFLOPS), 4),
                                  "what": f"value / (n_gpus x {C4_ONE_GPU_TFLOPS:g} TFLOP/s, the one-GPU c4 figure; +- 4 % between boxes)"}
    # ---- BASELINE config 5 (B=64 H=16, 1024 slabs) sharded over the ranks, same timing protocol: at N = 1 the whole of it on one GPU
    if not args.no_extras and args.workload == "c4":
        B5, H5, d5, n5, dt5, _ = WORKLOADS["c5"]
        s0, s1 = fa.shard_range(B5 * H5, world, rank)
        q5, k5, v5 = make_inputs(s1 - s0, n5, d5, dt5, device, seed=100 + rank)
        o5 = torch.empty_like(q5)
        step5 = lambda: fa.forward(q5, k5, v5, causal, scale=args.scale, out=o5)  # noqa: E731
        k5_steps = max(3, min(args.steps, 10))
        dt5s = timed_region(step5, k5_steps, 2, torch.cuda.synchronize, world, dist, device)
        tf5 = fwd_flop(B5 * H5, n5, d5, causal) * k5_steps / dt5s / 1e12
        extras["c5"] = {"workload": f"B={B5} H={H5} d={d5} N={n5} {dt5}: 1024 slabs, contiguous split over {world} GPU(s), no collective",
                        "scaling": "strong", "n_gpus": world, "bh_per_gpu": s1 - s0, "steps": k5_steps,
                        "ms_per_step": round(dt5s / k5_steps * 1e3, 4), "per_rank_ms": [round(x / k5_steps * 1e3, 4) for x in PER_RANK_S],
                        "tflops": round(tf5, 2), "tflops_per_gpu": round(tf5 / world, 2),
                        "frac_bf16_mfma_peak_per_gpu": round(tf5 / world / PEAK_TFLOPS["bf16"], 4)}
        # no collective on the path: N GPUs should take 1/N of the one-GPU time (DESIGN.md section 6 has the table); a straggler GPU or
        # an RCCL-init artefact shows as a ratio well below 1
        pred5 = C5_ONE_GPU_MS / world
        extras["c5"]["predicted_ms_per_step"] = round(pred5, 3)
        extras["c5"]["efficiency_vs_prediction"] = round(pred5 / (dt5s / k5_steps * 1e3), 4)
        if rank == 0:   # one shard's output against the rung-0 kernel (first and last slab of this rank's shard)
            chk5 = validate_output(fa, q5, k5, v5, o5, causal, args.scale, TOLERANCE["bf16_p_bf16_out"], "extra.c5")
            extras["c5"]["max_abs_err"], extras["c5"]["tolerance"] = chk5["max_abs_err"], chk5["tolerance"]
        del q5, k5, v5, o5
        torch.cuda.empty_cache()

    # ---- roofline of the dominant kernel: HIP events on the launch stream, inside the C ABI
    roof = None
    if rank == 0:
        L = _cabi.lib()
        dt_id = (_cabi.FA_DTYPE_BF16_OUT_F32 if args.accurate else _cabi.FA_DTYPE_BF16) if dtype == "bf16" else _cabi.FA_DTYPE_F32
        kname = L.fa_kernel_name_for(dt_id, d, int(causal), bh, n).decode()
        if dtype == "f32" and route == 2:
            kname = "fa_fwd_f32_kernel"   # the range guard sent (part of) this workload to fp32 arithmetic
        kms = kstats["median"]   # (of the last attempt above: taken right behind the timed region `value` comes from)
        achieved = fwd_flop(bh, n, d, causal) / (kms * 1e-3) / 1e12
        elem = 2 if dtype == "bf16" else 4
        pmc = load_pmc_traffic() or {}
        peak, peak_note = kernel_peak(kname, dtype)
        # traffic: HBM bytes of one launch from the committed rocprofv3 PMC passes of this command -- a STATIC figure, quoted only
        # while the kernel it was measured on is still the kernel this run launches
        # ... AND while the library loaded here is the very binary the passes ran on (sha256 recorded by profiles/collect.sh)
        sha = lib_sha256()
        tkey = args.workload + ("acc" if (args.accurate and dtype == "bf16") else "")
        traffic = (pmc.get(f"{tkey}_hbm_bytes_per_launch")
                   if pmc.get(f"{tkey}_kernel") == kname and pmc.get("lib_sha256") == sha and not causal else None)
        roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),   # (from kernel_ms, the event-timed loop on the launch stream)
                # the same fraction from the line's other clock (the two agree to 2 % or `timing_note` below says how far apart they stayed)
                "frac_from_ms_per_step": round(fwd_flop(bh, n, d, causal) / (ms_per_step * 1e-3) / 1e12 / peak, 4),
                "traffic": traffic,
                "traffic_source": (f"profiles/pmc_traffic.json (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                   f"{pmc.get('_tag', '?')}; 2*FETCH_SIZE + WRITE_SIZE)") if traffic is not None else None,
                "kernel": kname, "kernel_ms": round(kms, 4), "kernel_ms_stats": kstats, "lib_sha256": sha[:16], "peak_note": peak_note,
                "algorithmic_flop_per_launch": fwd_flop(bh, n, d, causal),
                "algorithmic_hbm_bytes_per_launch": algorithmic_bytes(bh, n, d, elem),
                "hbm_gbps_at_algorithmic_bytes": round(algorithmic_bytes(bh, n, d, elem) / (kms * 1e-3) / 1e9, 1)}
        roof["max_abs_err"], roof["tolerance"] = head_check["max_abs_err"], head_check["tolerance"]
        roof["accuracy"] = head_check
        # the two clocks of this line: K back-to-back forwards from Python (ms_per_step) and the C ABI's event-timed loop (kernel_ms)
        ratio = kms / ms_per_step if ms_per_step > 0 else 0.0
        roof["kernel_ms_over_ms_per_step"] = round(ratio, 4)
        roof["timing_note"] = (f"{len(attempts)} attempt(s): " + ("the two clocks agree to 2 %" if abs(ratio - 1.0) <= 0.02 else
                               f"the two clocks stayed {abs(ratio - 1.0) * 100:.1f} % apart after {len(attempts)} attempts (extra.timing_attempts)"))
        if abs(ratio - 1.0) > 0.02:
            roof["timing_warning"] = (f"kernel_ms (median of {kstats['reps']} event-timed loops) and ms_per_step differ by {abs(ratio - 1.0) * 100:.1f} %: "
                                      f"min / median / p90 of the loops = {kstats['min']} / {kstats['median']} / {kstats['p90']} ms; "
                                      "expect ms_per_step above kernel_ms when a forward is several launches (key shares + combine: host enqueue per launch) and "
                                      "either one high right after an idle period (clock ramp)")
        if dtype == "f32":
            roof["arithmetic"] = {0: "single launch", 1: "split products on the 16-bit pipes: Q.K^T as fp16 hi + lo terms, P.V as bf16 hi + lo terms (range guard quiet)",
                                  2: "exact fp32 for some workgroups (range guard fired)"}[route]
        if not args.no_extras:
            # the same launches captured into one hipGraph and replayed (median of three timed replays); reported beside the
            # stream-launch figures above, never instead of them.  per_launch_delta_us = what a graph node costs more (+) or less (-)
            # than the same launch enqueued on a stream
            try:
                gi = max(10, min(args.steps, 50))
                gms = fa.time_forward(q, k, v, causal, scale=args.scale, warmup=3, iters=gi, out=out, graph=True)
                gtf = fwd_flop(bh, n, d, causal) / (gms * 1e-3) / 1e12
                extras["graph_replay"] = {"kernel_ms": round(gms, 4), "tflops": round(gtf, 2),
                                          "frac_mfma_peak": round(gtf / peak, 4), "nodes_per_replay": gi, "replays_timed": 3,
                                          "per_launch_delta_us_vs_stream": round((gms - kms) * 1e3, 2)}
            except Exception as e:  # pragma: no cover - informational only
                extras["graph_replay"] = {"error": repr(e)}
        if not args.no_extras and world == 1 and args.workload == "c4":
            extras.update(c4_side_measurements(fa, _cabi, q, k, v, causal, args, device))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_sdpa_baseline(total_bh, n, d, causal, args.scale)
        try:
            extras["cpu_oracle_port"] = cpu_oracle_port(total_bh, n, d, causal, args.scale)
        except Exception as e:  # pragma: no cover - informational only
            extras["cpu_oracle_port"] = {"error": repr(e)}

    if world > 1:
        dist.barrier()
    if rank == 0:
        line = {
            "metric": "flash-attention fwd achieved TFLOP/s (fwd ms in ms_per_step; % MFMA peak in roofline.frac), B=2 H=8 d=64 N=8192"
            if args.workload in ("c3", "c4") else f"flash-attention fwd achieved TFLOP/s, workload {args.workload}",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic randn (seeded), resident in HBM" + (" -- TEST HOOK --same-device: all ranks share cuda:0, not a measurement" if args.same_device else ""),
            "config": {"workload": f"{args.workload}: B={B} H={H} d={d} N={n} {dtype}, {'causal' if causal else 'non-causal'}, "
                                   f"scale={args.scale:g}" + (", fp32 output (accurate P)" if args.accurate and dtype == "bf16" else "")
                                   + (" per GPU" if scaling == "weak" and world > 1 else ""),
                       "global_bh": global_bh, "bh_per_gpu": bh, "seq_len": n, "head_dim": d,
                       "parallelism": f"batch*head sharded x{world}, no collective"},
            "roofline": roof, "cpu_baseline": cpu, "extra": extras,
        }
        # the target's OTHER half: the fastest path whose output is inside 1e-3 of the fp32 reference at the reference's scale 1 --
        # the headline kernel (bf16 P, bf16 out) is not (roofline.max_abs_err); this block is the accurate path on the same tensors
        acc = extras.get("c4_accurate_mode") if isinstance(extras.get("c4_accurate_mode"), dict) else None
        if args.accurate and roof is not None:
            acc = {"kernel": roof["kernel"], "kernel_ms": roof["kernel_ms"], "tflops": roof["achieved"], "frac_mfma_peak": roof["frac"],
                   "max_abs_err": roof["max_abs_err"], "tolerance": roof["tolerance"]}
        if acc is not None and "kernel_ms" in acc:
            line["roofline_at_1e-3"] = {"bound": "mfma", "achieved": acc["tflops"], "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                                        "frac": acc["frac_mfma_peak"], "kernel": acc.get("kernel"), "kernel_ms": acc["kernel_ms"],
                                        "max_abs_err": acc.get("max_abs_err"), "tolerance": acc.get("tolerance"),
                                        "traffic": ((load_pmc_traffic() or {}).get("c4acc_hbm_bytes_per_launch")
                                                    if args.workload == "c4" and not causal and (load_pmc_traffic() or {}).get("c4acc_kernel") == acc.get("kernel")
                                                    and (load_pmc_traffic() or {}).get("lib_sha256") == lib_sha256() else None),
                                        "algorithmic_hbm_bytes_per_launch": algorithmic_bytes(bh, n, d, 2) + bh * n * d * 2.0,   # fp32 output: 2 more bytes per element
                                        "what": "same tensors, fp32 output through FA_KERNEL_AUTO (P as bf16 hi + lo, one launch): the figure to "
                                                "hold against BASELINE.json's '>= 60 % of peak within 1e-3'"}
        line["validation"] = {"status": "FAILED" if VALIDATION_FAILURES else "ok", "failures": list(VALIDATION_FAILURES),
                              "what": "every figure that carries max_abs_err was checked on the tensor its timed launches wrote (rung-0 kernel "
                                      "on the device; headline also against the fp64 CPU oracle); a failure makes this process exit 1"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and VALIDATION_FAILURES:
        print("bench.py: output validation FAILED: " + "; ".join(VALIDATION_FAILURES), file=sys.stderr)
        raise SystemExit(1)


def c4_side_measurements(fa, _cabi, q, k, v, causal, args, device):
    """Rank 0, N = 1, workload c4: every other shape SURVEY.md section 8(d) / the README name, a few launches each (HIP events on the
    launch stream; algorithmic TFLOP/s).  Informational: the headline value and roofline above are c4's."""
    import torch
    ex = {}
    bh, n, d = q.shape
    flop = fwd_flop(bh, n, d, causal)

    def timed(what, fn_kwargs, flop_, tensors=(q, k, v), peak=PEAK_TFLOPS["bf16"], note=None, warm=30, iters=20, tol=None):
        try:
            st = time_stats(fa, tensors, reps=5, warmup=warm, iters=max(2, iters // 2), **fn_kwargs)
            ms = st["median"]
            tf = flop_ / (ms * 1e-3) / 1e12
            ent = {"kernel_ms": round(ms, 4), "ms_min_median_p90": [st["min"], st["median"], st["p90"]], "tflops": round(tf, 2),
                   "frac_mfma_peak": round(tf / peak, 4)}
            if note:
                ent["what"] = note
            if tol is not None and fn_kwargs.get("out") is not None:   # the tensor the timed launches wrote
                chk = validate_output(fa, *tensors, fn_kwargs["out"], fn_kwargs.get("causal", False), fn_kwargs.get("scale", 1.0), tol, what)
                ent["max_abs_err"], ent["tolerance"] = chk["max_abs_err"], chk["tolerance"]
            ex[what] = ent
        except Exception as e:  # pragma: no cover - informational only
            ex[what] = {"error": repr(e)}

    # the accurate bf16 path on the same tensors: what fa_forward picks for an fp32 output (P as bf16 hi + bf16 lo, FA_KERNEL_PB2)
    o32 = torch.empty(q.shape, dtype=torch.float32, device=device)
    timed("c4_accurate_mode", dict(causal=causal, scale=args.scale, out=o32), flop, tol=TOLERANCE["accurate"],
          note="FA_KERNEL_AUTO for an fp32 output: bf16 Q, K, V; P as bf16 hi + bf16 lo (lo = the exact residual p - hi from one v_dot2c_f32_bf16 "
               "per element; twice the P.V and row-sum MFMAs), optimistic softmax with the rescaled redo behind it, fp32 out -- ONE launch, no "
               "scratch; frac = algorithmic FLOP over the dense bf16 peak")
    if "kernel_ms" in ex.get("c4_accurate_mode", {}):
        ex["c4_accurate_mode"]["frac"] = ex["c4_accurate_mode"]["frac_mfma_peak"]
        ex["c4_accurate_mode"]["route"] = fa.last_forward_route()
        ex["c4_accurate_mode"]["kernel"] = _cabi.lib().fa_kernel_name_for(_cabi.FA_DTYPE_BF16_OUT_F32, d, int(causal), bh, n).decode()
    timed("c4_bf16_p_f32_out", dict(causal=causal, scale=args.scale, out=o32, kernel="mfma"), flop, tol=TOLERANCE["bf16_p_f32_out"],
          note="the headline kernel (bf16 P) storing its fp32 accumulator: separates P's rounding from the bf16 output's")
    timed("c4_accurate_mode_split", dict(causal=causal, scale=args.scale, out=o32, kernel="split"), flop, peak=PEAK_TFLOPS["bf16"] / 2.0,
          tol=TOLERANCE["accurate"],
          note="the round-1 accurate mode (hi + lo bf16 terms of P and Q' in the split kernel; two products per contraction), frac of bf16 peak / 2", iters=10)
    del o32
    out = torch.empty_like(q)
    # causal reported separately; 1/sqrt(d) as a second line
    timed("c4_causal", dict(causal=True, scale=args.scale, out=out), fwd_flop(bh, n, d, True), tol=TOLERANCE["bf16_p_bf16_out"],
          note="c4 shape, causal (algorithmic FLOP halved)")
    timed("c4_scale_rsqrt_d", dict(causal=causal, scale=d ** -0.5, out=out), flop, tol=1e-3,
          note="c4 shape at scale 1/sqrt(d) instead of the reference's 1.0: the bf16-P kernel, bf16 output, is inside 1e-3 here")
    # one slab of the same length: a grid that leaves the chip idle -> key-split launch (S workgroups per q-tile + combine)
    q1, k1, v1 = make_inputs(1, n, d, "bf16", device, seed=3)
    timed("bh1_n8192_bf16_keysplit", dict(causal=False, scale=args.scale), fwd_flop(1, n, d, False), tensors=(q1, k1, v1),
          note="B*H = 1 at the c4 length, bf16 tensors: 32 q-tiles x 8 key shares + combine instead of 32 workgroups", warm=100, iters=50)
    timed("bh1_n8192_bf16_unsplit", dict(causal=False, scale=args.scale, kernel="mfma:50"), fwd_flop(1, n, d, False), tensors=(q1, k1, v1),
          note="the same launch without the key split (kernel=\"mfma:50\": 32 workgroups of 256 rows over all 8192 keys)", warm=100, iters=50)
    timed("bh1_n8192_bf16_causal_keysplit", dict(causal=True, scale=args.scale), fwd_flop(1, n, d, True), tensors=(q1, k1, v1),
          note="B*H = 1, causal: key shares of 1024 keys (multiples of the tile height; shares above a tile's rows are empty and skipped) + combine", warm=100, iters=50)
    del q1, k1, v1
    # causal launches of up to 256 tiles are key-split as well (a causal launch lasts as long as its heaviest tile): B*H = 8
    q8, k8, v8 = make_inputs(8, n, d, "bf16", device, seed=4)
    timed("bh8_n8192_bf16_causal_keysplit", dict(causal=True, scale=args.scale), fwd_flop(8, n, d, True), tensors=(q8, k8, v8),
          note="B*H = 8, causal, bf16 tensors: 256 tiles x 2 key shares + combine (FA_KERNEL_AUTO)", warm=100, iters=50)
    timed("bh8_n8192_bf16_causal_unsplit", dict(causal=True, scale=args.scale, kernel="mfma:50"), fwd_flop(8, n, d, True), tensors=(q8, k8, v8),
          note="the same launch without the key split (kernel=\"mfma:50\"): round 2's path", warm=100, iters=50)
    del q8, k8, v8
    # fp32 tensors, one slab: the split kernel over key shares inside the guarded chain
    qf, kf, vf = make_inputs(1, n, d, "f32", device, seed=5)
    timed("bh1_n8192_f32_keysplit", dict(causal=False, scale=args.scale), fwd_flop(1, n, d, False), tensors=(qf, kf, vf), peak=PEAK_TFLOPS["bf16"] / 3.0,
          note="B*H = 1, fp32 tensors (FA_KERNEL_AUTO): split kernel over 8 key shares (each guarding its own keys) + combine; frac of the 16-bit MFMA peak / 3", warm=100, iters=50)
    timed("bh1_n8192_f32_unsplit", dict(causal=False, scale=args.scale, kernel="split"), fwd_flop(1, n, d, False), tensors=(qf, kf, vf), peak=PEAK_TFLOPS["bf16"] / 3.0,
          note="the same tensors through kernel=\"split\" (one launch, no key split, no guard): round 2's path", warm=100, iters=50)
    timed("bh1_n8192_f32_causal_keysplit", dict(causal=True, scale=args.scale), fwd_flop(1, n, d, True), tensors=(qf, kf, vf), peak=PEAK_TFLOPS["bf16"] / 3.0,
          note="B*H = 1, fp32 tensors, causal (FA_KERNEL_AUTO): key shares of 1024 keys in the share's local coordinates, empty shares skipped", warm=100, iters=50)
    timed("bh1_n8192_f32_causal_unsplit", dict(causal=True, scale=args.scale, kernel="split"), fwd_flop(1, n, d, True), tensors=(qf, kf, vf), peak=PEAK_TFLOPS["bf16"] / 3.0,
          note="the same tensors through kernel=\"split\" (no key split, no guard)", warm=100, iters=50)
    del qf, kf, vf
    # README rows 2 and 4 (d = 32), bf16 and fp32 tensors
    for name, (B2, H2, n2) in (("d32_n8192", (2, 8, 8192)), ("d32_n1024", (8, 16, 1024))):
        q2, k2, v2 = make_inputs(B2 * H2, n2, 32, "bf16", device, seed=2)
        fl2 = fwd_flop(B2 * H2, n2, 32, causal)
        timed(name + "_bf16", dict(causal=causal, scale=args.scale), fl2, tensors=(q2, k2, v2), note=f"README shape B={B2} H={H2} d=32 N={n2}, bf16 tensors",
              warm=30 if n2 > 2048 else 200, iters=20 if n2 > 2048 else 100)
        q3, k3, v3 = (t.float() for t in (q2, k2, v2))
        timed(name + "_f32", dict(causal=causal, scale=args.scale), fl2, tensors=(q3, k3, v3), peak=PEAK_TFLOPS["bf16"] / 3.0,
              note=f"README shape B={B2} H={H2} d=32 N={n2}, fp32 tensors (FA_KERNEL_AUTO), frac of bf16 peak / 3",
              warm=30 if n2 > 2048 else 200, iters=10 if n2 > 2048 else 100)
        del q2, k2, v2, q3, k3, v3
    # the same shape with fp32 tensors (config c3) and the README shape (c2): FA_KERNEL_AUTO (round 5: Q.K^T as fp16 hi + lo terms, P.V as
    # bf16 hi + lo terms, behind the range guard; the route says which arithmetic produced the output) and the fp32-arithmetic kernel beside it
    AUTO_ARITH = ("Q.K^T: 3 v_mfma_f32_32x32x16_f16 products of fp16 hi/lo splits (22 bits); P.V: 3 v_mfma_f32_32x32x16_bf16 products of bf16 hi/lo "
                  "splits; fp32 accumulate; range guard with its in-kernel fp32 fallback included in the time")

    def f32_pair(name, workload, q2, k2, v2, caus, warm, iters):
        fl2 = fwd_flop(q2.shape[0], q2.shape[1], q2.shape[2], caus)
        ent = {"workload": workload}
        o2 = torch.empty_like(q2)
        for label, kern in (("auto", "auto"), ("exact", "exact")):
            st2 = time_stats(fa, (q2, k2, v2), reps=5, causal=caus, scale=args.scale, kernel=kern, warmup=warm, iters=iters, out=o2)
            ms2 = st2["median"]
            tf2 = fl2 / (ms2 * 1e-3) / 1e12
            r = fa.last_forward_route()   # (before the check below launches the rung-0 kernel: the route is this thread's LAST forward's)
            chk2 = validate_output(fa, q2, k2, v2, o2, caus, args.scale, TOLERANCE["f32"], f"{name}.{label}")
            ent[label] = {"ms": round(ms2, 4), "ms_min_median_p90": [st2["min"], st2["median"], st2["p90"]], "tflops": round(tf2, 2),
                          "max_abs_err": chk2["max_abs_err"], "tolerance": chk2["tolerance"]}
            if label == "auto":
                ent[label].update(arithmetic=AUTO_ARITH if r == 1 else "the range guard fired: some workgroups in exact fp32", route=r,
                                  frac_16bit_mfma_peak_at_3x_flop=round(3.0 * tf2 / PEAK_TFLOPS["bf16"], 4))
            else:
                ws = fa.workspace_bytes(q2.shape[0], q2.shape[1], q2.shape[2], caus, kernel="exact")
                ent[label].update(arithmetic="v_mfma_f32_32x32x2_f32 (FA_KERNEL_MFMA)" + (", key shares + combine" if ws else "")
                                  + (", paired causal tiles where the launcher chooses them" if caus else ""),
                                  frac_f32_mfma_peak=round(tf2 / PEAK_TFLOPS["f32"], 4))
        ent["ms"], ent["tflops"] = ent["auto"]["ms"], ent["auto"]["tflops"]
        ent["default_arithmetic"] = "auto (FA_KERNEL_AUTO): " + AUTO_ARITH
        # the figure to quote for "fp32" in the reference's sense: fp32 ARITHMETIC (v_mfma_f32_32x32x2_f32), against the fp32 MFMA peak
        ent["reference_arithmetic"] = {"ms": ent["exact"]["ms"], "tflops": ent["exact"]["tflops"],
                                       "frac_f32_mfma_peak": ent["exact"]["frac_f32_mfma_peak"], "kernel": "fa_fwd_f32_kernel (kernel=\"exact\")"}
        ex[name] = ent

    for name in ("c3", "c2"):
        B2, H2, d2, n2, dt2, _ = WORKLOADS[name]
        q2, k2, v2 = make_inputs(B2 * H2, n2, d2, dt2, device, seed=1)
        f32_pair(name, f"B={B2} H={H2} d={d2} N={n2} {dt2}", q2, k2, v2, causal, 30 if name == "c3" else 100, 4 if name == "c3" else 20)
        if name == "c3" and not causal:
            f32_pair("c3_causal", f"B={B2} H={H2} d={d2} N={n2} {dt2}, causal", q2, k2, v2, True, 30, 6)
        del q2, k2, v2
    # fp32 tensors at the other head dims and on one slab (exact arithmetic over key shares): B=2 H=8 N=8192
    for name, (bh2, n2, d2, caus) in (("f32_d128", (16, 8192, 128, False)), ("f32_d128_causal", (16, 8192, 128, True)), ("f32_d32", (16, 8192, 32, False)),
                                      ("f32_bh1", (1, 8192, 64, False)), ("f32_bh1_causal", (1, 8192, 64, True))):
        q2, k2, v2 = make_inputs(bh2, n2, d2, "f32", device, seed=6)
        f32_pair(name, f"BH={bh2} d={d2} N={n2} f32" + (", causal" if caus else ""), q2, k2, v2, caus, 20, 4 if bh2 > 1 else 20)
        if bh2 == 1:
            try:
                ms1 = time_stats(fa, (q2, k2, v2), reps=3, causal=caus, scale=args.scale, kernel="exact:1", warmup=10, iters=10)["median"]
                ex[name]["exact"]["unsplit_ms"] = round(ms1, 4)
                ex[name]["exact"]["speedup_of_key_shares"] = round(ms1 / ex[name]["exact"]["ms"], 2)
            except Exception as e:  # pragma: no cover - informational only
                ex[name]["exact"]["unsplit_ms"] = repr(e)
        del q2, k2, v2
    # head dims outside {32, 64, 128} (round 6): the reference compiles any d % 32 == 0 (flashattention.cu:15,164); fp32 tensors through the plain
    # call run the exact fp32 MFMA kernel there (fa_fwd_f32_wide.hip) -- d = 96 and 256 at B=2 H=8 N=8192, validated like every other figure
    for d2 in (96, 256):
        try:
            q2, k2, v2 = make_inputs(16, 8192, d2, "f32", device, seed=8)
            o2 = torch.empty_like(q2)
            st2 = time_stats(fa, (q2, k2, v2), reps=3, causal=False, scale=args.scale, kernel="auto", warmup=4, iters=3, out=o2)
            chk2 = validate_output(fa, q2, k2, v2, o2, False, args.scale, TOLERANCE["f32"], f"f32_d{d2}")
            tf2 = fwd_flop(16, 8192, d2, False) / (st2["median"] * 1e-3) / 1e12
            ex[f"f32_d{d2}"] = {"workload": f"BH=16 d={d2} N=8192 f32", "ms": round(st2["median"], 4), "tflops": round(tf2, 2),
                                "frac_f32_mfma_peak": round(tf2 / PEAK_TFLOPS["f32"], 4), "max_abs_err": chk2["max_abs_err"], "tolerance": chk2["tolerance"],
                                "kernel": _cabi.lib().fa_kernel_name_for(_cabi.FA_DTYPE_F32, d2, 0, 16, 8192).decode(),
                                "what": "FA_KERNEL_AUTO at a head dim outside {32, 64, 128}: v_mfma_f32_32x32x2_f32 for both contractions (fp32 arithmetic)"}
            del q2, k2, v2, o2
        except Exception as e:  # pragma: no cover - informational only
            ex[f"f32_d{d2}"] = {"error": repr(e)}
        try:   # the same call on bf16 tensors: the same kernel, the tensors widened on load (until the end of round 6: the rung-0 kernel, 75x slower)
            q2, k2, v2 = make_inputs(16, 8192, d2, "bf16", device, seed=8)
            o2 = torch.empty_like(q2)
            st2 = time_stats(fa, (q2, k2, v2), reps=3, causal=False, scale=args.scale, kernel="auto", warmup=4, iters=3, out=o2)
            chk2 = validate_output(fa, q2, k2, v2, o2, False, args.scale, TOLERANCE["bf16_p_bf16_out"], f"bf16_d{d2}")
            tf2 = fwd_flop(16, 8192, d2, False) / (st2["median"] * 1e-3) / 1e12
            ex[f"bf16_d{d2}"] = {"workload": f"BH=16 d={d2} N=8192 bf16", "ms": round(st2["median"], 4), "tflops": round(tf2, 2),
                                 "frac_f32_mfma_peak": round(tf2 / PEAK_TFLOPS["f32"], 4), "max_abs_err": chk2["max_abs_err"], "tolerance": chk2["tolerance"],
                                 "kernel": _cabi.lib().fa_kernel_name_for(_cabi.FA_DTYPE_BF16, d2, 0, 16, 8192).decode(),
                                 "what": "bf16 tensors at a head dim outside {32, 64, 128}: widened on load into the exact kernel's fp32 LDS images, fp32 arithmetic, bf16 output"}
            del q2, k2, v2, o2
        except Exception as e:  # pragma: no cover - informational only
            ex[f"bf16_d{d2}"] = {"error": repr(e)}
    # llm.c harness size (attention_forward.cu:1217-1220): B=6 T=4096 C=768 NH=12, packed (B, T, 3C) fp32, causal, 1/sqrt(hs); mean of
    # 100 launches like benchmark_kernel (:1279-1288)
    try:
        B6, T6, C6, NH6 = 6, 4096, 768, 12
        inp = torch.rand(B6, T6, 3 * C6, device=device) * 2.0 - 1.0
        for _ in range(10):
            fa.forward_packed_qkv(inp, NH6)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fa.forward_packed_qkv(inp, NH6)
        e1.record()
        e1.synchronize()
        ms6 = e0.elapsed_time(e1) / 100.0
        fl6 = fwd_flop(B6 * NH6, T6, C6 // NH6, True)
        ex["llmc_packed_qkv"] = {"workload": f"B={B6} T={T6} C={C6} NH={NH6} fp32 packed QKV, causal, scale 1/sqrt(64)", "ms": round(ms6, 4),
                                 "tflops": round(fl6 / (ms6 * 1e-3) / 1e12, 2), "route": fa.last_forward_route(),
                                 "what": "fa_forward_packed_qkv (replaces attention_forward6 incl. its permute / unpermute kernels), mean of 100 launches"}
        del inp
    except Exception as e:  # pragma: no cover - informational only
        ex["llmc_packed_qkv"] = {"error": repr(e)}
    torch.cuda.synchronize()
    return ex


def stub_main(args, rank: int, world: int):
    """--cpu-stub: the launcher / sharding / timing protocol on CPU ranks (gloo), the kernel replaced by a sleep proportional to
    the rank's shard.  Not a measurement -- covers `python bench.py --gpus N` end to end where there is no GPU."""
    import flashattention_c_amd as fa
    dist = init_dist(world, "gloo") if world > 1 else None
    B, H, d, n, dtype, scaling = WORKLOADS[args.workload]
    total_bh = B * H
    if scaling == "strong":
        b0, b1 = fa.shard_range(total_bh, world, rank)
        bh, global_bh = b1 - b0, total_bh
    else:
        bh, global_bh = total_bh, total_bh * world
    step = lambda: time.sleep(1e-5 * bh)  # noqa: E731
    dt = timed_region(step, args.steps, args.warmup, lambda: None, world, dist, None)
    b5 = fa.shard_range(1024, world, rank)
    sizes = [b5[1] - b5[0]]
    if world > 1:
        import torch
        t = torch.tensor(sizes, dtype=torch.int64)
        dist.all_reduce(t)
        sizes = [int(t.item())]
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "STUB (no GPU, no measurement): launcher / sharding / timing protocol only", "value": None, "unit": "TFLOP/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "none (cpu stub)",
                          "config": {"workload": f"{args.workload} (cpu stub)", "global_bh": global_bh, "bh_per_gpu": bh},
                          "roofline": None, "cpu_baseline": None,
                          "extra": {"c5": {"slabs_covered": sizes[0], "n_gpus": world},
                                    "per_rank_ms": [round(x / args.steps * 1e3, 4) for x in PER_RANK_S]}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
