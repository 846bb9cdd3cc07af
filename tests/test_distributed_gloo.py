"""CPU, world_size 2 over gloo: the N > 1 path of bench.py (barrier + max-over-ranks timing) and the batch*head
sharding.  The attention itself is evaluated by the CPU oracle here -- this test is about the partition and the timing
protocol, which are identical on RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, tmpdir: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), OMP_NUM_THREADS="2")
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import flashattention_c_amd as fa
    from oracle import oracle as orc

    assert bench.dist_env() == (rank, world, rank)
    d_ = bench.init_dist(world, "gloo")

    # 1. sharded evaluation == whole evaluation (no exchange of K/V between ranks is needed)
    rng = np.random.default_rng(1)
    bh, n, d = 5, 48, 32
    q, k, v = (rng.standard_normal((bh, n, d)).astype(np.float32) for _ in range(3))
    b0, b1 = fa.shard_range(bh, world, rank)
    mine = orc.attention_f32(q[b0:b1], k[b0:b1], v[b0:b1], causal=True)
    sizes = fa.shard_sizes(bh, world)
    gathered = [torch.zeros(sizes[r], n, d) for r in range(world)]
    # ragged all_gather via per-rank broadcast (gloo): the only collective, and it is OFF the data path
    for r in range(world):
        buf = torch.from_numpy(mine) if r == rank else gathered[r]
        d_.broadcast(buf, src=r)
        gathered[r] = buf
    whole = orc.attention_f32(q, k, v, causal=True)
    assert np.array_equal(torch.cat(gathered).numpy(), whole)

    # 2. the timing protocol: every rank reports the MAX over ranks
    import time
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.02 * (rank + 1))  # rank 1 is slower

    dt = bench.timed_region(step, steps=3, warmup=1, sync_fn=lambda: None, world=world, dist=d_)
    assert len(calls) == 4
    assert dt >= 3 * 0.02 * world * 0.95, dt   # the slow rank's time, on every rank
    t = torch.tensor([dt], dtype=torch.float64)
    lo, hi = t.clone(), t.clone()
    d_.all_reduce(lo, op=d_.ReduceOp.MIN)
    d_.all_reduce(hi, op=d_.ReduceOp.MAX)
    assert lo.item() == hi.item()
    with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
        f.write("ok")
    d_.destroy_process_group()


def test_two_rank_shard_and_timing(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_bench_self_launches_its_ranks_when_typed_directly():
    """`python3 bench.py --gpus 2` typed without torch.distributed.run (the way the driver invokes N = 1) has to start its own
    ranks as child processes and relay rank 0's JSON line.  Here: CPU ranks over gloo with the kernel stubbed out (--cpu-stub),
    which exercises exactly that launcher path, the c5 shard arithmetic and the barrier / max-over-ranks protocol."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    for workload, bh_per_gpu, global_bh in (("c4", 16, 32), ("c5", 512, 1024)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-stub",
                            "--workload", workload], capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout           # ONE JSON line, from rank 0
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
        assert line["config"]["bh_per_gpu"] == bh_per_gpu and line["config"]["global_bh"] == global_bh
        assert line["extra"]["c5"]["slabs_covered"] == 1024       # the two ranks' c5 shards cover B*H = 64*16 exactly once
        assert line["scaling"] == ("weak" if workload == "c4" else "strong")
        assert len(line["extra"]["per_rank_ms"]) == 2 and all(t > 0 for t in line["extra"]["per_rank_ms"])   # every rank's own time (all-gather)
        assert abs(max(line["extra"]["per_rank_ms"]) - line["ms_per_step"]) < 1e-3                          # the reported time is the slowest rank's


def test_bench_under_torchrun_is_one_rank_not_a_launcher():
    """Under torch.distributed.run (RANK / WORLD_SIZE in the environment) bench.py must NOT launch again."""
    import bench
    os.environ.update(RANK="0", WORLD_SIZE="2")
    try:
        assert bench.under_launcher()
    finally:
        os.environ.pop("RANK"), os.environ.pop("WORLD_SIZE")
    assert not bench.under_launcher()
