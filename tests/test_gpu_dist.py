"""GPU (-m gpu): the launcher + RCCL leg of bench.py's multi-GPU protocol, as far as one GPU can show it."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_timing_protocol_over_rccl_with_one_rank():
    """torch.distributed.run -> one rank -> init_process_group('nccl') -> bench.timed_region with its collectives forced on."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "nccl_one_rank_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0 and "NCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_under_the_drivers_launch_line_with_one_gpu():
    """The driver's N > 1 command line with N = 1: `python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`
    must print exactly one JSON line carrying the contract's keys."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["warmup"] == 2 and line["value"] > 100.0
    assert 0.2 < line["roofline"]["frac"] < 1.0 and line["roofline"]["bound"] == "mfma"


def test_bench_config_5_under_the_drivers_launch_line_with_one_gpu():
    """`--workload c5` (BASELINE config 5 as the headline: 1024 slabs split over the ranks, strong scaling) under the driver's launcher with
    one rank over RCCL: all 1024 slabs on the one GPU."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "c5", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras", "--backend", "nccl"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["config"]["global_bh"] == 1024 and line["config"]["bh_per_gpu"] == 1024
    assert line["value"] > 500.0 and line["roofline"]["kernel"] == "fa_fwd_bf16_x4_kernel"


def test_bench_with_two_ranks_on_the_one_gpu():
    """The world > 1 code of bench.py -- the attempts loop's broadcast, the all-gather of per-rank times, ranks_seen / dist_backend, the
    c5 split over two ranks, weak-scaling prediction -- with two ranks sharing cuda:0 (--same-device, a test hook) over gloo (RCCL refuses
    two ranks on one device).  Not a measurement; the first real multi-GPU run is the driver's."""
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--no-cpu-baseline", "--backend", "gloo", "--same-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    ex = line["extra"]
    assert line["n_gpus"] == 2 and ex["ranks_seen"] == 2 and ex["dist_backend"] == "gloo" and len(ex["per_rank_ms"]) == 2
    assert line["config"]["global_bh"] == 32 and line["scaling"] == "weak" and "weak_scaling" in ex
    assert ex["c5"]["n_gpus"] == 2 and ex["c5"]["bh_per_gpu"] == 512 and len(ex["c5"]["per_rank_ms"]) == 2
    assert 1 <= len(ex["timing_attempts"]) <= 3 and line["validation"]["status"] == "ok"
    assert line["roofline"]["frac"] > 0.1 and line["roofline"]["frac_from_ms_per_step"] > 0.05
