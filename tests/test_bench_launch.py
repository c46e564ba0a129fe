"""bench.py's own N-rank launch (CPU, gloo): `python bench.py --gpus N` without WORLD_SIZE starts N ranks through torch.distributed.run, as the
driver's launcher would, and relays rank 0's line and the exit code.  --launch-check stops after the rendezvous, a barrier and an all-gather of the
ranks, before anything touches a GPU (the real N > 1 bench runs on the GPU box: tests/test_gpu_multi_device.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_starts_its_own_ranks():
    r = _run(["--gpus", "2", "--backend", "gloo", "--one-device", "--no-cpu-baseline", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d == {"launch_check": True, "world": 2, "ranks": [0, 1], "distinct_processes": 2}


def test_a_failing_rank_ends_the_launch_non_zero():
    # WORLD_SIZE set by a launcher but different from --gpus: the rank refuses, and the self-launcher is not entered
    r = _run(["--gpus", "2", "--launch-check"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
