"""bench.py --gpus N starts its own N ranks (no GPU needed for these checks): the launcher
rendezvous self-test over gloo, the fail-fast paths and their non-zero exit codes."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_2_spawns_two_ranks_that_rendezvous():
    r = _run(["--gpus", "2", "--check-launch", "--dist-backend", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"check_launch": True, "n_gpus": 2, "ranks": 2}


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="would really launch on a multi-GPU box")
def test_more_ranks_than_gpus_fails_fast_and_loudly():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=120)
    assert r.returncode == 2
    assert "needs 2 visible GPUs" in r.stderr and not r.stdout.strip()


def test_world_size_mismatch_is_an_error_not_a_single_gpu_number():
    r = _run(["--gpus", "2", "--check-launch"], env_extra={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
    r = _run(["--gpus", "1", "--check-launch"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 2
