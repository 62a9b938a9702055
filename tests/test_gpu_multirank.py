"""The N > 1 control flow on ONE MI355X (the reference has no multi-process path):

* `bench.py --gpus 2 --dist-backend gloo` as a child process -- launcher -> rendezvous -> shard -> encode / decode ->
  compact -> gather -> configs[3] strong-scaling block -> one JSON line.  Both ranks share cuda:0 (the launcher
  starts fresh children before any GPU call; 2 processes on the card stay far below the box's limit of 6).
* the gather functions of muahuff.dist with DEVICE tensors on a world-size-1 `nccl` (= RCCL) group: the device-side
  size exchange, the slicing and the root's own copy, i.e. everything RCCL sees except a second peer.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


@pytest.fixture(scope="module")
def mh():
    import muahuff
    from muahuff import codec, sclv, synth  # noqa: F401
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return muahuff


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_on_one_gpu_shard_encode_gather_and_verify():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    args = ["--gpus", "2", "--dist-backend", "gloo", "--channels-per-gpu", "8", "--bins", "200000", "--steps", "2",
            "--warmup", "1", "--no-per-S", "--no-small-shape", "--no-cpu-baseline", "--verify",
            "--total-channels", "13", "--configs3-steps", "2"]
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 prints THE line, nobody else prints one
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dist_backend"] == "gloo"
    assert line["scaling"] == "weak" and line["verified_roundtrip"] is True
    assert len(line["rank_devices"]) == 2 and len(line["rank_payload_words"]) == 2
    g = line["gather"]
    assert "error" not in g and g["root_shard_intact"] is True
    # the gather moved exactly what the two encoders wrote
    assert g["words_per_rank"] == line["rank_payload_words"]
    assert g["bytes_total"] == 4 * sum(line["rank_payload_words"])
    assert g["pipelined"]["bytes_total"] == g["bytes_total"] and g["pipelined"]["blocks"] == 4
    # whole-job value: both ranks' samples
    win = 200000 - 64
    assert abs(line["value"] - 2 * 8 * win * 2 / (line["ms_per_step"] * 2 * 1e-3) / 1e6) / line["value"] < 1e-6
    # configs[3]: one fixed channel set sharded 7 + 6, strong scaling
    c3 = line["configs3"]
    assert c3["scaling"] == "strong" and c3["total_channels"] == 13 and c3["channels_rank0"] == 7
    assert c3["verified_roundtrip"] is True
    assert abs(c3["MSamples_s"] - 13 * win * 2 / (c3["ms_per_step"] * 2 * 1e-3) / 1e6) / c3["MSamples_s"] < 1e-6


def test_gathers_with_device_tensors_on_a_one_rank_rccl_group(mh):
    """dist.gather_metadata / gather_payload / gather_payload_pipelined fed CUDA tensors on backend "nccl"."""
    import torch.distributed as dist
    from muahuff import dist as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cs = mh.synth.generate(12, 150_000, seed=3)
        tab = mh.sclv.table(5)
        plan = mh.codec.Plan(cs.ch_off, cs.ch_len, 5, 6, 1, mh.WIN_AFTER_CAL, tab)
        enc = plan.encode(cs.data)
        dense, tot = plan.compact(enc)
        total = int(tot.item())
        assert dense.payload.is_cuda
        meta = mdist.gather_metadata({"ch_bits": enc.ch_bits, "peak": enc.peak, "seg_words": enc.seg_words})
        assert meta["ch_bits"][0].is_cuda and torch.equal(meta["ch_bits"][0], enc.ch_bits)
        assert torch.equal(meta["peak"][0], enc.peak) and torch.equal(meta["seg_words"][0], enc.seg_words)
        pay, offs = mdist.gather_payload(dense.payload, total, dst=0)
        assert pay.is_cuda and list(offs) == [0, total] and torch.equal(pay[:total], dense.payload[:total])
        # pipelined: four channel blocks, each with its own plan and buffers; concatenation == the one-plan stream
        plans, items = [], []
        for k in range(4):
            lo, hi = mdist.shard_channels(12, 4, k)
            pb = mh.codec.Plan(cs.ch_off[lo:hi], cs.ch_len[lo:hi], 5, 6, 1, mh.WIN_AFTER_CAL, tab, seg_chunks=plan.seg_chunks)
            plans.append(pb)
            eb = pb.encode(cs.data)
            db, tb = pb.compact(eb)
            items.append((db.payload, tb))
        pay2, offs2 = mdist.gather_payload_pipelined(iter(items), dst=0)
        assert pay2.is_cuda and int(offs2[-1, 0]) == total
        assert torch.equal(pay2[:total], dense.payload[:total])
        torch.cuda.synchronize()
        for pb in plans:
            pb.close()
        plan.close()
    finally:
        dist.destroy_process_group()
