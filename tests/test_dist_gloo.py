"""N>1 path on CPU: world_size-2 gloo processes shard channels, encode their shard (here with
the oracle standing in for the GPU kernels, which a CPU box cannot run) and gather metadata +
packed payload on rank 0 with the product's muahuff.dist functions.  The gathered stream must
equal the single-process encoding of all channels, segment by segment."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _channels():
    rng = np.random.RandomState(4)
    lens = [40000, 16384, 70001, 5, 33000, 1000, 16385]
    return [np.minimum(rng.poisson(r, size=T), 255).astype(np.uint8)
            for T, r in zip(lens, [0.1, 0.5, 1.0, 2.0, 3.0, 0.3, 1.5])]


def _dense(oc, enc):
    seg = enc["seg"]
    parts = [enc["payload"][int(o):int(o) + int(n)] for o, n in zip(seg["off"], enc["seg_words"])]
    return np.concatenate(parts) if parts else np.zeros(0, np.uint32)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from muahuff import dist as mdist
    OC = oracle.c
    chans = _channels()
    lo, hi = mdist.shard_channels(len(chans), world, rank)
    mine = chans[lo:hi]
    tab = np.array([[1, 2, 3, 4, 4], [2, 2, 2, 3, 3], [1, 3, 3, 3, 3]], np.uint8)
    p = OC.Params(5, 6, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=2)
    data, off, ln = OC.flatten(mine)
    enc = OC.encode(data, off, ln, p)
    dense = _dense(OC, enc)
    meta = mdist.gather_metadata({"ch_bits": torch.from_numpy(enc["ch_bits"].astype(np.int64)),
                                  "peak": torch.from_numpy(enc["peak"]), "enc": torch.from_numpy(enc["enc"]),
                                  "seg_words": torch.from_numpy(enc["seg_words"].astype(np.int64))})
    pay, offs = mdist.gather_payload(torch.from_numpy(dense.view(np.int32)), len(dense))
    if rank == 0:
        q.put(dict(payload=pay[:int(offs[-1])].numpy().view(np.uint32).copy(), offs=offs,
                   ch_bits=torch.cat(meta["ch_bits"]).numpy(), peak=torch.cat(meta["peak"]).numpy(),
                   enc=torch.cat(meta["enc"]).numpy(), seg_words=torch.cat(meta["seg_words"]).numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_encode_gathers_to_single_process_stream(world):
    import oracle
    OC = oracle.c
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    got = q.get(timeout=120)
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    chans = _channels()
    tab = np.array([[1, 2, 3, 4, 4], [2, 2, 2, 3, 3], [1, 3, 3, 3, 3]], np.uint8)
    p = OC.Params(5, 6, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=2)
    data, off, ln = OC.flatten(chans)
    ref = OC.encode(data, off, ln, p)
    assert np.array_equal(got["ch_bits"], ref["ch_bits"].astype(np.int64))
    assert np.array_equal(got["peak"], ref["peak"]) and np.array_equal(got["enc"], ref["enc"])
    assert np.array_equal(got["seg_words"], ref["seg_words"].astype(np.int64))
    assert np.array_equal(got["payload"], _dense(OC, ref))  # rank order == channel order


def test_shard_channels_partition():
    from muahuff import dist as mdist
    for C in (1, 7, 8, 1024, 10000):
        for world in (1, 2, 3, 8):
            cuts = [mdist.shard_channels(C, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == C
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def _worker_pipelined(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from muahuff import dist as mdist
    OC = oracle.c
    chans = _channels()
    lo, hi = mdist.shard_channels(len(chans), world, rank)
    tab = np.array([[1, 2, 3, 4, 4], [2, 2, 2, 3, 3], [1, 3, 3, 3, 3]], np.uint8)
    p = OC.Params(5, 6, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=2)
    produced = []

    def blocks(nblocks=3):  # this rank's channels in 3 channel blocks (some may be empty)
        n = hi - lo
        for b in range(nblocks):
            b0, b1 = mdist.shard_channels(n, nblocks, b)
            produced.append(b)
            if b1 == b0:
                yield torch.zeros(0, dtype=torch.int32), 0
                continue
            data, off, ln = OC.flatten(chans[lo + b0:lo + b1])
            dense = _dense(OC, OC.encode(data, off, ln, p))
            yield torch.from_numpy(dense.view(np.int32).copy()), torch.tensor([len(dense)])

    pay, offs = mdist.gather_payload_pipelined(blocks())
    assert produced == [0, 1, 2]
    if rank == 0:
        q.put(dict(payload=pay[:int(offs[-1, 0])].numpy().view(np.uint32).copy(), offs=offs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_gather_equals_single_process_stream(world):
    """Channel blocks sent while later blocks are still being produced arrive in channel order."""
    import oracle
    OC = oracle.c
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    got = q.get(timeout=120)
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    chans = _channels()
    tab = np.array([[1, 2, 3, 4, 4], [2, 2, 2, 3, 3], [1, 3, 3, 3, 3]], np.uint8)
    p = OC.Params(5, 6, 1, OC.WIN_AFTER_CAL, tab, seg_chunks=2)
    data, off, ln = OC.flatten(chans)
    want = _dense(OC, OC.encode(data, off, ln, p))  # segments never span channels: blocks concatenate exactly
    assert np.array_equal(got["payload"], want)
    assert got["offs"][-1, 0] == len(want)
