"""Multi-GPU: channels shard across ranks (one process per GPU); the only exchange is the
gather that concatenates the packed bitstreams on rank 0 (RCCL over xGMI; ``gloo`` on CPU in
tests).  Channels are independent (per-channel calibration, codebook choice and bitstream,
SURVEY.md section 8e), so measure / encode / decode need no collective at all.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_channels(C, world_size, rank):
    """Contiguous channel block [lo, hi) of this rank; the first C % world ranks get one more."""
    base, extra = divmod(int(C), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_metadata(tensors, dst=0, group=None):
    """Gather small fixed-size per-channel arrays (bit lengths, peak, enc ...) of every rank on
    ``dst``.  tensors: dict name -> tensor (same dtype per name on all ranks; lengths may
    differ).  Returns dict name -> list of per-rank tensors on dst, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = {}
    for name, t in tensors.items():
        n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        sizes = [int(s.item()) for s in sizes]
        m = max(sizes) if sizes else 0
        pad = torch.zeros(m, dtype=t.dtype, device=t.device)
        pad[:t.numel()] = t.reshape(-1)
        bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst, group=group)
        out[name] = [b[:s] for b, s in zip(bufs, sizes)] if rank == dst else None
    return out


def gather_payload(dense, total_words, dst=0, group=None):
    """Concatenate the dense payloads of all ranks on ``dst`` in rank (= channel) order.

    dense: int32 words of this rank (only the first total_words are sent).  Every peer sends
    its shard over its own xGMI link to the root concurrently (point-to-point, no ring), which
    is the cheapest pattern for a gather on a full mesh.  Returns (payload, word offsets per
    rank) on dst and (None, offsets) elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = torch.tensor([int(total_words)], dtype=torch.int64, device=dense.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    if rank == dst:
        out = torch.empty(int(offs[-1]) + 4, dtype=dense.dtype, device=dense.device)
        out[offs[rank]:offs[rank + 1]] = dense[:sizes[rank]]
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group) for r in range(world)
               if r != dst and sizes[r] > 0]
    else:
        out = None
        ops = [dist.P2POp(dist.isend, dense[:sizes[rank]], dst, group)] if sizes[rank] > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, offs


def gather_payload_pipelined(blocks, dst=0, group=None):
    """Gather behind the encoder: each rank encodes its channels in a few channel blocks and the
    packed stream of block b travels to ``dst`` while block b+1 is still being encoded
    (at ~0.19 B/sample the xGMI links need several times the kernel time,
    so the transfer should start as early as possible).

    blocks: iterable yielding ``(dense, total_words)`` per channel block in channel order --
    ``dense`` the compacted int32 words of the block, ``total_words`` an int or a 1-element
    tensor.  Producing an item should only ENQUEUE device work (encode + compact); this
    function pulls block b+1 from the iterable before it reads block b's size, so with RCCL
    (sends run on the communicator's own stream) the copy overlaps the next encode.  Every rank
    must yield the same number of blocks.  Because block b+1 is enqueued before block b's words
    and size are read, EVERY in-flight block needs its own ``dense`` and ``total_words`` buffers:
    a producer that recycles one buffer per shape (stream.StreamEncoder.encode_block_device) must
    clone them before yielding.

    Returns on dst ``(payload, offs)`` with the words in rank-major, block-minor (= channel)
    order and ``offs[r][b]`` the word offset of rank r's block b (plus a final total at
    ``offs[-1][0]``); ``(None, offs)`` elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    it = iter(blocks)
    cur = next(it, None)
    sizes_all, pieces, keep, works = [], [], [], []
    while cur is not None:
        nxt = next(it, None)  # enqueue the next block's device work first
        dense, tot = cur
        n_here = int(tot.item()) if torch.is_tensor(tot) else int(tot)
        n = torch.tensor([n_here], dtype=torch.int64, device=dense.device)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n, group=group)
        sizes = [int(x.item()) for x in sizes]
        sizes_all.append(sizes)
        ops = []
        if rank == dst:
            bufs = [dense[:n_here] if r == dst else torch.empty(sizes[r], dtype=dense.dtype, device=dense.device)
                    for r in range(world)]
            pieces.append(bufs)
            ops = [dist.P2POp(dist.irecv, bufs[r], r, group) for r in range(world) if r != dst and sizes[r] > 0]
        elif n_here > 0:
            ops = [dist.P2POp(dist.isend, dense[:n_here], dst, group)]
        keep.append(dense)  # the block's buffer must outlive its send
        if ops:
            works.extend(dist.batch_isend_irecv(ops))
        cur = nxt
    for w in works:
        w.wait()
    nb = len(sizes_all)
    offs = np.zeros((world + 1, max(nb, 1)), dtype=np.int64)
    run = 0
    for r in range(world):
        for b in range(nb):
            offs[r, b] = run
            run += sizes_all[b][r]
    offs[world, 0] = run
    if rank != dst:
        return None, offs
    if nb == 0:
        return torch.empty(0, dtype=torch.int32), offs
    out = torch.empty(run + 4, dtype=pieces[0][dst].dtype, device=pieces[0][dst].device)
    for r in range(world):
        for b in range(nb):
            out[offs[r, b]:offs[r, b] + sizes_all[b][r]] = pieces[b][r]
    return out, offs
