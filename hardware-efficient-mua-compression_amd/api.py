"""Two calls that cover what a user of the reference usually wants.

bit_rates(...)   per-channel bit rate exactly as the reference's scripts compute it for one
                 design point (get_BR_with_approx_sort.py:164-193, 281-292): clip at S,
                 calibrate on the first 2^hist_bits bins, map (approx-sort or identity), pick
                 the best of the given static Huffman encoders, measure the next T/2 bins;
                 BR = 1000 / (BP / (bits / n)) in bits/s/channel, NaN for skipped channels.
compress(...)    the same design point, but actually emitting the bitstream (container_io).
"""
import numpy as np


def bit_rates(channels, S=3, hist_bits=6, approx=True, sclv_rows=None, BP=50):
    """channels: list of 1-D count arrays (or a ChannelSet).  Returns dict with BR (float64
    [C]), bits, n, enc, peak, skipped -- all host arrays."""
    import torch

    from . import MODE_APPROX, MODE_NOSORT, WIN_REF_HALF, codec, sclv
    from .container import ChannelSet
    cs = channels if isinstance(channels, ChannelSet) else ChannelSet.from_channels(channels)
    rows = sclv.table(S) if sclv_rows is None else np.asarray(sclv_rows, dtype=np.uint8).reshape(-1, S)
    plan = codec.Plan(cs.ch_off, cs.ch_len, S, hist_bits, MODE_APPROX if approx else MODE_NOSORT, WIN_REF_HALF, rows)
    m = plan.measure(cs.data)
    torch.cuda.synchronize()
    bits = m.bits.cpu().numpy().astype(np.float64)
    n = m.post_hist.sum(1).cpu().numpy().astype(np.float64)
    out = dict(BR=codec.bit_rate(bits, n, BP), bits=bits.astype(np.int64), n=n.astype(np.int64),
               enc=m.enc.cpu().numpy(), peak=m.peak.cpu().numpy(), skipped=m.skipped.cpu().numpy())
    plan.close()
    return out


def compress(channels, S=3, hist_bits=6, approx=True, sclv_rows=None, path=None):
    """Encode everything after the calibration window of every channel.  Returns a
    container_io.Compressed (and writes it to `path` when given)."""
    from . import MODE_APPROX, MODE_NOSORT, container_io, sclv
    from .container import ChannelSet
    cs = channels if isinstance(channels, ChannelSet) else ChannelSet.from_channels(channels)
    rows = sclv.table(S) if sclv_rows is None else np.asarray(sclv_rows, dtype=np.uint8).reshape(-1, S)
    c = container_io.compress(cs, S, hist_bits, MODE_APPROX if approx else MODE_NOSORT, rows)
    if path is not None:
        container_io.save(path, c)
    return c


def decompress(c_or_path, channels=None):
    """-> list of uint8 arrays: min(x, S-1) after the calibration window, zeros before it.
    channels: optional list of channel indices to decode (random access through the directory)."""
    from . import container_io
    c = container_io.load(c_or_path) if isinstance(c_or_path, (str, bytes)) or hasattr(c_or_path, "__fspath__") else c_or_path
    return container_io.decompress(c, channels=channels).to_channels()
