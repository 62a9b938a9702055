"""Wire / file format of a compressed channel set (format revision 3; revision 2 is still read), and the two calls that
make the codec usable end to end: compress() and decompress().

The reference never serialises a bitstream; this container is the
build's own.  File layout (little-endian):

    b"MUAHUFF1" | u32 header_len | header (UTF-8 JSON) | arrays, each padded to 8 bytes:
        ch_len u64[C] | peak u8[C] | enc u8[C] | skipped u8[C] | ch_bits u64[C]
        | seg_words u64[n_segments] | payload u32[total_words]

The header carries everything a decoder needs to rebuild the plan: format revision, chunk
geometry, S, h, mapper, window rule, seg_chunks and the K SCLV rows (the static codebooks).
Per channel the stream only needs (peak, enc) -- the 3+2-bit RAM word of the reference's RTL
(FPGA implementation/RAM.v:4) -- because codebooks are static and the symbol permutation is a
closed form of the calibration peak.  `payload` is the dense concatenation of all segments in
directory order (mh_compact); segment s starts at word sum(seg_words[:s]).
"""
import io
import json
import struct
from dataclasses import dataclass

import numpy as np

MAGIC = b"MUAHUFF1"
FORMAT_REVISION = 3          # written; revision 2 (no head segments, include/muahuff.h MH_WIN_REV2_SEGMENTS) is still read
READ_REVISIONS = (2, 3)
WIN_REV2_SEGMENTS = 0x100


@dataclass
class Compressed:
    header: dict
    ch_len: np.ndarray     # uint64 [C]
    peak: np.ndarray       # uint8  [C]
    enc: np.ndarray        # uint8  [C]
    skipped: np.ndarray    # uint8  [C]
    ch_bits: np.ndarray    # uint64 [C]  exact code bits (== reference histogram . SCLV)
    seg_words: np.ndarray  # uint64 [n_segments]
    payload: np.ndarray    # uint32 [sum(seg_words)]

    @property
    def payload_bits(self):
        return int(self.ch_bits.sum())

    @property
    def container_bits(self):
        return int(self.payload.size) * 32

    def tobytes(self):
        buf = io.BytesIO()
        write(buf, self)
        return buf.getvalue()


def _arrays(c):
    return [("ch_len", c.ch_len, np.uint64), ("peak", c.peak, np.uint8), ("enc", c.enc, np.uint8),
            ("skipped", c.skipped, np.uint8), ("ch_bits", c.ch_bits, np.uint64),
            ("seg_words", c.seg_words, np.uint64), ("payload", c.payload, np.uint32)]


def write(f, c):
    hdr = dict(c.header)
    hdr["sizes"] = {name: int(np.asarray(a).size) for name, a, _ in _arrays(c)}
    blob = json.dumps(hdr, sort_keys=True).encode()
    f.write(MAGIC)
    f.write(struct.pack("<I", len(blob)))
    f.write(blob)
    for _name, a, dt in _arrays(c):
        raw = np.ascontiguousarray(a, dtype=dt).tobytes()
        f.write(raw)
        f.write(b"\0" * (-len(raw) % 8))


def read(f):
    if f.read(8) != MAGIC:
        raise ValueError("not a MUAHUFF1 container")
    (n,) = struct.unpack("<I", f.read(4))
    hdr = json.loads(f.read(n).decode())
    if hdr.get("format_revision") not in READ_REVISIONS:
        raise ValueError("unsupported container revision %r" % hdr.get("format_revision"))
    out = {}
    for name, dt in (("ch_len", np.uint64), ("peak", np.uint8), ("enc", np.uint8), ("skipped", np.uint8),
                     ("ch_bits", np.uint64), ("seg_words", np.uint64), ("payload", np.uint32)):
        cnt = hdr["sizes"][name]
        nbytes = cnt * np.dtype(dt).itemsize
        raw = f.read(nbytes)
        if len(raw) != nbytes:
            raise ValueError("truncated container (%s)" % name)
        f.read(-nbytes % 8)
        out[name] = np.frombuffer(raw, dtype=dt).copy()
    return Compressed(hdr, **out)


def save(path, c):
    with open(path, "wb") as f:
        write(f, c)


def load(path):
    with open(path, "rb") as f:
        return read(f)


def make_header(S, h, mode, window, seg_chunks, sclv):
    from . import _lib
    sclv = np.asarray(sclv, dtype=np.uint8).reshape(-1, int(S))
    return {"format_revision": FORMAT_REVISION, "piece": _lib.PIECE, "lanes": _lib.LANES, "rows": _lib.ROWS,
            "S": int(S), "h": int(h), "mode": int(mode), "window": int(window), "seg_chunks": int(seg_chunks),
            "K": int(sclv.shape[0]), "sclv": [[int(v) for v in r] for r in sclv]}


# ---- GPU end-to-end ----------------------------------------------------------------------------
def compress(cs, S, h, mode, sclv, window=None, seg_chunks=0):
    """Calibrate + encode every channel of a ChannelSet on the GPU and bring the dense stream to
    the host as a `Compressed`.  seg_chunks = 0: the planner's choice; the header records it."""
    import torch

    from . import WIN_AFTER_CAL, codec
    window = WIN_AFTER_CAL if window is None else window
    plan = codec.Plan(cs.ch_off, cs.ch_len, S, h, mode, window, sclv, seg_chunks=seg_chunks)
    enc = plan.encode(cs.data)
    dense, tot = plan.compact(enc)
    torch.cuda.synchronize()
    total = int(tot.item())
    c = Compressed(make_header(S, h, mode, window, plan.seg_chunks, sclv), cs.ch_len.copy(),
                   enc.peak.cpu().numpy(), enc.enc.cpu().numpy(), enc.skipped.cpu().numpy(),
                   enc.ch_bits.cpu().numpy().astype(np.uint64),
                   enc.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments],
                   dense.payload[:total].cpu().numpy().view(np.uint32).copy())
    plan.close()
    return c


def plan_window(hd):
    """The `window` argument a plan for this container takes: the header's window rule, plus the flag that selects
    revision 2's segment directory when a revision-2 stream is read."""
    return int(hd["window"]) | (WIN_REV2_SEGMENTS if int(hd.get("format_revision", FORMAT_REVISION)) == 2 else 0)


def segments_per_channel(ch_len, h, window, seg_chunks, revision=FORMAT_REVISION):
    """Number of directory entries of each channel: the window rule of include/muahuff.h applied
    to the channel length, cut into segments of seg_chunks chunks (the planner's layout) -- from revision 3 on
    behind a head segment up to the next multiple of 128 samples when the window has at least 16 chunks."""
    from . import CHUNK
    seg = int(seg_chunks) * CHUNK
    n = window_lengths(ch_len, h, window)
    if int(revision) == 2:
        return (n + seg - 1) // seg
    T = np.asarray(ch_len, dtype=np.int64)
    w0 = np.zeros_like(T) if window == 3 else np.minimum(np.int64(1) << int(h), T)
    head = np.where((n >= 16 * CHUNK) & (w0 % 128 != 0), 128 - w0 % 128, 0)
    return (n - head + seg - 1) // seg + (head > 0)


def window_lengths(ch_len, h, window):
    """Samples in the encoded window of each channel (the rules of include/muahuff.h)."""
    from . import WIN_AFTER_CAL, WIN_FULL, WIN_REF_HALF, WIN_REF_HALF_TRUNC
    T = np.asarray(ch_len, dtype=np.int64)
    c = np.minimum(np.int64(1) << int(h), T)
    e = c + T // 2
    if window == WIN_REF_HALF:
        return np.where(e > T, 0, e - c)
    if window == WIN_REF_HALF_TRUNC:
        return np.minimum(e, T) - c
    if window == WIN_AFTER_CAL:
        return T - c
    if window == WIN_FULL:
        return T
    raise ValueError("unknown window rule %r" % (window,))


def validate(c):
    """Structural check of a container before it goes to the GPU.  The header fields are range-
    checked here; the walk over every chunk header of every segment -- header sizes, sub-stream
    lengths possible for the channel's code, chunk sizes adding up exactly to the directory and the
    payload -- is mh_validate_stream (host-only C, no GPU needed).  mh_decode itself never reads
    outside the payload it is given, so this check is about detecting corruption, not about
    memory safety.  Raises ValueError."""
    import ctypes as ct

    from . import _lib
    hd = c.header
    try:
        S, K, h, window, seg_chunks, mode = (int(hd[k]) for k in ("S", "K", "h", "window", "seg_chunks", "mode"))
        sclv = np.ascontiguousarray(np.array(hd["sclv"], np.int64).reshape(K, S))
    except (KeyError, TypeError, ValueError) as e:
        raise ValueError("container header: %r" % (e,))
    if (hd.get("piece"), hd.get("lanes"), hd.get("rows")) != (_lib.PIECE, _lib.LANES, _lib.ROWS):
        raise ValueError("container header: chunk geometry differs from this library's")
    if not (2 <= S <= 10) or not (1 <= K <= 255) or sclv.min() < 1 or sclv.max() > 9:
        raise ValueError("container header: S / K / code lengths out of range")
    if not (0 <= h <= 30) or not (0 <= window <= 3) or not (0 <= mode <= 1) or not (1 <= seg_chunks <= 0xFFFFFFFF // (_lib.PIECE * _lib.LANES * _lib.ROWS)):
        raise ValueError("container header: h / window / mode / seg_chunks out of range")
    C = len(c.ch_len)
    if not (len(c.peak) == len(c.enc) == len(c.skipped) == len(c.ch_bits) == C):
        raise ValueError("container arrays disagree about the channel count")
    if C == 0:
        if len(c.seg_words) or c.payload.size:
            raise ValueError("container directory does not match its header")
        return
    if int(c.ch_len.min()) == 0:
        raise ValueError("container holds an empty channel")
    ch_len = np.ascontiguousarray(c.ch_len, np.uint64)
    rows = np.ascontiguousarray(sclv, np.uint8)
    pay = np.ascontiguousarray(c.payload, np.uint32)
    segw = np.ascontiguousarray(c.seg_words, np.uint64)
    peak, enc = np.ascontiguousarray(c.peak, np.uint8), np.ascontiguousarray(c.enc, np.uint8)
    if hd.get("format_revision") not in READ_REVISIONS:
        raise ValueError("container header: unsupported revision %r" % (hd.get("format_revision"),))
    rc = _lib.lib().mh_validate_stream(ch_len.ctypes.data, C, S, h, mode, plan_window(hd), rows.ctypes.data, K, seg_chunks,
                                       pay.ctypes.data if pay.size else np.zeros(1, np.uint32).ctypes.data, pay.size,
                                       segw.ctypes.data if segw.size else np.zeros(1, np.uint64).ctypes.data, segw.size,
                                       peak.ctypes.data, enc.ctypes.data)
    if rc != 0:
        raise ValueError("corrupt container: " + _lib.lib().mh_last_error().decode(errors="replace"))


def decompress(c, device="cuda", channels=None, check=True):
    """Inverse of compress(): a ChannelSet whose windows hold min(x, S-1) (bytes outside the
    encoded windows are zero).  channels: optional list of channel indices -- only their segments
    are uploaded and decoded (the directory gives random access per channel); the returned set
    holds them in the order given.  check=True runs validate() first (containers from disk are
    untrusted input for a kernel that follows their headers)."""
    import torch

    from . import codec
    from .container import ChannelSet
    if check:
        validate(c)
    hd = c.header
    nseg_ch = segments_per_channel(c.ch_len, hd["h"], hd["window"], hd["seg_chunks"], hd.get("format_revision", FORMAT_REVISION))
    if int(nseg_ch.sum()) != len(c.seg_words):
        raise ValueError("container directory does not match its header")
    seg_words, payload, peak, enc, skipped, ch_bits, ch_len = (c.seg_words, c.payload, c.peak, c.enc, c.skipped,
                                                                c.ch_bits, c.ch_len)
    if channels is not None:
        sel = np.asarray(channels, dtype=np.int64)
        if sel.size and (sel.min() < 0 or sel.max() >= len(c.ch_len)):
            raise IndexError("channel index out of range")
        first = np.concatenate([[0], np.cumsum(nseg_ch)]).astype(np.int64)       # channel -> first segment
        off = np.concatenate([[0], np.cumsum(c.seg_words)]).astype(np.int64)    # segment -> first word
        segs = np.concatenate([np.arange(first[i], first[i + 1]) for i in sel]) if sel.size else np.zeros(0, np.int64)
        payload = (np.concatenate([c.payload[off[s]:off[s + 1]] for s in segs]) if segs.size
                   else np.zeros(0, np.uint32))
        seg_words = c.seg_words[segs]
        peak, enc, skipped, ch_bits, ch_len = c.peak[sel], c.enc[sel], c.skipped[sel], c.ch_bits[sel], c.ch_len[sel]
    cs = ChannelSet.empty([int(n) for n in ch_len], device=device)
    if len(ch_len) == 0:
        return cs
    plan = codec.Plan(cs.ch_off, cs.ch_len, hd["S"], hd["h"], hd["mode"], plan_window(hd),
                      np.array(hd["sclv"], np.uint8), seg_chunks=hd["seg_chunks"])
    if plan.n_segments != len(seg_words):
        raise ValueError("container directory does not match its header")
    dev = cs.data.device
    pay = torch.zeros(payload.size + 4, dtype=torch.int32, device=dev)
    pay[:payload.size] = torch.from_numpy(np.ascontiguousarray(payload).view(np.int32)).to(dev)
    seg_off = np.concatenate([[0], np.cumsum(seg_words)[:-1]]).astype(np.int64) if len(seg_words) else np.zeros(1, np.int64)
    e = codec.Encoded(pay, torch.from_numpy(seg_words.astype(np.int64)).to(dev),
                      torch.from_numpy(ch_bits.astype(np.int64)).to(dev), torch.from_numpy(np.ascontiguousarray(peak)).to(dev),
                      torch.from_numpy(np.ascontiguousarray(enc)).to(dev), torch.from_numpy(np.ascontiguousarray(skipped)).to(dev),
                      torch.from_numpy(seg_off).to(dev), True)
    plan.decode(e, cs.data)
    ok = plan.decode_ok()  # synchronises
    plan.close()
    if not ok:
        raise ValueError("corrupt container: a chunk header points outside the payload (decode abandoned)")
    return cs
