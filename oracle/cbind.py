"""ctypes binding of the C oracle (oracle/mh_oracle.c).  Test infrastructure only."""
import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmh_oracle.so")

PIECE, LANES, ROWS = 16, 64, 16
SUB = PIECE * ROWS
CHUNK = SUB * LANES
HDR_WORDS = LANES // 2

MODE_NOSORT, MODE_APPROX = 0, 1
WIN_REF_HALF, WIN_REF_HALF_TRUNC, WIN_AFTER_CAL, WIN_FULL = 0, 1, 2, 3
WIN_REV2_SEGMENTS = 0x100  # OR-ed into a window rule: format revision 2's segment directory (no head segments)


def build(force=False):
    src = os.path.join(_HERE, "mh_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libmh_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class _Params(ct.Structure):
    _fields_ = [("S", ct.c_uint32), ("h", ct.c_uint32), ("mode", ct.c_uint32),
                ("window", ct.c_uint32), ("K", ct.c_uint32), ("seg_chunks", ct.c_uint32),
                ("sclv", ct.c_void_p)]


class _Chan(ct.Structure):
    _fields_ = [("cutoff", ct.c_uint64), ("w0", ct.c_uint64), ("w1", ct.c_uint64),
                ("cal_sorted", ct.c_uint32 * 16), ("idx", ct.c_uint8 * 16),
                ("rank_of", ct.c_uint8 * 16), ("peak", ct.c_uint8), ("enc", ct.c_uint8),
                ("skipped", ct.c_uint8)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ct.CDLL(_SO)
        _lib.mho_online_cutoff_literal.restype = ct.c_uint64
        _lib.mho_online_cutoff_literal.argtypes = [ct.c_void_p, ct.c_uint64, ct.c_uint64, ct.c_int]
        _lib.mho_slot_words.restype = ct.c_uint64
        _lib.mho_slot_words.argtypes = [ct.c_uint64, ct.c_uint32]
        _lib.mho_plan_segments.restype = ct.c_uint64
        _lib.mho_encode_segment.restype = ct.c_uint64
        _lib.mho_decode_segment.restype = ct.c_uint64
    return _lib


def _p(a):
    return a.ctypes.data_as(ct.c_void_p) if a is not None else None


class Params:
    """S, h, mode, window, sclv[K,S] (uint8), seg_chunks."""

    def __init__(self, S, h, mode, window, sclv, seg_chunks=2):
        self.sclv = np.ascontiguousarray(np.asarray(sclv, dtype=np.uint8).reshape(-1, S))
        self.S, self.h, self.mode, self.window = int(S), int(h), int(mode), int(window)
        self.K = self.sclv.shape[0]
        self.seg_chunks = int(seg_chunks)
        self.c = _Params(self.S, self.h, self.mode, self.window, self.K, self.seg_chunks,
                         self.sclv.ctypes.data)

    @property
    def ref(self):
        return ct.byref(self.c)


def flatten(channels):
    """list of 1-D uint8 arrays -> (flat data, ch_off, ch_len); offsets padded to 16 B."""
    lens = np.array([len(x) for x in channels], dtype=np.uint64)
    pad = (lens + np.uint64(15)) & ~np.uint64(15)
    off = np.zeros(len(channels), dtype=np.uint64)
    if len(channels) > 1:
        off[1:] = np.cumsum(pad)[:-1]
    total = int(pad.sum()) if len(channels) else 0
    data = np.zeros(total + 64, dtype=np.uint8)
    for x, o in zip(channels, off):
        data[int(o):int(o) + len(x)] = x
    return data, off, lens


def online_cutoff_literal(data, cutoff, max_rate):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    r = lib().mho_online_cutoff_literal(_p(d), len(d), int(cutoff), int(max_rate))
    if r == 2 ** 64 - 1:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    data[...] = d
    return int(r)


def approx_sort_literal(hist):
    h = np.ascontiguousarray(hist, dtype=np.uint64)
    idx = np.zeros(16, dtype=np.uint8)
    lib().mho_approx_sort_literal(_p(h), len(h), _p(idx))
    return idx[:len(h)].astype(np.int64)


def approx_sort_rule(S, peak):
    idx = np.zeros(16, dtype=np.uint8)
    lib().mho_approx_sort_rule(int(S), int(peak), _p(idx))
    return idx[:S].astype(np.int64)


def codebook(row):
    row = np.ascontiguousarray(row, dtype=np.uint8)
    code = np.zeros(16, dtype=np.uint16)
    ln = np.zeros(16, dtype=np.uint8)
    rc = lib().mho_codebook(_p(row), len(row), _p(code), _p(ln))
    if rc != 0:
        raise ValueError("not a complete non-decreasing SCLV row (rc=%d)" % rc)
    return code[:len(row)].copy(), ln[:len(row)].copy()


def measure(data, ch_off, ch_len, params, nthreads=1):
    C, S = len(ch_off), params.S
    out = dict(cutoff=np.zeros(C, np.uint64), cal_sorted=np.zeros((C, S), np.uint32),
               peak=np.zeros(C, np.uint8), enc=np.zeros(C, np.uint8),
               post_mapped=np.zeros((C, S), np.uint64), bits=np.zeros(C, np.uint64),
               skipped=np.zeros(C, np.uint8))
    rc = lib().mho_measure(_p(data), _p(ch_off), _p(ch_len), ct.c_uint32(C), params.ref,
                           _p(out["cutoff"]), _p(out["cal_sorted"]), _p(out["peak"]),
                           _p(out["enc"]), _p(out["post_mapped"]), _p(out["bits"]),
                           _p(out["skipped"]), int(nthreads))
    if rc != 0:
        raise IndexError("empty channel")
    return out


def plan_segments(ch_len, params):
    ch_len = np.ascontiguousarray(ch_len, dtype=np.uint64)
    cap = ct.c_uint64(0)
    n = lib().mho_plan_segments(_p(ch_len), ct.c_uint32(len(ch_len)), params.ref, None, None,
                                None, None, ct.byref(cap))
    seg = dict(ch=np.zeros(n, np.uint32), first=np.zeros(n, np.uint64), n=np.zeros(n, np.uint64),
               off=np.zeros(n, np.uint64))
    lib().mho_plan_segments(_p(ch_len), ct.c_uint32(len(ch_len)), params.ref, _p(seg["ch"]),
                            _p(seg["first"]), _p(seg["n"]), _p(seg["off"]), ct.byref(cap))
    seg["cap_words"] = int(cap.value)
    return seg


def slot_words(n, maxlen):
    return int(lib().mho_slot_words(int(n), int(maxlen)))


def encode(data, ch_off, ch_len, params, nthreads=1):
    C = len(ch_off)
    seg = plan_segments(ch_len, params)
    out = dict(seg=seg, payload=np.zeros(seg["cap_words"] + 4, np.uint32),
               seg_words=np.zeros(len(seg["ch"]), np.uint64), ch_bits=np.zeros(C, np.uint64),
               peak=np.zeros(C, np.uint8), enc=np.zeros(C, np.uint8),
               skipped=np.zeros(C, np.uint8))
    rc = lib().mho_encode(_p(data), _p(ch_off), _p(ch_len), ct.c_uint32(C), params.ref,
                          _p(out["payload"]), ct.c_uint64(seg["cap_words"]), _p(out["seg_words"]),
                          _p(out["ch_bits"]), _p(out["peak"]), _p(out["enc"]), _p(out["skipped"]),
                          int(nthreads))
    if rc != 0:
        raise RuntimeError("oracle encode failed rc=%d" % rc)
    return out


def decode(payload, ch_off, ch_len, params, peak, enc, out_size, nthreads=1):
    out = np.zeros(out_size, np.uint8)
    payload = np.ascontiguousarray(payload, dtype=np.uint32)
    lib().mho_decode(_p(payload), _p(ch_off), _p(ch_len), ct.c_uint32(len(ch_off)), params.ref,
                     _p(np.ascontiguousarray(peak, np.uint8)),
                     _p(np.ascontiguousarray(enc, np.uint8)), _p(out), int(nthreads))
    return out


def synth(ch_off, ch_len, thr, seed, total=None, nthreads=1):
    ch_off = np.ascontiguousarray(ch_off, np.uint64)
    ch_len = np.ascontiguousarray(ch_len, np.uint64)
    thr = np.ascontiguousarray(thr, np.uint32)
    if total is None:
        total = int((ch_off + ch_len).max()) + 64 if len(ch_off) else 64
    data = np.zeros(total, np.uint8)
    lib().mho_synth(_p(data), _p(ch_off), _p(ch_len), ct.c_uint32(len(ch_off)), _p(thr),
                    ct.c_uint64(seed), int(nthreads))
    return data


def rebin_u32(x, r):
    x = np.ascontiguousarray(x, np.uint8)
    out = np.zeros((len(x) + r - 1) // r, np.uint32)
    lib().mho_rebin_u32(_p(x), ct.c_uint64(len(x)), ct.c_uint32(r), _p(out))
    return out


def rebin_u8(x, r):
    x = np.ascontiguousarray(x, np.uint8)
    out = np.zeros((len(x) + r - 1) // r, np.uint8)
    lib().mho_rebin_u8(_p(x), ct.c_uint64(len(x)), ct.c_uint32(r), _p(out))
    return out
