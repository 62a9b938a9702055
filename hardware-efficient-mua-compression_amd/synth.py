"""Synthetic Poisson-like MUA, generated on the GPU.

Rates per channel are log-uniform over [lo, hi] counts/bin, indexed by a fixed integer hash
of the channel number; they are turned ON THE HOST into integer inverse-CDF thresholds
thr[s] = floor(65536 * P(X <= s)), s < 15, which is all the device (and the CPU oracle) ever
sees -- so both produce identical bytes.
"""
import ctypes as ct
import math

import numpy as np
import torch

from . import _lib
from .codec import _ptr, _stream
from .container import ChannelSet


def channel_rates(C, lo=0.2, hi=3.0, first_channel=0):
    ch = np.arange(first_channel, first_channel + C, dtype=np.uint64)
    frac = ((ch * np.uint64(2654435761)) % np.uint64(2 ** 32)).astype(np.float64) / 2.0 ** 32
    return lo * (hi / lo) ** frac


def thresholds(rates):
    thr = np.zeros((len(rates), 15), dtype=np.uint32)
    for c, lam in enumerate(rates):
        cdf, term = 0.0, math.exp(-lam)
        for s in range(15):
            cdf += term
            thr[c, s] = min(65536, int(math.floor(65536.0 * cdf)))
            term *= lam / (s + 1)
    return thr


def generate(C, T, seed=0, lo=0.2, hi=3.0, first_channel=0, device=None):
    """C channels x T bins of synthetic MUA as a ChannelSet on the GPU."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    cs = ChannelSet.empty([T] * C, device=dev)
    thr = thresholds(channel_rates(C, lo, hi, first_channel))
    fill(cs, thr, seed, first_channel)
    return cs


def fill(cs, thr, seed, first_channel=0):
    dev = cs.device
    d_off = torch.from_numpy(cs.ch_off.astype(np.int64)).to(dev)
    d_len = torch.from_numpy(cs.ch_len.astype(np.int64)).to(dev)
    d_thr = torch.from_numpy(np.ascontiguousarray(thr, dtype=np.uint32).view(np.int32)).to(dev)
    # the generator hashes (seed, channel index within this set + first_channel folded in seed)
    _lib.check(_lib.lib().mh_synth_poisson(_ptr(cs.data), _ptr(d_off), _ptr(d_len), cs.C,
                                           int(cs.ch_len.max()) if cs.C else 0, _ptr(d_thr),
                                           ct.c_uint64(int(seed) + (int(first_channel) << 20)), _stream()))
    return cs
