"""Drop-in for the reference's helper module ``Compressing data/functions_1.py``.

Same three functions, same signatures, return types and in-place side effects, so that
``from functions_1 import *`` in reference-shaped scripts keeps working (it also leaks ``np``
and ``math`` like the original).  ``approx_sort`` and
``online_histogram_w_sat_based_nb_of_samples`` are O(S) / closed-form host logic;
``bin_MUA_data`` runs on the GPU (mh_rebin).
"""
import math  # noqa: F401  (leaked by the reference module)

import numpy as np

__all__ = ["bin_MUA_data", "online_histogram_w_sat_based_nb_of_samples", "approx_sort", "np", "math"]


def bin_MUA_data(MUA, bin_res):
    """Sum ``bin_res`` consecutive rows of a [T x C] count matrix -> int [ceil(T/bin_res) x C]
    (reference: functions_1.py:11-24).  Counts are carried as uint8 on the GPU like the
    reference's MATLAB-binned inputs; values above 255 in ``MUA`` are rejected."""
    import torch

    from . import _lib
    from .codec import _ptr, _stream
    from .container import layout

    MUA = np.asarray(MUA)
    if MUA.ndim != 2:
        raise IndexError("too many indices for array")  # MUA[:,1] in the reference
    if MUA.size and (MUA.min() < 0 or MUA.max() > 255):
        raise ValueError("bin_MUA_data: counts must fit uint8 (the GPU container is uint8)")
    T, C = MUA.shape
    bin_res = int(bin_res)
    nb = math.ceil(T / bin_res)
    off, ln, total = layout([T] * C)
    host = np.zeros(total + 16, np.uint8)
    for c in range(C):
        host[int(off[c]):int(off[c]) + T] = MUA[:, c]
    dev = torch.device("cuda", torch.cuda.current_device())
    d = torch.from_numpy(host).to(dev)
    out = torch.zeros(C * nb, dtype=torch.int32, device=dev)
    in_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    in_len = torch.from_numpy(ln.astype(np.int64)).to(dev)
    out_off = torch.arange(C, dtype=torch.int64, device=dev) * nb
    _lib.check(_lib.lib().mh_rebin(_ptr(d), _ptr(in_off), _ptr(in_len), C, T, bin_res, 0, _ptr(out),
                                   _ptr(out_off), _stream()))
    return out.cpu().numpy().reshape(C, nb).T.astype(int)


def online_histogram_w_sat_based_nb_of_samples(data_in, sample_val_cutoff, max_firing_rate):
    """Length of the calibration window and the histogram collected in it
    (reference: functions_1.py:27-68).  Returns ``(hist_dict, i)`` with
    ``i == min(sample_val_cutoff, len(data_in))``; saturates ``data_in[:i]`` in place at
    ``max_firing_rate``; keys are ``str(value)`` in first-seen order after the seeded '0';
    raises IndexError on an empty channel, all as the original does."""
    if len(data_in) == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    # the reference tests the counter AFTER entering a sample, so at least one is consumed
    i = int(min(max(int(math.ceil(sample_val_cutoff)), 1), len(data_in)))
    head = data_in[:i]
    head[head >= max_firing_rate] = max_firing_rate
    hist = {"0": 0}
    vals, first, counts = np.unique(head, return_index=True, return_counts=True)
    for k in np.argsort(first, kind="stable"):
        key = str(vals[k])
        hist[key] = hist.get(key, 0) + int(counts[k])
    return hist, i


def approx_sort(hist):
    """Unimodal approximate sort (reference: functions_1.py:75-90): the first maximum gets
    rank 0, then p-1, p+1, p-2, p+2, ... with the exhausted side skipped.  Returns
    ``(idx, hist[idx])`` with ``idx[k]`` = symbol that holds rank k."""
    hist = np.asarray(hist) if not isinstance(hist, np.ndarray) else hist
    S = len(hist)
    p = int(np.argmax(hist))
    m = min(p, S - 1 - p)
    idx = np.empty(S, dtype=int)
    idx[0] = p
    for k in range(1, S):
        if k <= 2 * m:
            j = (k + 1) >> 1
            idx[k] = p - j if (k & 1) else p + j
        else:
            j = k - m
            idx[k] = p - j if p > S - 1 - p else p + j
    return idx, hist[idx]
