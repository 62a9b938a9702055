"""Drop-in for the reference's helper module ``Compressing data/functions_1.py``.

Same three functions, same signatures, return types and in-place side effects, so that
``from functions_1 import *`` in reference-shaped scripts keeps working (it also leaks ``np``
and ``math`` like the original).  ``approx_sort`` and
``online_histogram_w_sat_based_nb_of_samples`` are O(S) / closed-form host logic;
``bin_MUA_data`` runs on the GPU (mh_deinterleave + mh_rebin).
"""
import math  # noqa: F401  (leaked by the reference module)

import numpy as np

__all__ = ["bin_MUA_data", "online_histogram_w_sat_based_nb_of_samples", "approx_sort", "np", "math"]


def bin_MUA_data(MUA, bin_res):
    """Sum ``bin_res`` consecutive rows of a [T x C] count matrix -> int [ceil(T/bin_res) x C]
    (reference: functions_1.py:11-24), on the GPU: the matrix is the time-major layout
    mh_deinterleave reads, and mh_rebin sums each channel.  Any integer dtype is accepted: values
    outside 0..255 are split into byte planes (after subtracting the minimum), every plane is
    summed on the GPU and the planes are recombined exactly in int64.  Like the reference, needs at
    least 2 rows and 2 columns (it indexes MUA[:,1] and MUA[1,:]).  Deviations: non-integer dtypes
    raise TypeError (the reference would sum floats and truncate), bin_res above 4096 raises
    MuaHuffError (mh_rebin's limit; the reference's bin periods are 5..100)."""
    from .container import ChannelSet

    MUA = np.asarray(MUA)
    if MUA.ndim != 2:
        raise IndexError("too many indices for array")  # MUA[:,1] in the reference
    T, C = MUA.shape
    if C < 2:
        raise IndexError("index 1 is out of bounds for axis 1 with size %d" % C)
    if T < 2:
        raise IndexError("index 1 is out of bounds for axis 0 with size %d" % T)
    if MUA.dtype != np.bool_ and not np.issubdtype(MUA.dtype, np.integer):
        raise TypeError("bin_MUA_data: integer counts expected, got %s" % MUA.dtype)
    bin_res = int(bin_res)
    nb = math.ceil(T / bin_res)
    lo = int(MUA.min())
    v = MUA.astype(np.int64) - lo if lo < 0 or int(MUA.max()) > 255 else MUA
    top = int(v.max())
    total = np.zeros((nb, C), dtype=np.int64)
    plane, shift = 0, 0
    while True:
        byte = np.ascontiguousarray((v >> shift) & 0xFF if top > 255 else v, dtype=np.uint8)
        cs = ChannelSet.from_time_major(byte)
        sums, off, n = cs.rebin(bin_res, saturate=False)
        h = sums.cpu().numpy()
        total += (h[:nb * C].reshape(C, nb).T.astype(np.int64)) << shift
        plane, shift = plane + 1, shift + 8
        if (top >> shift) == 0:
            break
    if v is not MUA and lo != 0:  # undo the offset: every bin holds bin_res rows, the last one the rest
        rows = np.full(nb, bin_res, dtype=np.int64)
        rows[-1] = T - (nb - 1) * bin_res
        total += lo * rows[:, None]
    return total.astype(int)


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
