"""Fused sweep histograms: ONE pass over a channel set gives every histogram the reference's
train/validate sweep needs -- for all S = 2..10, all histogram sizes 2^h and every CV split
(mh_sweep_* in include/muahuff.h).  The reference re-reads each validation channel once per
(S, h) and per split (get_BR_with_approx_sort.py:107,157,161).

After `SweepHist.run()` everything is host arithmetic on [C, 19, 10] integers:
  calibration histogram  H(c_h)                 (get_BR_with_approx_sort.py:171)
  post histogram         H(c_h + T/2) - H(c_h)  (:189), zero + skipped when c_h + T/2 > T (:183)
  training histogram     H(T)                   (:146)
with the bins >= S-1 merged for a dynamic range S (:143,164).
"""
import ctypes as ct

import numpy as np
import torch

from . import _lib
from .codec import _need_gpu, _ptr, _stream

BINS = 10


def perm_table(S, approx):
    """perm[p] = symbols in rank order for a calibration histogram peaking at p
    (functions_1.py:75-90 closed form); identity rows for the no-sort mapper."""
    out = np.zeros((S, S), dtype=np.int64)
    for p in range(S):
        if not approx:
            out[p] = np.arange(S)
            continue
        idx = np.zeros(S, np.uint8)
        _lib.check(_lib.lib().mh_approx_sort_perm(S, p, idx.ctypes.data))
        out[p] = idx
    return out


class SweepHist:
    def __init__(self, ch_off, ch_len, hist_bits=(2, 3, 4, 5, 6, 7, 8, 9, 10)):
        _need_gpu()
        self.ch_off = np.ascontiguousarray(ch_off, dtype=np.uint64)
        self.ch_len = np.ascontiguousarray(ch_len, dtype=np.uint64)
        self.hist_bits = tuple(int(h) for h in hist_bits)
        self.C = len(self.ch_len)
        hb = np.array(self.hist_bits, dtype=np.uint32)
        h_ = ct.c_void_p()
        rc = _lib.lib().mh_sweep_create(ct.byref(h_), self.ch_off.ctypes.data, self.ch_len.ctypes.data,
                                        self.C, hb.ctypes.data, len(hb))
        if rc == _lib.ERR_EMPTY_CHANNEL:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")
        _lib.check(rc)
        self._h = h_
        ni = ct.c_uint32(0)
        _lib.check(_lib.lib().mh_sweep_info(self._h, ct.byref(ni), None))
        self.n_intervals = int(ni.value)
        self.bounds = np.zeros((self.C, self.n_intervals + 1), dtype=np.uint64)
        _lib.check(_lib.lib().mh_sweep_info(self._h, None, self.bounds.ctypes.data))
        self.cum = None

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().mh_sweep_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, data):
        """One pass over `data` (uint8 device tensor in this layout).  Fills self.cum[C, ni+1, 10]:
        cum[c, j] = histogram of min(x, 9) over [0, bounds[c, j])."""
        dev = data.device
        hist = torch.zeros((self.C, self.n_intervals, BINS), dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().mh_sweep_run(self._h, _ptr(data), _ptr(hist), _stream()))
        h = hist.cpu().numpy()
        self.cum = np.zeros((self.C, self.n_intervals + 1, BINS), dtype=np.int64)
        np.cumsum(h, axis=1, out=self.cum[:, 1:, :])
        return self

    # ---- host arithmetic ------------------------------------------------------------------
    def _at(self, idx, pos):
        """prefix histogram [len(idx), 10] at sample position pos[i] of channel idx[i]"""
        b = self.bounds[idx]  # [n, intervals + 1], ascending per row
        j = (b < np.asarray(pos, dtype=b.dtype)[:, None]).sum(axis=1)  # row-wise searchsorted(side="left")
        return self.cum[idx, j, :]

    @staticmethod
    def _merge(h10, S):
        out = h10[:, :S].copy()
        out[:, S - 1] = h10[:, S - 1:].sum(axis=1)
        return out

    def train_hist(self, idx, S):
        """whole-channel histogram by symbol, [len(idx), S]"""
        idx = np.asarray(idx, dtype=np.int64)
        if len(idx) == 0:
            return np.zeros((0, S), np.int64)
        return self._merge(self._at(idx, self.ch_len[idx]), S)

    def validation(self, idx, S, h, approx):
        """What mh_measure returns for window MH_WIN_REF_HALF at (S, h): dict with cutoff,
        cal (rank order), post (rank order, zero when skipped), skipped, length."""
        idx = np.asarray(idx, dtype=np.int64)
        n = len(idx)
        T = self.ch_len[idx].astype(np.int64)
        if n == 0:
            z = np.zeros((0, S))
            return dict(cutoff=np.zeros(0, np.int64), cal=z, post=z.copy(), skipped=np.zeros(0, np.uint8), length=T)
        c = np.minimum(np.int64(1) << h, T)
        e = c + T // 2
        skipped = e > T
        Hc = self._at(idx, c.astype(np.uint64))
        He = self._at(idx, np.minimum(e, T).astype(np.uint64))
        cal = self._merge(Hc, S)
        post = self._merge(He - Hc, S)
        post[skipped] = 0
        peak = np.argmax(cal, axis=1) if approx else np.zeros(n, dtype=np.int64)  # first max
        perm = perm_table(S, approx)[peak]                                       # [n, S]
        return dict(cutoff=c, cal=np.take_along_axis(cal, perm, axis=1).astype(np.float64),
                    post=np.take_along_axis(post, perm, axis=1).astype(np.float64),
                    skipped=skipped.astype(np.uint8), length=T)
