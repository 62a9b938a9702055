"""Input container: channels of binned MUA counts, channel-major, in one uint8 device buffer.

The reference keeps ``all_binned_data[BP_idx][dataset_idx][channel]`` = 1-D uint8 array
(Data/get_all_binned_data.py:36-69; consumers get_BR_with_approx_sort.py:60-75).  On the GPU
the same channels live back to back in HBM, each starting on a 16-byte boundary so that a
wavefront reads 1 KiB of one channel per load instruction.
"""
import numpy as np
import torch

ALIGN = 16          # every channel starts on a 16-byte boundary (one piece)
LINE_ALIGN = 128    # ... and a LONG one on a 128-byte line: with the head segment of container format revision 3
LONG = 16 * 16384   # (include/muahuff.h, MH_HEAD_ALIGN / MH_HEAD_MIN_WINDOW) all its 1-KiB rows then start on a line


def layout(lengths, align=None):
    """(ch_off, ch_len, total_bytes) for channels of the given lengths.  align=None: 16 bytes, 128 for channels of
    at least 2^18 bins (whose windows get head segments)."""
    ln = np.asarray(lengths, dtype=np.uint64)
    al = (np.where(ln >= LONG, LINE_ALIGN, ALIGN) if align is None else np.full(len(ln), align)).astype(np.uint64)
    off = np.zeros(len(ln), dtype=np.uint64)
    if len(ln) and (al == al[0]).all():      # one alignment throughout: closed form
        pad = (ln + al - np.uint64(1)) & ~(al - np.uint64(1))
        off[1:] = np.cumsum(pad)[:-1]
        return off, ln, int(pad.sum())
    cur = 0
    for i in range(len(ln)):
        a = int(al[i])
        cur = (cur + a - 1) // a * a
        off[i] = cur
        cur += int(ln[i])
    return off, ln, (cur + ALIGN - 1) // ALIGN * ALIGN


class ChannelSet:
    """A ragged set of uint8 channels resident on one GPU."""

    def __init__(self, data, ch_off, ch_len):
        self.data = data  # torch.uint8 [total] on the GPU
        self.ch_off = np.ascontiguousarray(ch_off, dtype=np.uint64)
        self.ch_len = np.ascontiguousarray(ch_len, dtype=np.uint64)

    @property
    def C(self):
        return len(self.ch_len)

    @property
    def device(self):
        return self.data.device

    def matrix(self, buf=None):
        """[C, T] strided view of the channels (all of one length) in `buf` -- the set's own buffer, or another
        one laid out like it (a decoder's output).  Channel starts are padded (16 bytes; 128 for channels of >= 2^18
        bins), so this is the set's buffer seen with a row pitch -- not a reshape of its first C * T bytes."""
        buf = self.data if buf is None else buf
        if self.C == 0:
            return buf[:0].view(0, 0)
        T = int(self.ch_len[0])
        if not (self.ch_len == self.ch_len[0]).all():
            raise ValueError("matrix(): channels of different lengths")
        pitch = int(self.ch_off[1] - self.ch_off[0]) if self.C > 1 else T
        return torch.as_strided(buf, (self.C, T), (pitch, 1), int(self.ch_off[0]))

    @classmethod
    def from_channels(cls, channels, device="cuda"):
        """channels: list of 1-D arrays of counts (values above 255 saturate like MATLAB uint8)."""
        off, ln, total = layout([len(c) for c in channels])
        host = np.zeros(total + ALIGN, dtype=np.uint8)
        for c, o in zip(channels, off):
            a = np.asarray(c)
            if a.dtype != np.uint8:
                a = np.clip(a, 0, 255).astype(np.uint8)
            host[int(o):int(o) + len(a)] = a
        return cls(torch.from_numpy(host).to(device), off, ln)

    @classmethod
    def from_time_major(cls, samples, device="cuda"):
        """samples: [T, C] uint8 (host array or device tensor), one row per time step with the
        channels interleaved |CH1|CH2|...|CHN| -- the order an implant streams binned counts
        (reference RTL compression phase, FPGA implementation/README.md:31).  De-interleaved on
        the GPU (mh_deinterleave) into the channel-major layout the codec reads."""
        import ctypes as ct

        from . import _lib
        t = samples if isinstance(samples, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(samples, np.uint8))
        t = t.to(device).contiguous()
        T, C = int(t.shape[0]), int(t.shape[1])
        cs = cls.empty([T] * C, device=t.device)
        d_off = torch.from_numpy(cs.ch_off.astype(np.int64)).to(t.device)
        _lib.check(_lib.lib().mh_deinterleave(ct.c_void_p(t.data_ptr()), T, C, ct.c_void_p(cs.data.data_ptr()),
                                              ct.c_void_p(d_off.data_ptr()),
                                              ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return cs

    @classmethod
    def empty(cls, lengths, device="cuda"):
        off, ln, total = layout(lengths)
        return cls(torch.zeros(total + ALIGN, dtype=torch.uint8, device=device), off, ln)

    def channel(self, i):
        o, n = int(self.ch_off[i]), int(self.ch_len[i])
        return self.data[o:o + n]

    def to_time_major(self):
        """[T, C] uint8 device tensor, |CH1|CH2|...|CHN| per time step (mh_interleave); all
        channels must have the same length."""
        import ctypes as ct

        from . import _lib
        T = int(self.ch_len[0]) if self.C else 0
        if not np.all(self.ch_len == np.uint64(T)):
            raise ValueError("to_time_major needs channels of equal length")
        out = torch.empty((T, self.C), dtype=torch.uint8, device=self.data.device)
        d_off = torch.from_numpy(self.ch_off.astype(np.int64)).to(self.data.device)
        _lib.check(_lib.lib().mh_interleave(ct.c_void_p(self.data.data_ptr()), ct.c_void_p(d_off.data_ptr()), T, self.C,
                                            ct.c_void_p(out.data_ptr()),
                                            ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def rebin(self, r, saturate=True):
        """Sum ``r`` consecutive bins of every channel on the GPU (mh_rebin): one 1 ms recording
        feeds all of the reference's bin periods (Data/Load_and_bin_Sabes_store_as_mat_file.m:50-54
        bins at 1, 5, 10, 20, 50, 100 ms; Compressing data/functions_1.py:11-24).  The last,
        partial bin is kept (ceil).  saturate=True returns a uint8 ChannelSet whose sums clamp at
        255 like MATLAB's uint8(); saturate=False returns (int32 tensor, offsets, lengths) with the
        exact sums."""
        import ctypes as ct

        from . import _lib
        r = int(r)
        nb = (self.ch_len + np.uint64(r - 1)) // np.uint64(r)
        dev = self.data.device
        d_off = torch.from_numpy(self.ch_off.astype(np.int64)).to(dev)
        d_len = torch.from_numpy(self.ch_len.astype(np.int64)).to(dev)
        vp = ct.c_void_p
        st = vp(torch.cuda.current_stream().cuda_stream)
        max_len = int(self.ch_len.max()) if self.C else 0
        if saturate:
            out = ChannelSet.empty([int(n) for n in nb], device=dev)
            o_off = torch.from_numpy(out.ch_off.astype(np.int64)).to(dev)
            _lib.check(_lib.lib().mh_rebin(vp(self.data.data_ptr()), vp(d_off.data_ptr()), vp(d_len.data_ptr()), self.C,
                                           max_len, r, 1, vp(out.data.data_ptr()), vp(o_off.data_ptr()), st))
            return out
        off = np.concatenate([[0], np.cumsum(nb)[:-1]]).astype(np.int64) if self.C else np.zeros(0, np.int64)
        out = torch.zeros(int(nb.sum()) + 4, dtype=torch.int32, device=dev)
        o_off = torch.from_numpy(off).to(dev)
        _lib.check(_lib.lib().mh_rebin(vp(self.data.data_ptr()), vp(d_off.data_ptr()), vp(d_len.data_ptr()), self.C,
                                       max_len, r, 0, vp(out.data_ptr()), vp(o_off.data_ptr()), st))
        return out, off, nb.astype(np.int64)

    def to_channels(self):
        host = self.data.cpu().numpy()
        return [host[int(o):int(o) + int(n)].copy() for o, n in zip(self.ch_off, self.ch_len)]
