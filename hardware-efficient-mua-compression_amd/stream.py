"""Calibrate-then-stream use of the codec on implant-style data.

The reference's RTL works in two phases (FPGA implementation/README.md:36-66): a CALIBRATION
phase fills, per channel, a RAM word {most frequent spike rate, selected encoder}
(RAM.v:4); in the COMPRESSION phase every later bin of every channel, arriving time-major
|CH1|CH2|...|CHN| per time step (README.md:31), is mapped and encoded with that fixed word.
`StreamEncoder` is the same protocol on the GPU: calibrate() on the first block, then
encode_block() on each later block -- de-interleave + encode with the preset word
(mh_encode_preset), no recalibration.  The channel-major intermediate between the two kernels
is PACKED: the encoder clips at S-1 anyway, so the de-interleaver writes min(x, 15) in 4 bits
per sample -- min(x, 3) in 2 bits when S <= 4 -- (mh_deinterleave_packed) and the encoder reads
those pieces directly; the intermediate's round trip through HBM shrinks from 2 x 1 byte per
sample to 2 x 1/2 resp. 2 x 1/4.
"""
import numpy as np
import torch

from . import MODE_APPROX, WIN_FULL, codec, container_io
from .container import ChannelSet


class StreamEncoder:
    def __init__(self, C, S, hist_bits, sclv, mode=MODE_APPROX, seg_chunks=2, device="cuda"):
        self.C, self.S, self.h, self.mode = int(C), int(S), int(hist_bits), int(mode)
        self.sclv = np.ascontiguousarray(np.asarray(sclv, dtype=np.uint8).reshape(-1, self.S))
        self.seg_chunks = int(seg_chunks)
        self.device = torch.device(device, torch.cuda.current_device()) if device == "cuda" else torch.device(device)
        self.peak = self.enc = None
        self._slots = {}

    def calibrate(self, block):
        """block: [T0, C] time-major counts holding at least 2^hist_bits time steps (fewer are
        accepted: the cutoff is min(2^h, T0), as functions_1.py:59-64).  Stores and returns the
        per-channel RAM word (peak, enc) as uint8 device tensors."""
        cs = ChannelSet.from_time_major(block, device=self.device)
        plan = codec.Plan(cs.ch_off, cs.ch_len, self.S, self.h, self.mode, WIN_FULL, self.sclv)
        m = plan.measure(cs.data)
        torch.cuda.synchronize()
        self.peak, self.enc = m.peak.clone(), m.enc.clone()
        plan.close()
        self.close()  # cached block plans point at the previous RAM word
        return self.peak, self.enc

    def _slot(self, Tb):
        """Plan and device buffers for blocks of Tb time steps, built once and reused: in the
        compression phase every block has the same shape, so nothing is planned or allocated
        per block."""
        slot = self._slots.get(Tb)
        if slot is None:
            bits = 2 if self.S <= 4 else 4
            # packed, chunk-blocked intermediate: chunk j (16384 samples = 1024 pieces) of channel c at
            # (j * C + c) * chunk_bytes -- the chunks of one time range are neighbours
            cb = 1024 * 2 * bits
            nchunks = (Tb + 16383) // 16384
            buf = torch.zeros(nchunks * self.C * cb + 16, dtype=torch.uint8, device=self.device)
            cs = ChannelSet(buf, np.arange(self.C, dtype=np.uint64) * np.uint64(cb), np.full(self.C, Tb, np.uint64))
            plan = codec.Plan(cs.ch_off, cs.ch_len, self.S, 0, self.mode, WIN_FULL, self.sclv,
                              seg_chunks=self.seg_chunks, input_bits=bits, chunk_stride=self.C * cb)
            e = plan.alloc_encoded()
            # the block's Encoded record points at the stored RAM word: nothing to copy per block
            e = codec.Encoded(e.payload, e.seg_words, e.ch_bits, self.peak, self.enc, e.skipped, e.seg_off, e.dense)
            slot = dict(cs=cs, plan=plan, enc=e,
                        d_off=torch.from_numpy(cs.ch_off.astype(np.int64)).to(self.device),
                        dense=torch.empty(plan.payload_cap_words, dtype=torch.int32, device=self.device),
                        off=torch.zeros(max(plan.n_segments, 1), dtype=torch.int64, device=self.device),
                        tot=torch.zeros(1, dtype=torch.int64, device=self.device))
            self._slots[Tb] = slot
        return slot

    def encode_block_device(self, block):
        """block: [Tb, C] time-major counts (device tensor or host array).  Enqueues
        de-interleave + preset encode + compaction on the current stream and returns
        (dense Encoded, total_words tensor, slot) without synchronising.  The buffers (dense
        words, sizes, total) belong to the shape's slot and are OVERWRITTEN by the next block of
        the same shape: a consumer that keeps a block in flight while the next one is enqueued
        (dist.gather_payload_pipelined) must copy them out -- or read the total and record an
        event -- before asking for the next block."""
        import ctypes as ct

        from . import _lib
        if self.peak is None:
            raise RuntimeError("calibrate() first")
        t = block if isinstance(block, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(block, np.uint8))
        t = t.to(self.device).contiguous()
        Tb, C = int(t.shape[0]), int(t.shape[1])
        if C != self.C:
            raise ValueError("block has %d channels, encoder was built for %d" % (C, self.C))
        slot = self._slot(Tb)
        cs, plan = slot["cs"], slot["plan"]
        _lib.check(_lib.lib().mh_deinterleave_packed(ct.c_void_p(t.data_ptr()), Tb, C, plan.input_bits,
                                                     ct.c_void_p(cs.data.data_ptr()),
                                                     ct.c_void_p(slot["d_off"].data_ptr()), plan.chunk_stride,
                                                     ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
        enc = slot["enc"]
        _lib.check(_lib.lib().mh_encode_preset(plan._h, ct.c_void_p(cs.data.data_ptr()), ct.c_void_p(self.peak.data_ptr()),
                                               ct.c_void_p(self.enc.data_ptr()), ct.c_void_p(enc.payload.data_ptr()),
                                               enc.payload.numel(), ct.c_void_p(enc.seg_words.data_ptr()),
                                               ct.c_void_p(enc.ch_bits.data_ptr()),
                                               ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
        dense, tot = plan.compact(enc, dense=slot["dense"], off=slot["off"], tot=slot["tot"])
        return dense, tot, slot

    def encode_block(self, block):
        """block: [Tb, C] time-major counts -> container_io.Compressed covering all Tb bins of
        every channel, coded with the stored RAM word."""
        dense, tot, slot = self.encode_block_device(block)
        plan, cs, enc = slot["plan"], slot["cs"], slot["enc"]
        torch.cuda.synchronize()
        total = int(tot.item())
        hdr = container_io.make_header(self.S, 0, self.mode, WIN_FULL, plan.seg_chunks, self.sclv)
        hdr["preset"] = True
        return container_io.Compressed(hdr, cs.ch_len.copy(), enc.peak.cpu().numpy(), enc.enc.cpu().numpy(),
                                       enc.skipped.cpu().numpy(), enc.ch_bits.cpu().numpy().astype(np.uint64),
                                       enc.seg_words.cpu().numpy().astype(np.uint64)[:plan.n_segments],
                                       dense.payload[:total].cpu().numpy().view(np.uint32).copy())

    def close(self):
        for slot in self._slots.values():
            slot["plan"].close()
        self._slots = {}

    @staticmethod
    def decode_block(c, device="cuda"):
        """-> [Tb, C] time-major array of min(x, S-1) (decoded and re-interleaved on the GPU)."""
        cs = container_io.decompress(c, device=device)
        if cs.C == 0:
            return np.zeros((0, 0), np.uint8)
        return cs.to_time_major().cpu().numpy()
