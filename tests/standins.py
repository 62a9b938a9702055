"""Synthetic stand-ins with the SHAPE of the reference's recordings (the datasets themselves are
not in the reference checkout: Data/*_data hold only placeholders).

Shapes follow the reference's own files: the test split lists 2 Flint, 10 Sabes and 2 Brochier
recordings (filenames_Flint_test.txt, filenames_Sabes_test.txt, filenames_Brochier_test.txt);
Sabes and Brochier arrays have 96 channels (Data/Load_and_bin_Sabes_store_as_mat_file.m:49-63,
Data/Load_and_bin_Brochier_store_as_mat_file.m:28,38); every recording is binned at 1, 5, 10, 20,
50 and 100 ms (Data/get_all_binned_data.py:16).  Lengths are 0.6-3.6 M bins at 1 ms (10-60
minute sessions), deliberately not multiples of any bin period.  Values are Poisson-like counts
from the product's integer-threshold generator, which the CPU oracle reproduces bit for bit.
"""
import numpy as np

BIN_VECTOR = [1, 5, 10, 20, 50, 100]          # Data/get_all_binned_data.py:16

# (dataset, channels, bins at 1 ms)
TEST_SET = [
    ("Flint", 104, 1_215_004), ("Flint", 137, 610_017),
    ("Sabes", 96, 603_001), ("Sabes", 96, 911_113), ("Sabes", 96, 1_250_007), ("Sabes", 96, 1_600_019),
    ("Sabes", 96, 1_999_999), ("Sabes", 96, 2_345_678), ("Sabes", 96, 2_718_281), ("Sabes", 96, 3_141_592),
    ("Sabes", 96, 3_333_331), ("Sabes", 96, 3_599_993),
    ("Brochier", 96, 711_003), ("Brochier", 96, 1_013_777),
]

SABES_RECORDING = ("Sabes", 96, 2_400_011)    # BASELINE configs[1]: one 96-channel recording


def lengths(recordings, scale=1):
    """Per-channel bin counts of the recordings laid end to end (every channel of a recording
    has the recording's length); scale > 1 shortens them for CPU-sized tests."""
    out = []
    for _name, C, T in recordings:
        out += [max(T // scale, 1)] * C
    return out


def thresholds(n_channels, lo=0.005, hi=0.08):
    """Integer inverse-CDF thresholds for rates log-uniform over [lo, hi] counts per 1 ms bin
    (0.25-4 per 50 ms bin: peaks at symbols 0, 1 and 2 all occur)."""
    from muahuff import synth
    return synth.thresholds(synth.channel_rates(n_channels, lo, hi))


def host_set(recordings, seed, scale=1):
    """The stand-in generated on the CPU by the oracle's generator -> (data, ch_off, ch_len)."""
    import oracle
    from muahuff import container
    ln = lengths(recordings, scale)
    off, ln, total = container.layout(ln)
    data = oracle.c.synth(off, ln, thresholds(len(ln)), seed, total=total + 64, nthreads=8)
    return data, off, ln


def device_set(recordings, seed, scale=1):
    """The same bytes generated on the GPU (mh_synth_poisson) as a ChannelSet."""
    from muahuff import container, synth
    cs = container.ChannelSet.empty(lengths(recordings, scale))
    synth.fill(cs, thresholds(cs.C), seed)
    return cs


def recording_slices(recordings):
    """[(name, first channel, one past last channel)] of each recording in the laid-out set."""
    out, c0 = [], 0
    for name, C, _T in recordings:
        out.append((name, c0, c0 + C))
        c0 += C
    return out


def dense_words(payload, seg_off, seg_words):
    """Used words of every segment back to back (payload: uint32 array with slots)."""
    parts = [payload[int(o):int(o) + int(n)] for o, n in zip(seg_off, seg_words)]
    return np.concatenate(parts) if parts else np.zeros(0, np.uint32)
