"""Shared engine of the two bit-rate sweep drivers.

Reproduces what ``Compressing data/get_BR_with_approx_sort.py`` and ``get_BR_no_sort.py`` of
the reference compute and write -- cross-validation x bin period x S x histogram size x
encoder-pool pruning -> ``BRs_S_<S>_BP_<BP>_CV_<cv>.pkl`` -- with every O(T) pass
(calibration, window histograms, whole-channel training histograms) on the GPU through
``mh_measure`` and the tiny O(K^2 C) training/pruning logic on the host in NumPy, in the
reference's floating-point operation order.  With the legacy global NumPy RNG seeded the same
way the output is bit-identical to the reference's (tests/test_gpu_drivers.py).
"""
import os
import pickle
import re

import numpy as np

HIST_BITS = (2, 3, 4, 5, 6, 7, 8, 9, 10)  # samples_per_channel_for_histogram_vector = 2**[2..10]


def read_directories(root_directory):
    """Parse ``<root>/directories.txt`` (``key = 'path'`` lines, reference format:
    get_BR_no_sort.py:34-53).  Returns {key: path}."""
    out = {}
    with open(os.path.join(root_directory, "directories.txt")) as f:
        for line in f:
            m = re.match(r"\s*([A-Za-z_0-9]+)\s*=", line)
            p = re.search(r"'(.*?)'", line)
            if m and p and not line.lstrip().startswith("%"):
                out[m.group(1)] = p.group(1)
    return out


def load_binned(directory, which):
    """``all_binned_data_<which>.pkl`` in the layout written by Data/get_all_binned_data.py:73-80
    (a file the user produced -- it is unpickled)."""
    with open(os.path.join(directory, "all_binned_data_%s.pkl" % which), "rb") as f:
        r = pickle.load(f)
    return r["all_binned_data"], list(r["bin_vector"]), list(r["datasets"])


def sclv_tables(directories):
    from .. import sclv
    tabs = sclv.all_tables()
    path = directories.get("SCLV_path")
    if path and os.path.isdir(path):
        tabs.update(sclv.load_directory(path))
    return tabs


class DeviceChannels:
    """All channels of one bin period resident on the GPU; subsets are views by index.
    With fused=True one mh_sweep_run pass at construction serves every later request
    (all S, all histogram sizes, all CV splits) from host-side integer tables."""

    def __init__(self, datasets, fused=False):
        from ..container import ChannelSet
        self.flat = [ch for ds in datasets for ch in ds]
        self.base = np.cumsum([0] + [len(ds) for ds in datasets])  # dataset -> first flat index
        self.cs = ChannelSet.from_channels(self.flat)
        self.sweep = None
        if fused and self.flat:
            import torch

            from ..sweep import SweepHist
            self.sweep = SweepHist(self.cs.ch_off, self.cs.ch_len, HIST_BITS).run(self.cs.data)
            torch.cuda.synchronize()

    def train_hist(self, idx, S, table):
        """whole-channel histograms by symbol, float [len(idx), S] (:140-146)"""
        from .. import MODE_NOSORT, WIN_FULL
        if self.sweep is not None:
            return self.sweep.train_hist(idx, S).astype(np.float64)
        return self.measure(idx, S, 0, MODE_NOSORT, WIN_FULL, table)["post"]

    def validation(self, idx, S, h, approx, table):
        from .. import MODE_APPROX, MODE_NOSORT, WIN_REF_HALF
        if self.sweep is not None:
            return self.sweep.validation(idx, S, h, approx)
        return self.measure(idx, S, h, MODE_APPROX if approx else MODE_NOSORT, WIN_REF_HALF, table)

    def measure(self, idx, S, h, mode, window, table):
        """mh_measure over the channels `idx` (flat indices) -> dict of host arrays."""
        import torch

        from .. import codec
        idx = np.asarray(idx, dtype=np.int64)
        C = len(idx)
        if C == 0:
            z = np.zeros((0, S))
            return dict(cutoff=np.zeros(0, np.int64), cal=z.copy(), post=z.copy(), skipped=np.zeros(0, np.uint8),
                        length=np.zeros(0, np.int64))
        plan = codec.Plan(self.cs.ch_off[idx], self.cs.ch_len[idx], S, h, mode, window, table)
        m = plan.measure(self.cs.data)
        torch.cuda.synchronize()
        out = dict(cutoff=m.cutoff.cpu().numpy(), cal=m.cal_hist.cpu().numpy().astype(np.float64),
                   post=m.post_hist.cpu().numpy().astype(np.float64), skipped=m.skipped.cpu().numpy(),
                   length=self.cs.ch_len[idx].astype(np.int64))
        plan.close()
        return out


def split_indices(n_per_dataset, base, how_many_sabes, train_percentage):
    """Channel shuffle + split (get_BR_with_approx_sort.py:78-97).  One
    ``np.random.permutation`` per dataset from the legacy global RNG, in dataset order."""
    train, val = [], []
    for ds, n in enumerate(n_per_dataset):
        order = np.random.permutation(n)
        chans = [int(base[ds] + i) for i in order]
        if ds == 1:  # Sabes
            chans = chans[:how_many_sabes]
        cut = int(np.round(train_percentage * len(chans) / 100))
        train.extend(chans[:cut])
        val.extend(chans[cut:])
    return train, val


def evaluate(dev, train_idx, val_idx, S, table, BP, approx, python_floats=False):
    """One (CV, BP, S) cell -> the dict the reference pickles (:138-334).
    python_floats: store each BR as a Python float instead of np.float64 (same bits; a list of
    1e6 np.float64 scalars takes ~1.7 s to pickle, the same list of floats 0.02 s)."""
    from ..codec import bit_rate
    S = int(S)
    sclvs = np.array([np.asarray(r, dtype=np.float64) for r in table], dtype=object)  # :125
    n_train, n_val = len(train_idx), len(val_idx)
    # training histograms of the whole channel, sorted descending (:140-147)
    tr = dev.train_hist(train_idx, S, table)
    histograms = np.zeros((S, n_train))
    for c in range(n_train):
        histograms[:, c] = np.flip(np.sort(tr[c]))
    # validation histograms per histogram size (:157-210)
    nh = len(HIST_BITS)
    cal_all = np.zeros((nh, n_val, S))   # [hist size, channel, rank]
    post_all = np.zeros((nh, n_val, S))  # zeros for skipped channels
    c_all = np.zeros((n_val, nh))
    e_all = np.zeros((n_val, nh))
    for hi, h in enumerate(HIST_BITS):
        v = dev.validation(val_idx, S, h, approx, table)
        c_all[:, hi] = v["cutoff"]
        e_all[:, hi] = v["cutoff"] + (v["length"] // 2)  # :180 int(len/2)
        cal_all[hi] = v["cal"]
        post_all[hi] = v["post"]
    n_all = post_all.sum(axis=2)         # [hist size, channel]
    with np.errstate(invalid="ignore", divide="ignore"):
        proportion = (e_all.astype(int) - c_all.astype(int)) / e_all.astype(int)  # :214
    stored_SCLVs, stored_BRs, stored_hist = [], [], []
    while len(sclvs) != 0:  # :223
        stored_SCLVs.append(sclvs)
        cur = sclvs.astype(np.float64)
        dot = histograms.T @ cur.T  # :231 (integer-valued, exact)
        assign = np.argmin(dot, axis=1) if n_train else np.zeros(0, dtype=np.int64)  # :236
        stored_hist.append(np.bincount(assign, minlength=len(cur)).astype(np.int64))  # :239-242
        # :250-296 for all histogram sizes and channels at once: every sum below is a sum of
        # integer-valued doubles (exact in any order), the two divisions are element-wise float64
        # as in the reference
        if n_val:
            k = np.argmin(cal_all @ cur.T, axis=2)             # :281 first min, [hist size, channel]
            bits = (cur[k] * post_all).sum(axis=2)
        else:
            bits = np.zeros((nh, 0))
        br = bit_rate(bits, n_all, BP)
        per_hist = [row.tolist() if python_floats else list(row) for row in br]
        stored_BRs.append(per_hist)
        if len(sclvs) != 1:  # :310-316 drop the encoder whose removal hurts the training set least
            # min over the columns other than r == the row minimum, or the second smallest where
            # column r holds (the first occurrence of) the minimum; np.mean sees the same vector
            # of values in the same order as np.mean(np.min(np.delete(dot, r, axis=1), axis=1))
            cost = np.zeros(len(sclvs))
            if n_train:
                part = np.partition(dot, 1, axis=1)
                m1, m2 = part[:, 0], part[:, 1]
                for r in range(len(sclvs)):
                    cost[r] = np.mean(np.where(assign == r, m2, m1))
            else:
                cost[:] = np.nan
            sclvs = np.delete(sclvs, np.argmin(cost), axis=0)
        else:
            sclvs = np.delete(sclvs, 0, axis=0)
    return {"stored_all_var_BRs": stored_BRs, "stored_SCLVs": stored_SCLVs,
            "stored_hist_SCLVs": stored_hist, "stored_val_BR_data_proportion": proportion}


def run(root_directory, approx, nb_CV_iterations=30, how_many_channels_Sabes=2000, train_percentage=50,
        S_values=range(2, 11), write=True, verbose=True, fused=True, python_floats=False):
    """Whole sweep.  Returns {(S, BP, CV): result dict}; writes the pickles when `write`.
    python_floats=True writes the bit rates as Python floats (identical values, ~90x faster
    pickling; the default keeps the reference's np.float64 elements).
    fused=True: one GPU pass per bin period (mh_sweep_run); fused=False: one mh_measure per
    (CV, S, h) -- same results, kept as a cross-check."""
    d = read_directories(root_directory)
    all_binned, bin_vector, _datasets = load_binned(d["Formatted_data_path"], "train")
    results_dir = d["BR_approx_sort_results" if approx else "BR_no_sort_results"]
    tabs = sclv_tables(d)
    devs = [DeviceChannels(all_binned[i], fused=fused) for i in range(len(bin_vector))]
    out = {}
    for cv in np.arange(1, nb_CV_iterations, 1):  # 1 .. nb-1, as the reference (:70)
        for bp_i, BP in enumerate(bin_vector):
            dev = devs[bp_i]
            n_per = [len(ds) for ds in all_binned[bp_i]]
            train_idx, val_idx = split_indices(n_per, dev.base, how_many_channels_Sabes, train_percentage)
            for S in S_values:
                if verbose:
                    print("BP: " + str(BP) + "; S: " + str(int(S)))
                res = evaluate(dev, train_idx, val_idx, int(S), tabs[int(S)], BP, approx, python_floats)
                out[(int(S), BP, int(cv))] = res
                if write:
                    os.makedirs(results_dir, exist_ok=True)
                    fn = os.path.join(results_dir, "BRs_S_%d_BP_%s_CV_%d.pkl" % (int(S), str(BP), int(cv)))
                    with open(fn, "wb") as f:
                        pickle.dump(res, f)
    return out
