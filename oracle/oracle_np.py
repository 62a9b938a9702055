"""NumPy restatement of the reference's compression-evaluation path -- TEST INFRASTRUCTURE ONLY.

Follows (paths relative to /root/reference):
  per-channel statement sequence   Compressing data/get_BR_with_approx_sort.py:164-193
                                   Compressing data/get_BR_no_sort.py:164-192
  bit-rate formula                 Compressing data/get_BR_with_approx_sort.py:284-292
  training + encoder-pool pruning  Compressing data/get_BR_with_approx_sort.py:138-147,223-321
  result container                 Compressing data/get_BR_with_approx_sort.py:327-334
  chosen-system evaluation         Compressing data/test_chosen_system.py:66-131

It is pinned against golden vectors made by running the reference itself
(oracle/make_golden.py -> tests/golden/).  Used by tests as the checker and by bench.py as
the "NumPy, single process" CPU baseline (closest analogue of how the reference executes).
"""
import numpy as np

HIST_SIZES = 2 ** np.array([2, 3, 4, 5, 6, 7, 8, 9, 10])  # get_BR_with_approx_sort.py:22


def cutoff(T, sample_val_cutoff):
    """functions_1.py:27-68 closed form: i == min(sample_val_cutoff, len(data_in))."""
    if T == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    return int(min(int(sample_val_cutoff), int(T)))


def approx_sort_idx(hist):
    """functions_1.py:75-90 as the rule p, p-1, p+1, ... (first max wins)."""
    S = len(hist)
    p = int(np.argmax(hist))
    out = [p]
    d = 1
    while len(out) < S:
        if p - d >= 0:
            out.append(p - d)
        if p + d < S and len(out) < S:
            out.append(p + d)
        d += 1
    return np.array(out, dtype=np.int64)


def hist_clipped(x, S):
    """np.histogram(clipped, arange(-0.5, S+0.5))[0] == bincount(min(x,S-1), minlength=S)."""
    return np.bincount(np.minimum(x, S - 1), minlength=S).astype(np.int64)


def channel_stats(x, S, sample_val_cutoff, mode_approx, skip_rule=True):
    """One validation channel at one histogram size (:164-193).  Returns a dict with
    c, e, skipped, idx, cal_sorted[S], post_mapped[S] (zeros when skipped)."""
    T = len(x)
    c = cutoff(T, sample_val_cutoff)
    cal = hist_clipped(x[:c], S)
    idx = approx_sort_idx(cal) if mode_approx else np.arange(S, dtype=np.int64)
    e = c + int(T / 2)
    skipped = bool(skip_rule and e > T)
    if skipped:
        post = np.zeros(S, dtype=np.int64)
    else:
        post = hist_clipped(x[c:e], S)[idx]
    return dict(c=c, e=e, skipped=skipped, idx=idx, cal_sorted=cal[idx], post_mapped=post)


def bit_rate(bits, n, BP):
    """:289-292  BR = 1000/(BP/(bits/n)) in float64; 0/0 -> nan for skipped channels."""
    with np.errstate(invalid="ignore", divide="ignore"):
        abps = np.float64(bits) / np.float64(n)
        return np.float64(1000) / (np.float64(BP) / abps)


def first_argmin_rows(cost):
    return np.argmin(cost, axis=1)


def evaluate_split(train, val, S, sclvs, BP, mode_approx):
    """Everything the reference does for one (CV, BP, S): returns the dict it pickles.
    train/val: lists of 1-D uint8 arrays; sclvs: float array [K,S] in pickle order."""
    sclvs = np.asarray(sclvs, dtype=np.float64)
    K = sclvs.shape[0]
    n_train, n_val = len(train), len(val)
    # :138-147 training histograms, sorted descending
    train_h = np.zeros((S, n_train))
    for i, x in enumerate(train):
        train_h[:, i] = np.flip(np.sort(hist_clipped(x, S)))
    # :150-210 validation histograms for every histogram size
    cal_mem, post_mem = [], []
    c_all = np.zeros((n_val, len(HIST_SIZES)))
    e_all = np.zeros((n_val, len(HIST_SIZES)))
    for hi, cut in enumerate(HIST_SIZES):
        cal = np.zeros((S, n_val))
        post = np.zeros((S, n_val))
        for ch, x in enumerate(val):
            st = channel_stats(x, S, cut, mode_approx)
            c_all[ch, hi] = st["c"]
            e_all[ch, hi] = st["e"]
            cal[:, ch] = st["cal_sorted"]
            post[:, ch] = st["post_mapped"]
        cal_mem.append(cal)
        post_mem.append(post)
    with np.errstate(invalid="ignore", divide="ignore"):
        proportion = (e_all.astype(int) - c_all.astype(int)) / e_all.astype(int)  # :214
    stored_SCLVs, stored_BRs, stored_hist = [], [], []
    cur = sclvs.copy()
    while cur.shape[0] != 0:  # :223
        stored_SCLVs.append(cur.copy())
        dot = train_h.T @ cur.T  # :231
        assign = first_argmin_rows(dot) if n_train else np.zeros(0, dtype=np.int64)
        stored_hist.append(np.bincount(assign, minlength=cur.shape[0]).astype(np.int64))  # :239-242
        rounds = []
        for hi in range(len(HIST_SIZES)):  # :250-296
            vdot = cal_mem[hi].T @ cur.T
            post = post_mem[hi]
            brs = []
            for ch in range(n_val):
                k = int(np.argmin(vdot[ch, :]))
                n = np.sum(post[:, ch])
                bits = np.sum(cur[k, :] * post[:, ch])
                brs.append(bit_rate(bits, n, BP))
            rounds.append(brs)
        stored_BRs.append(rounds)
        if cur.shape[0] != 1:  # :310-316
            cost = np.zeros(cur.shape[0])
            for r in range(cur.shape[0]):
                cost[r] = np.mean(np.min(np.delete(dot, r, axis=1), axis=1))
            cur = np.delete(cur, int(np.argmin(cost)), axis=0)
        else:
            cur = np.delete(cur, 0, axis=0)
    return dict(stored_all_var_BRs=stored_BRs, stored_SCLVs=stored_SCLVs,
                stored_hist_SCLVs=stored_hist, stored_val_BR_data_proportion=proportion)


def split_channels(all_data, how_many_sabes=2000, train_percentage=50):
    """:78-97 -- consumes one np.random.permutation per dataset from the legacy global RNG."""
    train, val = [], []
    for ds, data in enumerate(all_data):
        order = np.random.permutation(len(data))
        data = [data[i] for i in order]
        if ds == 1:
            data = data[:how_many_sabes]
        cut = int(np.round(train_percentage * len(data) / 100))
        train.extend(data[:cut])
        val.extend(data[cut:])
    return train, val


def run_sweep(all_binned_data, bin_vector, sclv_tables, mode_approx, nb_CV_iterations=30,
              S_values=range(2, 11), how_many_sabes=2000):
    """Whole driver (:70-334).  Yields ((S, BP, CV), result dict) in the reference's order.
    The caller seeds np.random beforehand."""
    for cv in np.arange(1, nb_CV_iterations, 1):
        for bp_i, BP in enumerate(bin_vector):
            train, val = split_channels(all_binned_data[bp_i], how_many_sabes)
            for S in S_values:
                yield (int(S), BP, int(cv)), evaluate_split(train, val, int(S), sclv_tables[int(S)],
                                                            BP, mode_approx)


def chosen_system(all_data, BP=50, S=3, hist_memory=6, sclv=(1, 2, 2)):
    """test_chosen_system.py:66-125 -- mean BR per dataset (no skip rule, slice truncates)."""
    out = []
    for data in all_data:
        abps = np.zeros(len(data))
        for ch, x in enumerate(data):
            st = channel_stats(x, S, 2 ** hist_memory, True, skip_rule=False)
            post = hist_clipped(x[st["c"]:st["e"]], S)[st["idx"]].astype(np.float64)
            with np.errstate(invalid="ignore", divide="ignore"):
                abps[ch] = np.matmul(post, np.array(sclv)) / np.sum(post)
        with np.errstate(invalid="ignore"):
            out.append(np.mean(abps) / (BP / 1000))
    if len(out) == 2:
        out.append(float("nan"))
    return out
