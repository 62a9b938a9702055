"""Consumers of the sweep results, reductions on the GPU.

design_point_table   mean / worst-channel bit rate per (BP, S, log2 histogram size, #encoders),
                     averaged over CV runs -- what the reference writes into columns M/N of its
                     spreadsheet (Analyse results/integrate_BR_and_BDP_results_into_excel.py:93-140).
                     np.mean / np.max of every design point in one mh_reduce_rows launch.
power_budget_test    how many of n random channel subsets of size Z exceed the implant power
                     budget (Analyse results/max_nb_channels_p_value_power_budget.py:76-133).
                     The subsets are drawn on the host from the legacy NumPy global RNG in the
                     reference's call order (np.random.choice per subset == one np.random.randint
                     call per subset size, same stream), gathered and summed on the GPU
                     (mh_power_draws) in NumPy's pairwise order: after np.random.seed(k) the
                     result equals the reference script's bit for bit
                     (tests/golden/power_budget.npz, made by running that script).
"""
import ctypes as ct

import numpy as np


def _dev():
    import torch

    from . import _lib
    if not torch.cuda.is_available():
        raise _lib.MuaHuffError(_lib.ERR_NO_DEVICE, "no MI355X visible; this package has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def reduce_rows(rows):
    """rows: list of 1-D float64 sequences -> (np.sum(row), np.max(row)) per row, computed on the
    GPU in NumPy's summation order (bit-exact; NaN propagates like np.max)."""
    import torch

    from . import _lib
    from .codec import _ptr, _stream
    dev = _dev()
    lens = np.array([len(r) for r in rows], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1) for r in rows]) if len(rows) else np.zeros(0)
    d_vals = torch.from_numpy(np.ascontiguousarray(flat if flat.size else np.zeros(1))).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_sum = torch.zeros(max(len(rows), 1), dtype=torch.float64, device=dev)
    d_max = torch.zeros_like(d_sum)
    _lib.check(_lib.lib().mh_reduce_rows(_ptr(d_vals), _ptr(d_off), len(rows), _ptr(d_sum), _ptr(d_max), _stream()))
    return d_sum.cpu().numpy()[:len(rows)], d_max.cpu().numpy()[:len(rows)], lens


def design_point_table(results, hist_bits=(2, 3, 4, 5, 6, 7, 8, 9, 10), bin_vector=None, S_vector=None, CV_vector=None):
    """results: {(S, BP, CV): dict with 'stored_all_var_BRs'} as returned by the sweep drivers.
    Returns float array of rows [BP, S, log2 hist, n_encoders, mean BR, worst BR] averaged over
    the CV runs, in the reference's iteration order (CV, then BP, then S, then reduction round,
    then histogram size; integrate_BR_and_BDP_results_into_excel.py:93-140; the worst-channel
    column is computed there at :119 and dropped from the sheet)."""
    cvs = list(CV_vector) if CV_vector is not None else sorted({k[2] for k in results})
    bps = list(bin_vector) if bin_vector is not None else list(dict.fromkeys(k[1] for k in results))
    Ss = list(S_vector) if S_vector is not None else sorted({k[0] for k in results})
    rows, keys = [], []
    for cv in cvs:
        for BP in bps:
            for S in Ss:
                brs = results[(S, BP, cv)]["stored_all_var_BRs"]
                rounds = len(brs)
                for ri, per_hist in enumerate(brs):
                    for hi, ch in enumerate(per_hist):
                        rows.append(np.asarray(ch, dtype=np.float64))
                        keys.append((BP, S, int(hist_bits[hi]), rounds - ri))
    sums, maxs, lens = reduce_rows(rows)
    with np.errstate(invalid="ignore", divide="ignore"):
        means = sums / lens  # np.mean = add.reduce / count
    per_cv = len(rows) // max(len(cvs), 1)
    acc = None
    for i in range(len(cvs)):
        sl = slice(i * per_cv, (i + 1) * per_cv)
        tab = np.column_stack([np.array(keys[sl], dtype=np.float64), means[sl], maxs[sl]])
        acc = tab if acc is None else acc + tab
    return acc / len(cvs)


def power_budget_test(BRs, nb_channels_vec, nb_draws=100000, static_process_power=0.1618e-3,
                      chan_processing_power=0.96e-6, comm_energy=20e-9, ADC_power=0,
                      total_power_budget=10e-3 * (2.5e-1 * 2.5e-1), rng=None, return_x=False):
    """BRs: list (one entry per CV run) of 1-D arrays of per-channel bit rates (bits/s), i.e.
    np.array(stored_all_var_BRs[rounds - nb_enc][hist_mem - 2]) of each BRs_*.pkl (:93).
    rng: a np.random.RandomState, default the legacy global one (np.random), which is what the
    reference draws from -- call np.random.seed(k) first to reproduce a run.
    Returns (exceed_counts[len(nb_channels_vec)], raw_power[len(nb_channels_vec)]) and, with
    return_x, the averaged power matrix x[nb_draws, len(nb_channels_vec)] of :111."""
    import torch

    from . import _lib
    from .codec import _ptr, _stream
    dev = _dev()
    rng = np.random if rng is None else rng
    nz = len(nb_channels_vec)
    x = torch.zeros((nb_draws, nz), dtype=torch.float64, device=dev)
    raw = np.zeros(nz)
    cv_count = 0
    for br in BRs:
        cv_count += 1
        b = np.ascontiguousarray(np.asarray(br, dtype=np.float64))
        d_br = torch.from_numpy(b).to(dev)
        for zi, Z in enumerate(nb_channels_vec):
            # nb_draws consecutive np.random.choice(channel_vec, Z) calls (:101) consume the legacy
            # stream exactly like one randint(0, len, (nb_draws, Z)) call
            idx = rng.randint(0, len(b), size=(nb_draws, int(Z)))
            d_idx = torch.from_numpy(idx.astype(np.int32)).to(dev).t().contiguous()  # [Z][draw]
            per_channels = Z * (ADC_power + chan_processing_power)                   # :104, numpy int * float
            _lib.check(_lib.lib().mh_power_draws(_ptr(d_br), len(b), _ptr(d_idx), int(Z), nb_draws,
                                                 ct.c_double(comm_energy), ct.c_double(float(per_channels)),
                                                 ct.c_double(static_process_power), ct.c_void_p(x[:, zi].data_ptr()), nz,
                                                 _stream()))
            raw[zi] += Z * (comm_energy * 1e3 + ADC_power + chan_processing_power) + static_process_power  # :107
            del d_idx
    raw = raw / cv_count          # :110
    xh = x.cpu().numpy() / cv_count  # :111
    exceed = np.sum(xh > total_power_budget, axis=0)  # :114, :118
    return (exceed, raw, xh) if return_x else (exceed, raw)
