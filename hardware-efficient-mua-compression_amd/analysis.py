"""Consumers of the sweep results.

design_point_table   mean / worst-channel bit rate per (BP, S, log2 histogram size, #encoders),
                     averaged over CV runs -- what the reference writes into columns M/N of its
                     spreadsheet (Analyse results/integrate_BR_and_BDP_results_into_excel.py:93-140).
power_budget_test    how many of n random channel subsets of size Z exceed the implant power
                     budget (Analyse results/max_nb_channels_p_value_power_budget.py:76-133).
                     Same estimator; the draws come from torch's generator (on the GPU when
                     available), so individual counts are statistically, not bit-wise, equal to the
                     reference's np.random stream -- parity unpinned for this function.
"""
import numpy as np


def design_point_table(results, hist_bits=(2, 3, 4, 5, 6, 7, 8, 9, 10)):
    """results: {(S, BP, CV): dict with 'stored_all_var_BRs'} as returned by the sweep drivers.
    Returns float array of rows [BP, S, log2 hist, n_encoders, mean BR, worst BR] averaged over
    the CV runs, in the reference's iteration order (BP, then S, then round, then hist)."""
    cvs = sorted({k[2] for k in results})
    bps = sorted({k[1] for k in results}, key=lambda v: list(dict.fromkeys(k[1] for k in results)).index(v))
    Ss = sorted({k[0] for k in results})
    acc = None
    for cv in cvs:
        rows = []
        for BP in bps:
            for S in Ss:
                brs = results[(S, BP, cv)]["stored_all_var_BRs"]
                rounds = len(brs)
                for ri, per_hist in enumerate(brs):
                    for hi, ch in enumerate(per_hist):
                        a = np.array(ch)
                        rows.append([BP, S, int(hist_bits[hi]), rounds - ri, np.mean(a), np.max(a)])
        rows = np.array(rows, dtype=np.float64)
        acc = rows if acc is None else acc + rows
    return acc / len(cvs)


def power_budget_test(BRs, nb_channels_vec, nb_draws=100000, static_process_power=0.1618e-3,
                      chan_processing_power=0.96e-6, comm_energy=20e-9, ADC_power=0.0,
                      total_power_budget=10e-3 * (2.5e-1 * 2.5e-1), seed=0, device=None):
    """BRs: list (one entry per CV run) of 1-D arrays of per-channel bit rates (bits/s).
    Returns (exceed_counts[len(nb_channels_vec)], raw_power[len(nb_channels_vec)])."""
    import torch
    dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = torch.zeros((nb_draws, len(nb_channels_vec)), dtype=torch.float64, device=dev)
    raw = np.zeros(len(nb_channels_vec))
    for br in BRs:
        b = torch.as_tensor(np.asarray(br, dtype=np.float64), device=dev)
        for zi, Z in enumerate(nb_channels_vec):
            Z = int(Z)
            idx = torch.randint(0, len(b), (nb_draws, Z), generator=g, device=dev)  # with replacement
            x[:, zi] += comm_energy * b[idx].sum(1) + Z * (ADC_power + chan_processing_power) + static_process_power
            raw[zi] += Z * (comm_energy * 1e3 + ADC_power + chan_processing_power) + static_process_power
    x /= len(BRs)
    raw /= len(BRs)
    exceed = (x > total_power_budget).sum(0).cpu().numpy()
    return exceed, raw
