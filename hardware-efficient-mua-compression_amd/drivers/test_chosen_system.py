"""Chosen design point on the held-out set: GPU counterpart of the reference's
``Compressing data/test_chosen_system.py`` (S=3, BP=50 ms, 2^6-sample histogram, one encoder
with SCLV [1,2,2]; no skip rule -- the measured window truncates, :99-103).

Returns / prints the mean bit rate per dataset exactly as the reference prints it (:125-131).
"""
import numpy as np

from . import _sweep

BIN_RESOLUTION = 50
BP_COUNTER = -2       # index of the 50 ms data in all_binned_data (reference :23)
S = 3
HIST_MEMORY = 6       # bits
SCLV = [1, 2, 2]      # encoder ['0', '10', '11'] (reference :26-27)


def run(root_directory, train_or_test="test", verbose=True):
    from .. import MODE_APPROX, WIN_REF_HALF_TRUNC
    d = _sweep.read_directories(root_directory)
    all_binned, _bin_vector, _datasets = _sweep.load_binned(d["Formatted_data_path"], train_or_test)
    all_data = all_binned[BP_COUNTER]
    BR = []
    for data in all_data:
        if verbose:
            print("BP: " + str(BIN_RESOLUTION) + "; S: " + str(int(S)))
        dev = _sweep.DeviceChannels([data])
        m = dev.measure(np.arange(len(data)), S, HIST_MEMORY, MODE_APPROX, WIN_REF_HALF_TRUNC, [SCLV])
        post = m["post"]                                     # [C, S] rank-mapped (:104)
        dot_prod = np.matmul(post, np.transpose(SCLV))       # :120
        len_data = np.sum(post, axis=1)                      # :106
        abps = np.zeros(len(data))
        with np.errstate(invalid="ignore", divide="ignore"):
            for i in range(len(data)):
                abps[i] = dot_prod[i] / len_data[i]          # :123
            BR.append(np.mean(abps) / (BIN_RESOLUTION / 1000))   # :125
    if len(BR) == 2:
        BR.append(float("nan"))
    if verbose:
        print("BR results for " + train_or_test + " data (Flint, Sabes, Brochier): ", BR)
        print("Total power per channel: ", 0.96 + np.array(BR) * 0.02)
    return BR


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--root", required=True)
    ap.add_argument("--set", default="test", choices=["train", "test"])
    a = ap.parse_args(argv)
    run(a.root, a.set)


if __name__ == "__main__":
    main()
