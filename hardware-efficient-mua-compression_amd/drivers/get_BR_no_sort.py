"""Train/validate sweep without sorting (identity mapper): the GPU counterpart of the reference's
``Compressing data/get_BR_no_sort.py`` (same inputs via ``directories.txt``, same
``BRs_S_<S>_BP_<BP>_CV_<cv>.pkl`` outputs under ``BR_no_sort_results``).

    python -m muahuff.drivers.get_BR_no_sort --root <dir with directories.txt> [--seed N]
"""
from . import _sweep


def run(root_directory, nb_CV_iterations=30, how_many_channels_Sabes=2000, **kw):
    return _sweep.run(root_directory, False, nb_CV_iterations, how_many_channels_Sabes, **kw)


def main(argv=None):
    import argparse

    import numpy as np
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--root", required=True)
    ap.add_argument("--cv", type=int, default=30, help="nb_CV_iterations (the reference runs 1..cv-1)")
    ap.add_argument("--seed", type=int, default=None, help="np.random.seed for a reproducible channel split")
    ap.add_argument("--python-floats", action="store_true",
                    help="store bit rates as Python floats (same values, ~90x faster to pickle than np.float64 scalars)")
    a = ap.parse_args(argv)
    if a.seed is not None:
        np.random.seed(a.seed)
    run(a.root, nb_CV_iterations=a.cv, python_floats=a.python_floats)


if __name__ == "__main__":
    main()
