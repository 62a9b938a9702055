"""Host-side pieces of the driver ports that need no GPU."""
import numpy as np

import oracle
from muahuff.drivers import _sweep


def test_read_directories_reference_format(tmp_path):
    (tmp_path / "directories.txt").write_text(
        "%%% Notes %%%\n% key = 'ignored comment'\nhome_directory = 'D:\\\\x y\\\\z'\n"
        "Formatted_data_path = '/a b/Formatted'\n\nSCLV_path = '/s'\nBR_no_sort_results = '/r n'\n")
    d = _sweep.read_directories(str(tmp_path))
    assert d["Formatted_data_path"] == "/a b/Formatted" and d["SCLV_path"] == "/s"
    assert d["BR_no_sort_results"] == "/r n" and "key" not in d


def test_split_consumes_rng_like_the_reference():
    """Same permutation calls, same order, same cap/rounding as the NumPy oracle (which is pinned
    to the reference scripts by tests/test_oracle_golden.py)."""
    n_per = [7, 9]
    base = np.cumsum([0] + n_per)
    flat = [np.full(3, i, np.uint8) for i in range(sum(n_per))]
    all_data = [flat[:7], flat[7:]]
    np.random.seed(99)
    tr, va = _sweep.split_indices(n_per, base, 6, 50)
    a = np.random.rand()
    np.random.seed(99)
    otr, ova = oracle.np_.split_channels(all_data, 6, 50)
    b = np.random.rand()
    assert a == b  # identical RNG consumption
    assert [int(x[0]) for x in otr] == tr and [int(x[0]) for x in ova] == va
