"""CPU-side checks of the product: tables, the functions_1 drop-in, and that libmuahuff.so
loads and exports exactly what include/muahuff.h declares.  No GPU, no compute calls."""
import ctypes as ct
import os
import re

import numpy as np
import pytest

import muahuff
from muahuff import _lib, sclv
from muahuff import functions_1 as f1
from tests import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "muahuff.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mh_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    raw = ct.CDLL(_lib.SO)
    for name in declared:
        assert hasattr(raw, name), name
    assert _lib.lib().mh_version() == 101


def test_geometry_constants_match_header():
    hdr = open(os.path.join(ROOT, "include", "muahuff.h")).read()
    for name, val in (("MH_PIECE", _lib.PIECE), ("MH_LANES", _lib.LANES), ("MH_ROWS", _lib.ROWS)):
        assert int(re.search(r"#define %s (\d+)" % name, hdr).group(1)) == val
    assert _lib.CHUNK == 16384 and _lib.HDR_WORDS == 32


def test_sclv_tables_equal_reference_pickles():
    gold = helpers.sclv_tables()
    for S in range(2, 11):
        assert np.array_equal(sclv.table(S), gold[S]), S


def test_codebook_host_helper():
    assert sclv.codewords([1, 2, 2]) == ["0", "10", "11"]  # test_chosen_system.py:26
    assert sclv.codewords([1, 2, 3, 4, 4]) == ["0", "10", "110", "1110", "1111"]
    with pytest.raises(muahuff.MuaHuffError) as e:
        sclv.codebook([1, 1, 2])
    assert e.value.code == _lib.ERR_SCLV


def test_approx_sort_perm_abi_and_dropin():
    t = helpers.tables()
    for S in range(2, 11):
        for p in range(S):
            want = t["approx_sort"][str(S)][p]
            idx = np.zeros(S, np.uint8)
            _lib.check(_lib.lib().mh_approx_sort_perm(S, p, idx.ctypes.data))
            assert list(idx) == want
            hist = np.ones(S, dtype=np.int64)
            hist[p] = 7
            got_idx, got_sorted = f1.approx_sort(hist)
            assert list(got_idx) == want and got_idx.dtype == int
            assert np.array_equal(got_sorted, hist[want])
    for rec in t["approx_sort_ties"]:
        got_idx, _ = f1.approx_sort(np.array(rec["hist"]))
        assert list(got_idx) == rec["idx"]


def test_online_histogram_dropin():
    for rec in helpers.tables()["cutoff"]:
        x = np.array(rec["x"], dtype=np.uint8)
        hist, i = f1.online_histogram_w_sat_based_nb_of_samples(x, rec["cutoff"], rec["S"] - 1)
        assert i == rec["i"]
        assert list(x) == rec["x_after"]
        assert sum(hist.values()) == i and "0" in hist
        clipped = np.minimum(np.array(rec["x"][:i]), rec["S"] - 1)
        for k, v in hist.items():
            assert v == int((clipped == int(k)).sum())
    with pytest.raises(IndexError):
        f1.online_histogram_w_sat_based_nb_of_samples(np.zeros(0, np.uint8), 4, 2)


def test_module_surface_matches_reference():
    # from functions_1 import * must give the three functions plus np and math
    ns = {}
    exec("from muahuff.functions_1 import *", ns)
    for name in ("bin_MUA_data", "online_histogram_w_sat_based_nb_of_samples", "approx_sort", "np", "math"):
        assert name in ns


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from muahuff import codec
    with pytest.raises(muahuff.MuaHuffError):
        codec.Plan(np.zeros(1, np.uint64), np.full(1, 100, np.uint64), 3, 6, 1, 0, sclv.table(3))
    # and straight through the C ABI: plan creation refuses, it does not fall back
    h = ct.c_void_p()
    off, ln, tab = np.zeros(1, np.uint64), np.full(1, 100, np.uint64), sclv.table(3)
    rc = _lib.lib().mh_plan_create(ct.byref(h), off.ctypes.data, ln.ctypes.data, 1, 3, 6, 1, 0,
                                   tab.ctypes.data, 1, 8)
    assert rc == _lib.ERR_NO_DEVICE and not h.value


def test_plan_argument_errors_without_gpu():
    h = ct.c_void_p()
    off, tab = np.zeros(1, np.uint64), sclv.table(3)
    rc = _lib.lib().mh_plan_create(ct.byref(h), off.ctypes.data, np.zeros(1, np.uint64).ctypes.data, 1, 3, 6,
                                   1, 0, tab.ctypes.data, 1, 8)
    assert rc == _lib.ERR_EMPTY_CHANNEL  # the reference raises IndexError on an empty channel
    rc = _lib.lib().mh_plan_create(ct.byref(h), off.ctypes.data, np.ones(1, np.uint64).ctypes.data, 1, 11, 6,
                                   1, 0, tab.ctypes.data, 1, 8)
    assert rc == _lib.ERR_ARG
    bad = np.array([[2, 1, 2]], np.uint8)
    rc = _lib.lib().mh_plan_create(ct.byref(h), off.ctypes.data, np.ones(1, np.uint64).ctypes.data, 1, 3, 6,
                                   1, 0, bad.ctypes.data, 1, 8)
    assert rc == _lib.ERR_SCLV


def test_container_file_roundtrip_without_gpu(tmp_path):
    from muahuff import container_io as cio
    c = cio.Compressed(cio.make_header(3, 6, 1, 2, 2, [[1, 2, 2]]), np.array([100, 7], np.uint64),
                       np.array([1, 0], np.uint8), np.array([0, 0], np.uint8), np.array([0, 1], np.uint8),
                       np.array([51, 0], np.uint64), np.array([35], np.uint64), np.arange(35, dtype=np.uint32))
    fn = tmp_path / "x.mhf"
    cio.save(fn, c)
    d = cio.load(fn)
    assert d.header["S"] == 3 and d.header["sclv"] == [[1, 2, 2]] and d.header["format_revision"] == 2
    for name in ("ch_len", "peak", "enc", "skipped", "ch_bits", "seg_words", "payload"):
        assert np.array_equal(getattr(c, name), getattr(d, name)), name
    assert d.container_bits == 35 * 32 and d.payload_bits == 51
    with pytest.raises(ValueError):
        cio.read(__import__("io").BytesIO(b"NOTMAGIC" + b"\0" * 16))


def test_design_point_table_matches_reference_formula():
    from muahuff import analysis
    z, params = helpers.sweep()
    res = {}
    for key in z.files:
        if key.startswith("approx/") and key.endswith("/BRs"):
            S, BP, cv = [int(t[1:] if t[0] == "S" else t[2:]) for t in key.split("/")[1].split("_")]
            res[(S, BP, cv)] = {"stored_all_var_BRs": z[key]}
    tab = analysis.design_point_table(res)
    # first row: BP=10, S=2, hist 2^2, 1 encoder, mean over channels and CVs (nan if any nan)
    a = np.mean([np.mean(z["approx/S2_BP10_CV%d/BRs" % cv][0][0]) for cv in (1, 2)])
    assert tab[0, :4].tolist() == [10, 2, 2, 1] and helpers.same_float(tab[0, 4], a)
    assert tab.shape[1] == 6 and tab.shape[0] == 2 * 94 * 9


def test_missing_library_fails_loudly(tmp_path):
    """No libmuahuff.so -> ImportError at first use, never a silent CPU path."""
    import subprocess
    import sys
    code = ("import muahuff\n"
            "try:\n    muahuff._lib.lib()\nexcept ImportError as e:\n    print('LOUD', 'no CPU fallback' in str(e))\n")
    env = dict(os.environ, MUAHUFF_LIB=str(tmp_path / "nope.so"), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert "LOUD True" in out.stdout, out.stdout + out.stderr


def test_plain_c_client_builds_and_fails_loudly_without_gpu():
    """examples/abi_roundtrip.c: include/muahuff.h is valid C11 and links from gcc; with no
    device the very first call reports MH_ERR_NO_DEVICE (exit 3), it does not fall back."""
    import subprocess
    import torch
    b = __import__("importlib").import_module("hardware-efficient-mua-compression_amd.build")
    exe = b.build_example()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "no HIP device" in r.stderr, r.stderr
