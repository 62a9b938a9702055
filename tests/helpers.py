"""Shared test helpers: golden-fixture loading (data only; nothing here reads /root/reference)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tables():
    with open(os.path.join(GOLDEN, "tables.json")) as f:
        return json.load(f)


def sclv_tables():
    return {int(S): np.array(rows, dtype=np.uint8) for S, rows in tables()["sclv"].items()}


def per_channel():
    z = np.load(os.path.join(GOLDEN, "per_channel.npz"))
    recs = json.loads(bytes(z["records"]).decode())
    off = np.concatenate([[0], np.cumsum(z["lens"])])
    chans = [z["data"][off[i]:off[i + 1]].copy() for i in range(len(z["lens"]))]
    return chans, recs


def unpack_dataset(z, prefix):
    """inverse of make_golden.pack_dataset -> all_binned_data[BP][dataset][channel]"""
    data, lens = z[prefix + "/data"], z[prefix + "/lens"]
    n_per = z[prefix + "/n_per_dataset"]
    nds = int(z[prefix + "/n_datasets"][0])
    off = np.concatenate([[0], np.cumsum(lens)])
    chans = [data[off[i]:off[i + 1]].copy() for i in range(len(lens))]
    out, k, j = [], 0, 0
    for _bp in range(len(n_per) // nds):
        bp = []
        for _d in range(nds):
            bp.append(chans[k:k + int(n_per[j])])
            k += int(n_per[j])
            j += 1
        out.append(bp)
    return out, [int(v) for v in z[prefix + "/bin_vector"]]


def sweep():
    z = np.load(os.path.join(GOLDEN, "sweep.npz"))
    params = json.loads(bytes(z["params"]).decode())
    return z, params


def chosen_system():
    with open(os.path.join(GOLDEN, "chosen_system.json")) as f:
        return [float.fromhex(v) if v != "nan" else float("nan") for v in json.load(f)["BR_hex"]]


def same_float(a, b):
    """bit-exact float64 equality, nan == nan"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def chunk_header(words):
    """Independent parser of a chunk header (container format revision 2):
    words = uint32 array starting at the chunk.  Returns (sub-stream bit lengths [64], header words)."""
    import numpy as np
    w0 = int(words[0])
    mn, w = w0 & 0xFFF, (w0 >> 12) & 15
    hw = (16 + 64 * w + 31) // 32
    bits = 0
    for i in range(hw):                      # the header as one little-endian bit string
        bits |= int(words[i]) << (32 * i)
    lens = np.array([mn + ((bits >> (16 + l * w)) & ((1 << w) - 1)) for l in range(64)], dtype=np.int64)
    return lens, hw
