"""ctypes binding of libmuahuff.so (include/muahuff.h).  Fails loudly: there is no CPU path."""
import ctypes as ct
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmuahuff.so")

MH_OK = 0
ERR_ARG, ERR_EMPTY_CHANNEL, ERR_SCLV, ERR_CAPACITY, ERR_HIP, ERR_NO_DEVICE, ERR_STREAM = -1, -2, -3, -4, -5, -6, -7

MODE_NOSORT, MODE_APPROX = 0, 1
WIN_REF_HALF, WIN_REF_HALF_TRUNC, WIN_AFTER_CAL, WIN_FULL = 0, 1, 2, 3
WIN_REV2_SEGMENTS = 0x100  # OR-ed into a window rule: the segment directory of container format revision 2 (no head segments)

PIECE, LANES, ROWS = 16, 64, 16
SUB = PIECE * ROWS
CHUNK = SUB * LANES
HDR_WORDS = LANES // 2


class MuaHuffError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libmuahuff error %d: %s" % (code, msg))
        self.code = code


class PlanInfo(ct.Structure):
    _fields_ = [(n, ct.c_uint32) for n in ("C", "S", "h", "mode", "window", "K", "seg_chunks", "maxlen")] + \
               [(n, ct.c_uint64) for n in ("n_segments", "payload_cap_words", "window_samples", "n_skipped")]


_vp, _u32, _u64, _int = ct.c_void_p, ct.c_uint32, ct.c_uint64, ct.c_int

# every symbol include/muahuff.h declares, with its prototype
PROTOTYPES = {
    "mh_version": (_int, []),
    "mh_last_error": (ct.c_char_p, []),
    "mh_device_info": (_int, [_int, ct.POINTER(_int), ct.POINTER(_u64), ct.c_char_p, _int, ct.c_char_p, _int]),
    "mh_codebook": (_int, [_vp, _int, _vp, _vp]),
    "mh_approx_sort_perm": (_int, [_int, _int, _vp]),
    "mh_plan_create": (_int, [ct.POINTER(_vp), _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _u32]),
    "mh_plan_create_packed": (_int, [ct.POINTER(_vp), _vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _u64]),
    "mh_plan_destroy": (_int, [_vp]),
    "mh_plan_info": (_int, [_vp, ct.POINTER(PlanInfo)]),
    "mh_plan_segments": (_int, [_vp, _vp, _vp, _vp, _vp]),
    "mh_measure": (_int, [_vp] * 10),
    "mh_encode": (_int, [_vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mh_encode_preset": (_int, [_vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp]),
    "mh_plan_query": (_int, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _u32, ct.POINTER(PlanInfo), _vp, _vp, _vp, _vp, _u64]),
    "mh_decode": (_int, [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    "mh_decode_status": (_int, [_vp, ct.POINTER(_u32), _vp]),
    "mh_validate_stream": (_int, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _u32, _u32, _vp, _u64, _vp, _u64, _vp, _vp]),
    "mh_power_draws": (_int, [_vp, _u32, _vp, _u32, _u64, ct.c_double, ct.c_double, ct.c_double, _vp, _u64, _vp]),
    "mh_reduce_rows": (_int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    "mh_compact": (_int, [_vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp]),
    "mh_synth_poisson": (_int, [_vp, _vp, _vp, _u32, _u64, _vp, _u64, _vp]),
    "mh_rebin": (_int, [_vp, _vp, _vp, _u32, _u64, _u32, _int, _vp, _vp, _vp]),
    "mh_deinterleave": (_int, [_vp, _u64, _u32, _vp, _vp, _vp]),
    "mh_deinterleave_packed": (_int, [_vp, _u64, _u32, _u32, _vp, _vp, _u64, _vp]),
    "mh_interleave": (_int, [_vp, _vp, _u64, _u32, _vp, _vp]),
    "mh_sweep_create": (_int, [ct.POINTER(_vp), _vp, _vp, _u32, _vp, _u32]),
    "mh_sweep_destroy": (_int, [_vp]),
    "mh_sweep_info": (_int, [_vp, ct.POINTER(_u32), _vp]),
    "mh_sweep_run": (_int, [_vp, _vp, _vp, _vp]),
}

_lib = None


def use_library(path):
    """Load the kernels from another build of libmuahuff.so (same-box A/B runs in tools/).  Must be
    called before the first use; there is no environment override."""
    global SO, _lib
    if _lib is not None:
        raise RuntimeError("libmuahuff.so is already loaded")
    SO = str(path)


def lib():
    """Load libmuahuff.so.  Raises if it has not been built: the product has no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise ImportError(
                "libmuahuff.so is missing (%s). Build it with `python __graft_entry__.py` or "
                "`python hardware-efficient-mua-compression_amd/build.py`; there is no CPU fallback." % SO)
        # torch first: its wheel bundles the HIP runtime, and a process must not end up with two of them -- a
        # libmuahuff.so loaded before torch binds the system libamdhip64, torch then brings its own, and the
        # library's runtime finds no device (seen as MH_ERR_NO_DEVICE when build() and smoke() share a process)
        import torch  # noqa: F401
        l = ct.CDLL(SO)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _lib = l
    return _lib


def check(rc):
    if rc != MH_OK:
        raise MuaHuffError(rc, lib().mh_last_error().decode(errors="replace"))
    return rc


def device_info(device=0):
    cu, mem = _int(0), _u64(0)
    name, arch = ct.create_string_buffer(256), ct.create_string_buffer(256)
    check(lib().mh_device_info(device, ct.byref(cu), ct.byref(mem), name, 256, arch, 256))
    return dict(cu_count=cu.value, hbm_bytes=mem.value, name=name.value.decode(), arch=arch.value.decode())
