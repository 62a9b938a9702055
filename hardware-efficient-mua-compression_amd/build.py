"""Build recipe for libmuahuff.so (hand-written gfx950 HIP kernels + C ABI).

hipcc cross-compiles for gfx950 without a GPU present.  The shared object is built in-tree
(next to this file) so that it travels with the source snapshot to the GPU box.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libmuahuff.so")
SOURCES = ["csrc/muahuff.hip"]
HEADERS = ["csrc/exports.map", "csrc/mh_kernels.hpp", "csrc/mh_device.hpp", "csrc/mh_codec2.hpp", "csrc/mh_layout.hpp",
           "csrc/mh_planner.hpp", "csrc/mh_analysis.hpp", "../include/muahuff.h"]
TUNING_SO = os.path.join(HERE, "libmuahuff_tuning.so")  # -DMH_TUNING: env knobs + ablation hook, tools/ only


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, tuning=False):
    """libmuahuff.so -- or, with tuning=True, libmuahuff_tuning.so: the same kernels plus the
    A/B knobs (MH_DEC_W, MH_DEC_NR, MH_WAVE_TASKS, mhdbg_set_ablation) that the production
    library does not contain; tools/ load it through _lib.use_library()."""
    so = TUNING_SO if tuning else SO
    if tuning:
        if not force and os.path.exists(so) and not any(
                os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(so) for f in SOURCES + HEADERS):
            return so
    elif not force and not stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]
    if tuning:
        cmd.append("-DMH_TUNING")   # (exports mhdbg_* besides the ABI)
    else:
        cmd.append("-Wl,--version-script=" + os.path.join(HERE, "csrc", "exports.map"))
    cmd += [os.path.join(HERE, s) for s in SOURCES] + ["-o", so]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return so


def build_example(force=False):
    """examples/abi_roundtrip: a plain-C client of include/muahuff.h (gcc, HIP runtime only)."""
    src = os.path.join(ROOT, "examples", "abi_roundtrip.c")
    exe = os.path.join(ROOT, "examples", "abi_roundtrip")
    if not force and os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(src), os.path.getmtime(SO)):
        return exe
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), src, "-o", exe,
                           "-L" + HERE, "-lmuahuff", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                           "-Wl,-rpath,$ORIGIN/../" + os.path.basename(HERE), "-Wl,-rpath," + os.path.join(rocm, "lib")])
    return exe


if __name__ == "__main__":
    print(build(force=True, verbose=True))
