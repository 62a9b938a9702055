"""The driver's entry points on the GPU box, as one process and as two."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    return subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)


def test_build_then_smoke_in_one_process():
    """build() loads the library before anything has imported torch; smoke() must still find the GPU
    (one HIP runtime per process: _lib.lib() imports torch first)."""
    r = _run(["__graft_entry__.py", "smoke"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout and "bit-exact vs oracle" in r.stdout


def test_smoke_alone():
    r = _run(["-c", "import __graft_entry__ as g; g.smoke()"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke ok" in r.stdout
