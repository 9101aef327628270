"""Randomised GPU-vs-oracle sweep as part of the GPU suite (tools/fuzz_gpu.py, fixed seed)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_sweep_fixed_seed():
    env = dict(os.environ, FUZZ_CASES="50", FUZZ_SEED="20260101")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 bad" in out.stdout
