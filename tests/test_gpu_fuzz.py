"""Randomised GPU-vs-oracle sweep as part of the GPU suite (tools/fuzz_gpu.py): the fixed regression seed of rounds 1-4
and ONE seed that changes every round (its number is in the test id, so a failure names the sweep to re-run)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = 6  # bump with the round: the second sweep below is new cases every round
SEEDS = [pytest.param(20260101, id="fixed-seed20260101"), pytest.param(20260100 + 7 * ROUND, id=f"round{ROUND}-seed{20260100 + 7 * ROUND}")]


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_sweep(seed):
    env = dict(os.environ, FUZZ_CASES="50", FUZZ_SEED=str(seed))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 bad" in out.stdout
