import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_present():
    # device_count() does not initialise HIP on this image (safe before fork/spawn)
    try:
        import torch

        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    from tests import helpers

    c = helpers.WINNER_CENSUS
    if c["exact"] + c["gap"]:
        terminalreporter.write_line(
            f"float winner rule: {c['exact']} winners were the oracle's arg-max, {c['gap']} were accepted through the gap branch "
            f"(largest gap used: {c['largest_gap_used']:.2e} of max(1, |ucb|); allowed 2e-5)")
