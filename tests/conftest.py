"""pytest configuration: registers the ``gpu`` marker and makes the repo root importable."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    # `pytest -m gpu --precision bf16x6` (or FFN_TEST_PRECISION=bf16x6): EVERY test runs with every
    # model it builds in the opt-in arithmetic mode -- the exact mode's bodies and tolerances
    # unchanged (FFN_PRECISION is read by the model constructors); tests marked `exact_only` state
    # why they need the exact-f32 kernels and are skipped with that reason
    parser.addoption("--precision", default=os.environ.get("FFN_TEST_PRECISION", "f32"),
                     choices=["f32", "bf16x3", "bf16x6"],
                     help="arithmetic mode of every model the tests build (default: exact f32)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "exact_only(reason, modes=...): the test pins something of the exact-f32 kernels "
                                       "themselves; skipped under --precision bf16x6 / bf16x3 (or only in `modes`)")
    mode = config.getoption("--precision")
    if mode != "f32":
        os.environ["FFN_PRECISION"] = mode


def pytest_report_header(config):
    return "arithmetic mode of the models under test: %s" % config.getoption("--precision")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    mode = config.getoption("--precision")
    if mode != "f32":
        for item in items:
            marker = item.get_closest_marker("exact_only")
            # (`modes=(...)`: only in those opt-in modes -- a 512-wide model falls back to exact f32 under
            # bf16x6, which has no kernels for it, but computes in bf16x3, which does)
            if marker is not None and mode in marker.kwargs.get("modes", (mode,)):
                why = marker.kwargs.get("reason") or (marker.args[0] if marker.args else "pins the exact-f32 kernels")
                item.add_marker(pytest.mark.skip(reason="--precision %s: %s" % (mode, why)))
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load
