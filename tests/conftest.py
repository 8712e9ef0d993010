import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


_MEASURED = {}


@pytest.fixture
def measured(request):
    """Record the values a parity test measured next to the bar it asserts: printed (`pytest -rA` / `-s` shows them) and, when
    gpurun_out/ exists, written to gpurun_out/measured_parity.json at the end of the session - bars are pinned from these numbers
    (no bar looser than 3x what was measured), so a regression inside a loose bar cannot hide."""
    def rec(name, value, bar=None):
        key = f"{request.node.name}::{name}"
        _MEASURED[key] = {"value": float(value), "bar": None if bar is None else float(bar)}
        print(f"MEASURED {key} = {float(value):.6g}" + ("" if bar is None else f"   (bar {float(bar):.6g})"))
    return rec


def pytest_sessionfinish(session, exitstatus):
    out = os.path.join(ROOT, "gpurun_out")
    if _MEASURED and os.path.isdir(out):
        import json
        with open(os.path.join(out, "measured_parity.json"), "w") as fh:
            json.dump(_MEASURED, fh, indent=1, sort_keys=True)
