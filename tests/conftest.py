import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Leave the gradient-parity margins of this session (tests/golden_util.py:MARGINS) where a GPU run's results are
    collected; tools/parity_margins.py renders them."""
    import json
    try:
        import golden_util as gu
    except Exception:
        return
    if not gu.MARGINS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump(gu.MARGINS, f, indent=1, sort_keys=True)
    except OSError:
        pass
