import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]

    return load


def pytest_terminal_summary(terminalreporter):
    """Achieved gradient parity (worst relative l2 error / worst cosine per whole-model test), next to the bounds."""
    from tests import _util
    if not _util.ACHIEVED:
        return
    terminalreporter.write_sep("-", f"gradient parity achieved (bounds: rel-l2 <= {_util.REL_L2_MAX}, cosine >= {_util.COS_MIN})")
    for k, (a, b) in sorted(_util.ACHIEVED.items()):
        terminalreporter.write_line(f"{k}: {a:.3e} / {b:.6f}")
