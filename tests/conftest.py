import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The product routes launches of fewer than 1500 elements to the generic LDS kernel (lower latency on small meshes); the
# parity tests run on small meshes and must exercise the one-wave-per-element kernel the benchmark uses, so the threshold
# is 0 here.  tests/test_gpu_edge_cases.py::test_small_launch_routing covers the default routing.
os.environ.setdefault("L3K_GENERIC_BELOW", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))

    return load
