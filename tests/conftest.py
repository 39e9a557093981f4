import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


@pytest.fixture(scope="session")
def report():
    """report(name, value): appends the MEASURED error of a tolerance check to gpurun_out/test_errors.txt
    (when that directory exists) so that tolerances can be kept at a small multiple of what the kernels
    actually deliver instead of a guess."""
    out = os.path.join(ROOT, "gpurun_out")

    def rec(name, value):
        if os.path.isdir(out):
            with open(os.path.join(out, "test_errors.txt"), "a") as f:
                f.write("%s %.4e\n" % (name, float(value)))
        return float(value)

    return rec
