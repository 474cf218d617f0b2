import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# Kernel selection is explicit in the tests: the staged pipeline unless a test opts into another path (monkeypatch of the
# AA_* variables that allegro_amd/_lib.py maps onto aa_plan_options).  Unset, AA_FUSED means "automatic" -- the fused
# forward whenever the graph allows it -- which tests/test_fused.py, the full-size block tests, bench.py's parity_sample and
# smoke() cover; pinning the staged pipeline here keeps its own kernels under test.
os.environ.setdefault("AA_FUSED", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
