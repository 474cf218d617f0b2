import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# Some tests import the reference verbatim from /root/reference (read-only by contract): neither this process nor any child
# it spawns (xdist workers, gloo ranks, compile workers) may leave byte-code caches there.
sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

# Kernel selection: nothing is pinned here.  With AA_FUSED unset the plan is created with aa_plan_options.fused_forward = 0
# ("automatic": the fused per-atom-tile forward whenever the graph allows it) -- the product's DEFAULT path, which is
# therefore what every test runs unless it opts into another one.  The model-level GPU modules additionally run every
# test three times through the `forward_mode` fixture below ("auto", "staged", "wide"), so all pipelines stay certified against
# the golden vectors, the oracle, finite differences, the ghost layout and the virial.
os.environ.pop("AA_FUSED", None)


@pytest.fixture(params=["auto", "staged", "wide"])
def forward_mode(request, monkeypatch):
    """AA_FUSED unset (automatic selection, the default) / AA_FUSED=0 (staged pipeline) for the plans a test creates; "wide": automatic
    selection with the two-waves-per-SIMD form of the fused forward (aa_fused8.hip) also on the small fixture boxes, where the default
    keeps the one-wave-per-SIMD kernel (it is what every box from 4 atoms per CU on runs)."""
    monkeypatch.delenv("AA_FUSED_NARROW", raising=False)
    if request.param == "staged":
        monkeypatch.setenv("AA_FUSED", "0")
    else:
        monkeypatch.delenv("AA_FUSED", raising=False)
        if request.param == "wide":
            monkeypatch.setenv("AA_FUSED_NARROW", "2")  # (eight-wave workgroups + the readout-reverse chain in the tail: what C4-sized boxes run)
    return request.param


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: ~280 tests, most of them kernels under the CPU emulation and multi-process gloo jobs) takes
    ~9 minutes on 6 workers and ~40 serially: when it is invoked without `-n`, run it on pytest-xdist workers.  Never for `-m gpu`
    (one device: the GPU tests run one at a time), never inside a worker, never when `-n` / `-p no:xdist` / AA_TEST_SERIAL=1 say otherwise."""
    opt = config.option
    if getattr(opt, "markexpr", "") != "not gpu" or os.environ.get("AA_TEST_SERIAL") == "1" or hasattr(config, "workerinput"):
        return None
    if not config.pluginmanager.hasplugin("xdist") or getattr(opt, "numprocesses", None) is not None or getattr(opt, "collectonly", False):
        return None
    opt.numprocesses = min(6, max(1, (os.cpu_count() or 1) - 2))
    opt.dist = "load"
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
