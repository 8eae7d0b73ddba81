import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    try:  # the GPU box has 256 host cores: torch-CPU oracles crawl when oversubscribed
        import torch

        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:  # noqa: BLE001
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def lib_built():
    """Build the HIP library once per session if it is not there (hipcc cross-compiles on CPU)."""
    so = os.path.join(ROOT, "probpose_code_amd", "libprobpose_mi355x.so")
    if not os.path.exists(so):
        import __graft_entry__ as g

        g.build()
    return so


@pytest.fixture(autouse=True)
def _library_options_restored():
    """Library options (pp_set_option) are process-global: whatever a test switched - also one that failed half way - is back at its
    first value before the next test runs, so no later test silently runs another kernel."""
    yield
    mod = sys.modules.get("probpose_code_amd._lib")
    if mod is not None:
        mod.restore_options()
