import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) on a box without a device, so a plain `pytest` works on the CPU container.  Asking
    the library for the device count initialises nothing on a GPU box beyond what the tests do themselves."""
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items:
        return
    try:
        import zvdb_amd
        have = zvdb_amd.device_count() > 0
    except Exception:  # library missing or broken: do NOT hide that behind skips -- let the GPU tests fail loudly
        have = True
    if not have:
        skip = pytest.mark.skip(reason="no MI355X visible (GPU tests run on the GPU box: pytest -m gpu)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc
