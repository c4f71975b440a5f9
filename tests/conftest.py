import contextlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a box without a GPU or without the HIP library."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def tune():
    """tune(SLM_W4_MT=8, ...): force launch-shape knobs through the library's explicit setter
    (slm_tuning_set) for the duration of one test; the previous values come back afterwards."""
    from scalellm_amd import kernels
    with contextlib.ExitStack() as stack:
        def _set(**knobs):
            stack.enter_context(kernels.tuning(**knobs))
        yield _set


@pytest.fixture(autouse=True, scope="session")
def _no_ambient_tuning():
    """Tests must not depend on SLM_* variables that happen to be exported: drop whatever the
    library picked up from the environment at load time."""
    try:
        from scalellm_amd import kernels
        kernels.clear_tuning()
    except Exception:  # library not built: the tests that need it fail on their own
        pass
    yield
