import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()


def pytest_collection_finish(session):
    """GPU runs: (1) the CPU oracle's convolutions run on at most 16 threads - on the 128-core GPU hosts oneDNN's small convolutions are twice as fast
    with 16 threads as with all of them (bench.py's thread probe); (2) when tests/test_gpu_variants.py is part of the run its subprocess pool starts NOW
    and works in the background, one variant at a time, while the main process runs the parity tests."""
    import torch
    if not torch.cuda.is_available() or session.config.option.collectonly:
        return
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 8)))
    if any(it.nodeid.startswith("tests/test_gpu_variants.py") or "test_gpu_variants.py::" in it.nodeid for it in session.items):
        import test_gpu_variants as V
        V.start_pool()
