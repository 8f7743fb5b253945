import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle on a many-core GPU host: torch's default of one thread per logical core is 10x slower than 16
    import torch

    torch.set_num_threads(min(16, os.cpu_count() or 1))


def pytest_collection_modifyitems(config, items):
    """``gpu``-marked tests need a device: skip them (instead of failing on 'No HIP GPUs are available') wherever
    torch sees none, whatever -m expression was given."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (run on the MI355X box: pytest -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "beat_this"))
