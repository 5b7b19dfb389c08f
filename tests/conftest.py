import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLD = ROOT / "tests" / "golden"
# tests pass GradCache chunk sizes they mean literally (chunk-invariance, multi-chunk paths); the auto re-chunking of
# contrastors_amd.loss.effective_chunk has its own test
os.environ.setdefault("CX_GRADCACHE_CHUNK", "exact")
# ... and the two-pass GradCache schedule unless a test asks for resident activations (tests/test_loss_gpu.py)
os.environ.setdefault("CX_GRADCACHE_RESIDENT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    return lambda name: np.load(GOLD / f"{name}.npz", allow_pickle=False)
