import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


# Order of the `-m gpu` files under `pytest -x`: single-process kernel parity first (ops -> engine -> vision -> fp32 encoder -> VCR ->
# fp16 build), the multi-process data-parallel file LAST -- a timing-dependent failure of a two-rank run must not stop the run in
# front of the kernel-parity evidence (round 4: one red DDP test hid 260 others).
_GPU_FILE_ORDER = ["test_ops_gpu", "test_engine_gpu", "test_vision_gpu", "test_f32_encoder_gpu", "test_vcr_gpu", "test_f16_build_gpu",
                   "test_dp_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _GPU_FILE_ORDER:
            return (1, _GPU_FILE_ORDER.index(name))
        return (0, 0) if "dp" not in name else (1, len(_GPU_FILE_ORDER))
    items.sort(key=rank)          # (stable: the order inside a file is kept)
