import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "refonly: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/smpl_sim")
    for it in items:
        if "refonly" in it.keywords and not have_ref:
            it.add_marker(pytest.mark.skip(reason="/root/reference not present"))
