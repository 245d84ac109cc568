import os
import sys

import pytest

# Some tests load modules from the read-only reference checkout (/root/reference): never leave __pycache__/*.pyc in it.  Set before
# any such import and inherited by the forked sampler workers of tests/test_reference_agent.py.
sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "refonly: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/smpl_sim")
    have_gpu = None
    for it in items:
        if "refonly" in it.keywords and not have_ref:
            it.add_marker(pytest.mark.skip(reason="/root/reference not present"))
        if "gpu" in it.keywords:
            if have_gpu is None:
                import torch
                have_gpu = torch.cuda.is_available()
            if not have_gpu:                                 # a plain `pytest tests` in the GPU-less container stays green
                it.add_marker(pytest.mark.skip(reason="no ROCm GPU in this environment (run on the MI355X box via gpurun)"))


@pytest.fixture()
def emu_backend(monkeypatch):
    """Point the package at the wavefront-emulator build of its C ABI (tests/wave_emu) with host tensors, for this test only.
    Everything is patched from here: smplsim_amd has no backend switch of its own."""
    import torch
    from smplsim_amd import _lib, batch
    from wave_emu import emu
    monkeypatch.setattr(_lib, "_LIB", emu.lib())
    monkeypatch.setattr(batch, "_shard_device", lambda index: torch.device("cpu"))
    monkeypatch.setattr(batch, "_launch_stream", lambda device: None)
    yield emu.lib()
