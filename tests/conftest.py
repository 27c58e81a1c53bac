import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib_path():
    """Host-emulator build of the kernel sources (tests only, see tests/emu/hip_emu.h)."""
    from auto_avsr_amd import build

    return build.build_emu()


BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def dev(request, emu_lib_path):
    """Device on which the kernels run: 'cpu' through the emulator build (CPU suite) or
    'cuda' through the shipped libavsr_hip.so (GPU suite, -m gpu)."""
    from auto_avsr_amd import _lib

    if request.param == "emu":
        _lib._install_for_tests(emu_lib_path)
        yield torch.device("cpu")
        _lib._lib = None
    else:
        assert torch.cuda.is_available(), "GPU suite needs a GPU"
        _lib._lib = None
        L = _lib.lib()
        assert not L.is_emulator and L.path.endswith("libavsr_hip.so")
        yield torch.device("cuda:0")
