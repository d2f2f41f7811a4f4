import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hip():
    """Bind the process to GPU 0 through the C ABI; fails loudly if libpvhip.so or the GPU is missing."""
    from pyopenvino_amd import device
    device.load_library()
    assert device.device_count() >= 1, 'no HIP device visible'
    device.init(0)
    return device


@pytest.fixture(autouse=True)
def _pvhip_settings_follow_the_environment():
    """libpvhip parses the PVHIP_* variables once; tests flip them through helpers.setenv (which re-reads them).  Autouse
    fixtures are set up first and torn down last, i.e. after monkeypatch has restored the environment: re-read it then."""
    yield
    from pyopenvino_amd import device
    if device._lib is not None:
        device.reload_settings()
