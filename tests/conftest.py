import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import tts_cpp_amd  # noqa: E402,F401  (registers the tts.cpp_amd/ directory as package tts_cpp_amd)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _have_gpu():
    try:
        from tts_cpp_amd import hip
        return hip.load_lib().tts_hip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


def pytest_collection_modifyitems(config, items):
    # GPU tests must fail loudly (not skip) when selected with -m gpu on a box without a device;
    # in a plain CPU run (-m "not gpu") they are deselected by the marker expression.
    pass
