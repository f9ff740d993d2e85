"""tts.cpp_amd — MI355X-native hot path for TTS.cpp's Parler-TTS generate():
hand-written HIP kernels behind a C ABI (include/tts_hip.h), a C++ host runner that mirrors
the reference's tts_generation_runner API (host/), and Python plumbing for tests and bench.

Python here is plumbing only: ctypes bindings (hip.py), a GGUF reader/writer (gguf.py) and the
synthetic-weight generator (synth.py).  Nothing in this package imports oracle/.
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)

from . import gguf, synth  # noqa: E402,F401
from .hip import HipEngine, HipError, lib_path, load_lib  # noqa: E402,F401
