"""Build recipe for the native pieces (hipcc cross-compiles gfx950 without a GPU)."""
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=True):
    """libtts_hip.so: the C-ABI HIP shim (include/tts_hip.h)."""
    csrc = os.path.join(PKG_DIR, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(REPO_ROOT, "include", "tts_hip.h")]
    out = os.path.join(PKG_DIR, "libtts_hip.so")
    if force or _newer(out, srcs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", "-o", out, os.path.join(csrc, "tts_hip.hip")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return out


def build_host(force=False, verbose=True):
    """libtts.so + tools: the C++ host runner mirroring the reference API (host/)."""
    host = os.path.join(PKG_DIR, "host")
    if not os.path.isdir(host) or not os.path.exists(os.path.join(host, "Makefile")):
        return None
    cmd = ["make", "-C", host, "-j8"] + (["-B"] if force else [])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return os.path.join(host, "libtts.so")


def build_all(force=False):
    build_hip(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
