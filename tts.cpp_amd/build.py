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


UNITS = ["shim_core", "shim_decoder", "shim_codec", "shim_llama", "shim_qtile"]
# per-unit code-generation switches (the reasons are in the unit's header comment)
UNIT_FLAGS = {"shim_qtile": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _deps(depfile, fallback):
    """the headers a unit really includes (hipcc -MMD of its last build), or every header when there is no record yet"""
    try:
        txt = open(depfile).read().replace("\\\n", " ")
        files = [f for f in txt.split(":", 1)[1].split() if os.path.exists(f)]
        return files or fallback
    except (OSError, IndexError):
        return fallback


def build_hip(force=False, verbose=True):
    """libtts_hip.so: the C-ABI HIP shim (include/tts_hip.h).  The translation units under csrc/ compile in parallel into
    csrc/obj/*.o (each rebuilt only when it or a header is newer), then link."""
    csrc = os.path.join(PKG_DIR, "csrc")
    objdir = os.path.join(csrc, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")] + [os.path.join(REPO_ROOT, "include", "tts_hip.h")]
    out = os.path.join(PKG_DIR, "libtts_hip.so")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    jobs = []
    for u in UNITS:
        src, obj, dep = os.path.join(csrc, u + ".hip"), os.path.join(objdir, u + ".o"), os.path.join(objdir, u + ".d")
        if force or _newer(obj, _deps(dep, [src] + headers)):
            cmd = [HIPCC] + flags + UNIT_FLAGS.get(u, []) + ["-MMD", "-MF", dep, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    objs = [os.path.join(objdir, u + ".o") for u in UNITS]
    if force or jobs or _newer(out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
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
