#!/usr/bin/env python3
"""The reference's perf_battery protocol (examples/perf_battery/perf_battery.cpp:100-117: the 30 Harvard sentences, one generate() each,
wall time per sentence, RTF = generation ms / audio ms) on a synthetic Parler-TTS-Mini fp16 GGUF.  Runs the reference's OWN harness,
compiled unchanged through the compat/ overlay into oracle/_ref/perf_battery_ref (oracle/Makefile), when it is there, else the engine's
host/perf_battery (same protocol).  Random weights never emit EOS, so every sentence runs to max_generation (2580 positions = 29.8 s of
audio): the numbers are batch-1 decode + codec throughput, not a statement about speech."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: F401
from tts_cpp_amd import gguf, synth

path = os.path.join(tempfile.gettempdir(), "parler_mini_f16_synth.gguf")
t0 = time.perf_counter()
synth.build(synth.parler_mini(weight_type=gguf.F16)).write_gguf(path)
print(f"synthetic GGUF written in {time.perf_counter() - t0:.1f}s ({os.path.getsize(path) / 1e9:.2f} GB)", flush=True)
ref = os.path.join(ROOT, "oracle", "_ref", "perf_battery_ref")
exe = ref if os.path.exists(ref) else os.path.join(ROOT, "tts.cpp_amd", "host", "perf_battery")
print("harness:", os.path.relpath(exe, ROOT), "(the reference's perf_battery.cpp, unchanged)" if exe == ref else "(engine's tool)", flush=True)
t0 = time.perf_counter()
out = subprocess.run([exe, "--model-path", path] + sys.argv[1:], capture_output=True, text=True)
print(out.stdout[-1500:])
print(out.stderr[-800:], file=sys.stderr)
print(f"wall {time.perf_counter() - t0:.1f}s rc={out.returncode}")
os.unlink(path)
