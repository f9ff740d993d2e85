#!/usr/bin/env python3
"""Malformed GGUF files against the host's reader and loader (host/gguf.cpp, host/loaders.cpp): truncations at every kind of boundary, byte flips in
the metadata (counts, lengths, types, offsets, the tokenizer arrays, tensor infos), absurd counts.  Every case must come back as an error code or a
clean load attempt — never a crash.  Usage: fuzz_gguf.py <libtts.so> [seed] [flips]; run under ASan + UBSan builds of the library at the end of round
4 (3 seeds x 766 cases: no report), and in the CPU suite against the in-tree library in a subprocess."""
import ctypes as C
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: E402,F401
from tts_cpp_amd import gguf, synth  # noqa: E402

L = C.CDLL(sys.argv[1])
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
flips = int(sys.argv[3]) if len(sys.argv) > 3 else 600
L.tts_c_runner_from_file.restype = C.c_void_p
tmp = tempfile.mkdtemp(prefix="fuzz_gguf_")
base, case = os.path.join(tmp, "base.gguf"), os.path.join(tmp, "case.gguf")
synth.build(synth.tiny(weight_type=gguf.Q8_0)).write_gguf(base)
data = open(base, "rb").read()
meta_end = gguf.Reader(base).data_offset
rng = random.Random(seed)


def probe(blob):
    open(case, "wb").write(blob)
    nt, nkv, off = C.c_uint64(), C.c_uint64(), C.c_uint64()
    arch = C.create_string_buffer(64)
    if L.tts_c_gguf_summary(case.encode(), C.byref(nt), C.byref(nkv), C.byref(off), arch, 64) == 0:
        for idx in (0, 1, max(0, nt.value - 1), nt.value, nt.value + 5):
            name = C.create_string_buffer(256)
            tt, ne, cs = C.c_int(), (C.c_int64 * 4)(), C.c_uint64()
            L.tts_c_gguf_tensor(case.encode(), idx, name, 256, C.byref(tt), ne, C.byref(cs))
    cfg = (C.c_char * 64)()
    L.tts_c_default_config(cfg)
    L.tts_c_runner_from_file(case.encode(), 1, cfg, 0)   # metadata + tokenizer are read before the device is asked for (no GPU: stops there)


n = 0
for cut in sorted(set([0, 1, 3, 4, 8, 16, 24] + [rng.randrange(0, meta_end) for _ in range(flips // 4)] + [meta_end - 1, meta_end, meta_end + 1, len(data) - 1])):
    probe(data[:cut])
    n += 1
for _ in range(flips):
    b = bytearray(data)
    for _ in range(rng.choice([1, 1, 2, 4])):
        b[rng.randrange(0, meta_end)] = rng.choice([0, 1, 0x7F, 0x80, 0xFF, rng.randrange(256)])
    probe(bytes(b))
    n += 1
for off in (8, 16):
    for v in (0xFFFFFFFFFFFFFFFF, 1 << 40, 1 << 31):
        b = bytearray(data)
        b[off:off + 8] = v.to_bytes(8, "little")
        probe(bytes(b))
        n += 1
for f in (base, case):
    os.unlink(f)
os.rmdir(tmp)
print(f"{n} malformed files, no crash")
