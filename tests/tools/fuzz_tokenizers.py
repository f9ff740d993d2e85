#!/usr/bin/env python3
"""Hostile text against the host tokenizers (host/tokenizer.cpp, dia_runner.cpp, kokoro_runner.cpp through include/tts_c.h): arbitrary bytes, invalid and
truncated UTF-8, long inputs, output buffers too short for the result.  Usage: fuzz_tokenizers.py <libtts.so> [seed] [cases]; clean under ASan + UBSan
builds of the library (3 seeds x 400 texts, end of round 4); the CPU suite runs 120 texts against the in-tree library in a subprocess."""
import ctypes as C
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tts_cpp_amd  # noqa: E402,F401
from tts_cpp_amd import gguf, synth  # noqa: E402

L = C.CDLL(sys.argv[1])
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 400
tmp = tempfile.mkdtemp(prefix="fuzz_tok_")
path = os.path.join(tmp, "tok.gguf")
synth.build(synth.tiny(weight_type=gguf.Q8_0)).write_gguf(path)


def text(n):
    kind = rng.randrange(5)
    if kind == 0:
        b = bytes(rng.randrange(1, 256) for _ in range(n))                       # anything but NUL
    elif kind == 1:
        b = "".join(rng.choice("abc xyz.,!?[S1][S2] ") for _ in range(n)).encode()
    elif kind == 2:
        b = "".join(chr(rng.choice([0x41, 0xE9, 0x4F60, 0x1F600, 0x2581, 0x20])) for _ in range(n)).encode("utf-8")
    elif kind == 3:
        b = (bytes([0xF0, 0x9F]) * n)[:n]                                         # truncated four-byte sequences
    else:
        b = bytes([rng.choice([0xC0, 0xE0, 0xF8, 0xFF, 0x80, 0x20, 0x61])]) * n
    return b.replace(b"\0", b"\1")


out = np.zeros(1 << 16, dtype=np.uint32)
op = out.ctypes.data_as(C.POINTER(C.c_uint32))
vocab = ["<unk>", "a", "b", " ", "é", "你", "ab", ".", ","]
va = (C.c_char_p * len(vocab))(*[v.encode() for v in vocab])
for _ in range(cases):
    t = text(rng.choice([0, 1, 2, 7, 33, 200, 1500, 9000]))
    L.tts_c_tokenize(path.encode(), t, op, out.size)
    L.tts_c_tokenize(path.encode(), t, op, rng.choice([0, 1, 5]))
    L.tts_c_dia_tokenize(t, rng.choice([1, 8, 64, 1024]), op)
    k = L.tts_c_single_pass_tokenize(va, len(vocab), t, None, 0)
    L.tts_c_single_pass_tokenize(va, len(vocab), t, op, max(k, 0))
    for mc in (3, 8, 64):
        k = L.tts_c_kokoro_chunks(va, len(vocab), t, mc, 3, None, 0)
        if k > 0:
            L.tts_c_kokoro_chunks(va, len(vocab), t, mc, 3, op, min(k, out.size))
os.unlink(path)
os.rmdir(tmp)
print(f"{cases} hostile texts, no crash")
