"""TEST INFRASTRUCTURE: the reference's noise streams restated.  random_uniform_gen (/root/reference/src/util.cpp:65-71) draws from a
function-static std::default_random_engine (libstdc++: minstd_rand0, x <- 16807 x mod 2^31 - 1, first state 1) through
std::uniform_real_distribution<float>(0, 1): one engine call per draw, (x - 1) / 2147483646 evaluated in float, a result of 1.0
replaced by the float below it.  tests/test_host_cpu.py compares this with a C++ program built by the local toolchain."""
import numpy as np


def minstd0_uniform(n, state=1):
    """-> (n float32 draws, engine state afterwards)"""
    out = np.empty(n, dtype=np.float32)
    x = state
    below_one = np.nextafter(np.float32(1.0), np.float32(0.0))
    for i in range(n):
        x = (x * 16807) % 2147483647
        v = np.float32(x - 1) / np.float32(2147483646.0)
        out[i] = v if v < 1.0 else below_one
    return out, x


def minstd0_normal(n, state=1, saved=None):
    """random_normal_gen (util.cpp:73-79): the same engine through std::normal_distribution<float>(0, 1) — libstdc++'s Marsaglia polar
    method in float (x = 2 u - 1 goes through double because of the `1.0` literal), second value of each pair kept for the next call;
    logf is the C library's (called through ctypes: its last-bit rounding is part of the stream).  -> (draws, engine state, saved value)"""
    import ctypes
    import math
    libm = ctypes.CDLL("libm.so.6")
    libm.logf.restype = ctypes.c_float
    libm.logf.argtypes = [ctypes.c_float]
    f32 = np.float32
    below_one = np.nextafter(f32(1.0), f32(0.0))

    def canon(x):
        v = f32(x - 1) / f32(2147483646.0)
        return v if v < 1.0 else below_one

    out = np.empty(n, dtype=np.float32)
    x, i = state, 0
    while i < n:
        if saved is not None:
            out[i], saved = saved, None
            i += 1
            continue
        while True:
            x = (x * 16807) % 2147483647
            a = canon(x)
            x = (x * 16807) % 2147483647
            b = canon(x)
            xx = f32(np.float64(f32(2.0) * a) - 1.0)
            yy = f32(np.float64(f32(2.0) * b) - 1.0)
            r2 = f32(f32(xx * xx) + f32(yy * yy))
            if not (r2 > 1.0 or r2 == 0.0):
                break
        mult = f32(math.sqrt(float(f32(f32(f32(-2.0) * f32(libm.logf(float(r2)))) / r2))))
        saved = f32(xx * mult)
        out[i] = f32(yy * mult)
        i += 1
    return out, x, saved
