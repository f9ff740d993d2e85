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
