// compat/include/ggml.h — the two ggml entry points the reference's applications call directly: ggml_time_init / ggml_time_us
// (examples/cli/cli.cpp:13-18, examples/server/server.cpp).  ggml itself is not part of this engine; these are monotonic-clock
// microseconds, which is what the callers use them for (a "total time" line).
#pragma once
#include <chrono>
#include <cstdint>

inline void ggml_time_init(void) {}
inline int64_t ggml_time_us(void) {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
inline int64_t ggml_time_ms(void) { return ggml_time_us() / 1000; }
