// compat/include/ggml.h — what the reference's applications use of ggml directly: ggml_time_init / ggml_time_us and GGML_ASSERT (server.cpp:7 hands it to json.hpp)
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

#include <cstdio>
#include <cstdlib>
#ifndef GGML_ASSERT
#define GGML_ASSERT(x) do { if (!(x)) { fprintf(stderr, "%s:%d: GGML_ASSERT(%s) failed\n", __FILE__, __LINE__, #x); abort(); } } while (0)
#endif
