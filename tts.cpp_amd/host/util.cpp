// util.cpp — the architecture table (src/models/loaders.cpp / include/common.h:19-43) and TTS_ABORT (src/util.cpp:14-22): print file:line + message and abort(); the C wrappers flip
// g_tts_throw_on_abort so that a language binding gets an error string instead of losing the process.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "common.h"

const std::map<std::string, tts_arch> SUPPORTED_ARCHITECTURES = {
    {"parler-tts", PARLER_TTS_ARCH}, {"kokoro", KOKORO_ARCH}, {"dia", DIA_ARCH}, {"orpheus", ORPHEUS_ARCH}};
const std::map<tts_arch, std::string> ARCHITECTURE_NAMES = {
    {PARLER_TTS_ARCH, "parler-tts"}, {KOKORO_ARCH, "kokoro"}, {DIA_ARCH, "dia"}, {ORPHEUS_ARCH, "orpheus"}};

std::atomic<bool> g_tts_throw_on_abort{false};  // set by the C wrapper so that language bindings get an error instead of abort()

void tts_abort(const char * file, int line, const char * fmt, ...) {
    char    msg[2048];
    va_list ap;
    va_start(ap, fmt);
    const int n = snprintf(msg, sizeof(msg), "%s:%d: ", file, line);
    vsnprintf(msg + n, sizeof(msg) - (size_t) n, fmt, ap);
    va_end(ap);
    if (g_tts_throw_on_abort) throw std::runtime_error(msg);
    fflush(stdout);
    fputs(msg, stderr);
    abort();  // util.cpp:14-22
}
