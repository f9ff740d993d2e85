// compat/include/util.h — stands where the reference's src/util.h stands for its applications (examples/server/server.cpp:3 includes "util.h" for
// has_prefix / split at :809-821 and TTS_ABORT).  The reference's header is mostly ggml graph helpers; an application needs the string helpers only.
// Same contracts as /root/reference/src/util.cpp:103-109, 219-281: a separator character ends a part, empty parts are dropped, separators are
// reported as parts of their own only on request.
#pragma once
#include <string>
#include <vector>

#include "../../tts.cpp_amd/host/common.h"   // tts_abort / TTS_ABORT

#ifndef TTS_ASSERT
#define TTS_ASSERT(x) if (!(x)) TTS_ABORT("TTS_ASSERT(%s) failed", #x)
#endif

inline bool has_prefix(std::string value, std::string prefix) { return value.rfind(prefix, 0) == 0; }
inline bool has_suffix(std::string value, std::string suffix) {
    return value.size() >= suffix.size() && value.compare(value.size() - suffix.size(), suffix.size(), suffix) == 0;
}
inline std::vector<std::string> split(std::string target, std::string split_on, bool include_split_characters = false) {
    std::vector<std::string> parts;
    std::string cur;
    for (const char ch : target) {
        if (split_on.find(ch) == std::string::npos) { cur.push_back(ch); continue; }
        if (!cur.empty()) { parts.push_back(cur); cur.clear(); }
        if (include_split_characters) parts.push_back(std::string(1, ch));
    }
    if (!cur.empty()) parts.push_back(cur);
    return parts;
}
inline std::vector<std::string> split(std::string target, const char split_on, bool include_split_characters = false) {
    return split(target, std::string(1, split_on), include_split_characters);
}
inline std::string strip(std::string target, std::string vals = " ") {
    const size_t a = target.find_first_not_of(vals);
    if (a == std::string::npos) return "";
    return target.substr(a, target.find_last_not_of(vals) - a + 1);
}
