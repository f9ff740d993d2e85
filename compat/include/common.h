// compat/include/common.h — stands where the reference's include/common.h stands (/root/reference/include/common.h).
//
// The reference's applications (examples/cli/cli.cpp:3-9, examples/perf_battery/perf_battery.cpp:7-9, examples/server/server.cpp)
// are written against include/common.h *including* its global `using namespace std;` (common.h:7): cli.cpp:79 and
// perf_battery.cpp:102 spell `unique_ptr<tts_generation_runner>` unqualified.  The engine's own header keeps std:: explicit; this
// overlay header adds the using-directive so that those translation units compile unchanged.
#pragma once
#include "../../tts.cpp_amd/host/common.h"

using namespace std;
