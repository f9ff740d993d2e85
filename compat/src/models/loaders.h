// compat/src/models/loaders.h — stands where the reference's src/models/loaders.h stands (the applications include it by the
// relative path "../../src/models/loaders.h": cli.cpp:3, perf_battery.cpp:7).  runner_from_file (loaders.h:19-20) and
// tts_model_loader (loaders.h:7-17) are declared by the engine's common.h.
#pragma once
#include "../../include/common.h"
