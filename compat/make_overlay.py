#!/usr/bin/env python3
"""make_overlay.py REFERENCE_ROOT DEST — lay this engine's headers over a TTS.cpp checkout so that its applications build unchanged.

DEST becomes a tree with the reference's layout:
  examples/cli/*, examples/perf_battery/*, examples/server/*, src/args.cpp, include/args.h, include/audio_file.h   -> symlinks to REFERENCE_ROOT
  include/common.h, include/ggml.h, include/util.h, src/models/loaders.h                          -> this engine's compat/ headers
  examples/server/index.html.hpp   -> generated here from REFERENCE_ROOT/examples/server/public/index.html (what the reference's CMake does
                                      with cmake/xxd.cmake: the page as `unsigned char index_html[]`, `unsigned int index_html_len`)
Nothing of the reference is copied; the applications' sources are compiled from where they lie (through the links), which is what
keeps their relative `#include "../../src/models/loaders.h"` resolving to the overlay.  Build:
  g++ -std=c++20 -I DEST/include DEST/examples/cli/{cli,playback,vad,write_file}.cpp DEST/src/args.cpp -L host -ltts -L .. -ltts_hip
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LINKS = ["examples/cli", "examples/perf_battery", "examples/server", "src/args.cpp", "include/args.h", "include/audio_file.h"]
OURS = ["include/common.h", "include/ggml.h", "include/util.h", "src/models/loaders.h"]


def make_overlay(ref_root, dest):
    for rel in LINKS:
        src = os.path.join(ref_root, rel)
        if not os.path.exists(src):
            raise FileNotFoundError(src)
        if os.path.isdir(src):
            os.makedirs(os.path.join(dest, rel), exist_ok=True)
            for f in sorted(os.listdir(src)):
                if f.endswith((".cpp", ".h", ".hpp")):
                    link = os.path.join(dest, rel, f)
                    if not os.path.lexists(link):
                        os.symlink(os.path.join(src, f), link)
        else:
            os.makedirs(os.path.dirname(os.path.join(dest, rel)), exist_ok=True)
            if not os.path.lexists(os.path.join(dest, rel)):
                os.symlink(src, os.path.join(dest, rel))
    page = os.path.join(ref_root, "examples/server/public/index.html")
    if os.path.exists(page):
        data = open(page, "rb").read()
        with open(os.path.join(dest, "examples/server/index.html.hpp"), "w") as f:
            f.write("unsigned char index_html[] = {" + ",".join(f"0x{b:02x}" for b in data) + "};\nunsigned int index_html_len = %d;\n" % len(data))
    for rel in OURS:
        os.makedirs(os.path.dirname(os.path.join(dest, rel)), exist_ok=True)
        with open(os.path.join(dest, rel), "w") as f:
            f.write(f'#pragma once\n#include "{os.path.join(HERE, rel)}"\n')
    return dest


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    print(make_overlay(os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])))
