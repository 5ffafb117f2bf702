#!/bin/bash
# Sanitizer pass over the CPU builds (GPU ASan is not available on the pool): the oracle under ASan + UBSan, the kernel source on the
# SIMT emulator under UBSan (ASan and the emulator's ucontext fibers do not get along).  Run from the repository root.
set -e
mkdir -p build/san
( cd oracle && gcc -O1 -g -std=c11 -fPIC -fsanitize=undefined,address -fno-omit-frame-pointer -shared -o ../build/san/liborc.so binarizer.c api.c v2d.c deint.c stitcher.c pcm1.c bin_pcm1.c v2d_p1.c bin_pcm16.c v2d_p16.c pcm16.c audio.c )
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=undefined -fno-sanitize=alignment -Wno-unused-function -Wno-attributes -o build/san/libsdvpcm_emu.so tests/emu/emu_engine.cpp
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 python3 tools/san_orc.py
LD_PRELOAD="$(gcc -print-file-name=libubsan.so)" UBSAN_OPTIONS=print_stacktrace=1 python3 tools/san_emu.py
