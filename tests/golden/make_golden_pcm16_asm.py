#!/usr/bin/env python3
"""Generates the fixtures of the PCM-16x0 re-assembled window (pcm16asm_<case>.npz) with the REAL reference (oracle/_ref/libsdvref.so): the sub-lines the
real PCM16X0DataStitcher hands to newLineProcessed (as sdv_pcm16x0_bin_rec, read through the object's public interface, an END_FRAME record where
MainWindow emits newFrameAssembled) and the canvases the real RenderPCM draws from them (renderNewLine(PCM16X0SubLine) per sub-line, prepareNewFrame per
frame).  Build container only (needs /root/reference)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
import pcm16_api as p16  # noqa: E402
import render_api as ra  # noqa: E402
import test_pcm16_asm as ta  # noqa: E402

if __name__ == "__main__":
    ref = libs.load_ref()
    only = sys.argv[1:]
    for name in ta.ASM_GOLDEN:
        if only and name not in only:
            continue
        recs, st = p16.make_input(name)
        pairs, frames, blocks, lines = p16.run_cpu_feeds(ref, "ref_", recs, st)
        lines = np.ascontiguousarray(lines)
        canv = ra.run_ref(ra.PCM16X0, lines)
        mask = ra.written(ra.PCM16X0, lines)
        path = os.path.join(HERE, "pcm16asm_" + name + ".npz")
        np.savez_compressed(path, input_sha256=hashlib.sha256(recs.tobytes()).hexdigest(), lines_sha256=hashlib.sha256(lines.tobytes()).hexdigest(),
                            canvases_sha256=ra.digest(canv, mask), last_canvas=np.where(mask[-1], canv[-1], 0).astype(np.uint32))
        print(f"{name}: {len(lines)} sub-line records -> {len(canv)} canvases, {os.path.getsize(path)} bytes")
