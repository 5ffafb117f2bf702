#!/usr/bin/env python3
"""Generates the fixtures of the PCM-16x0 data blocks window (pcm16vis_<case>.npz) with the REAL reference (oracle/_ref/libsdvref.so): the real
PCM16X0DataStitcher's newBlockProcessed blocks (as sdv_pcm16x0_block_rec, read through the object's public interface) and the canvases the real
RenderPCM draws when it is fed by the real stitcher (renderNewBlock per block, prepareNewFrame per frame).  Build container only (needs /root/reference)."""
import ctypes as C
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
import test_pcm16_vis as tv  # noqa: E402

if __name__ == "__main__":
    ref = libs.load_ref()
    only = sys.argv[1:]
    for name in p16.VIS_GOLDEN:
        if only and name not in only:
            continue
        recs, st = p16.make_input(name)
        pairs, frames, blocks = p16.run_cpu_vis(ref, "ref_", recs, st)
        per = np.ascontiguousarray((frames["blocks_total"][frames["service_type"] == 0] // 3).astype(np.uint32))
        canv = tv._ref_canvases(recs, st, len(per))
        mask = ra.written_blocks(ra.PCM16X0_BLOCKS, per)
        path = os.path.join(HERE, "pcm16vis_" + name + ".npz")
        np.savez_compressed(path, input_sha256=hashlib.sha256(recs.tobytes()).hexdigest(), blocks_sha256=hashlib.sha256(np.ascontiguousarray(blocks).tobytes()).hexdigest(),
                            canvases_sha256=ra.digest(canv, mask), last_canvas=np.where(mask[-1], canv[-1], 0).astype(np.uint32))
        print(f"{name}: {len(blocks)} blocks -> {len(per)} canvases, {os.path.getsize(path)} bytes")
