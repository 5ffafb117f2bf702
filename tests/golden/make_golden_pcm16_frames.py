#!/usr/bin/env python3
"""Generates the PCM-16x0 frame-driver golden fixtures (pcm16frames_<case>.npz) by running the REAL reference (oracle/_ref/libsdvref.so:
VideoToDigital with setPCMType(TYPE_PCM16X0) on its worker thread, ref_v2d16_run) on the seeded scenarios of tests/pcm16_frames_api.py.
Build container only (needs /root/reference).

Each fixture: sha256 of the input frames (regenerated from the seeds by the test), the expected PCM16X0SubLine records (36 bytes each)
and the FrameBinDescriptor rows (32 bytes each)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
import pcm16_frames_api as pf  # noqa: E402

if __name__ == "__main__":
    ref = libs.load_ref()
    for name in pf.GOLDEN:
        luma, mode, st = pf.make_input(name)
        recs, stats = pf.run_cpu(ref, "ref_", luma, mode, st)
        path = os.path.join(HERE, "pcm16frames_" + name + ".npz")
        np.savez_compressed(path, input_sha256=hashlib.sha256(luma.tobytes()).hexdigest(), recs=recs.view(np.uint8).reshape(len(recs), 36),
                            stats=stats.view(np.uint8).reshape(len(stats), 32))
        print(f"{name}: {len(recs)} records, {int((recs['flags'] & 64 != 0).sum())} with a valid CRC, {os.path.getsize(path)} bytes")
