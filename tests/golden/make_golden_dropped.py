#!/usr/bin/env python3
"""Generates dropped_<case>.npz: the records and frame descriptors the REAL VideoToDigital worker (oracle/_ref) produces for the
scenarios of tests/test_dropped_frames.py - frames whose lines arrive as the empty VideoLines VideoInFFMPEG::insertDummyFrame(false, true)
makes for a dropped frame (vin_ffmpeg.cpp:367-522), for the three PCM types.  Build container only (needs /root/reference)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
import test_dropped_frames as T  # noqa: E402

if __name__ == "__main__":
    ref = libs.load_ref()
    for name in sorted(T.CASES):
        recs, stats = T.run_cpu(ref, "ref_", name)
        fmt, luma, mode, mask, fl = T.make_case(name)
        path = os.path.join(HERE, "dropped_" + name + ".npz")
        np.savez_compressed(path, recs=np.ascontiguousarray(recs).view(np.uint8), stats=np.ascontiguousarray(stats).view(np.uint8).reshape(-1),
                            mask=mask, input_sha256=hashlib.sha256(luma.tobytes()).hexdigest())
        ok = int(((recs["flags"] & 64) != 0).sum())
        print(f"{name}: {len(recs)} records ({ok} with a valid CRC), {len(stats)} frame descriptors, {os.path.getsize(path)} bytes")
