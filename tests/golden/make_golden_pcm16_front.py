#!/usr/bin/env python3
"""Generates the PCM-16x0 front-half golden fixtures (pcm16front_<case>.npz) by running the REAL reference (oracle/_ref/libsdvref.so:
Binarizer::processLine with a PCM16X0SubLine output, one pass per line part, ref_bin16_process) on the seeded scenarios of
tests/pcm16_front_api.py.  Build container only (needs /root/reference).

Each fixture: sha256 of the input luma rows (regenerated from the seeds by the test) and the expected sub-line records (36 bytes
each), return codes and VideoLine::scan_done flags."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
import pcm16_front_api as pf  # noqa: E402

if __name__ == "__main__":
    ref = libs.load_ref()
    for name in pf.GOLDEN:
        luma, run = pf.make_case(name)
        recs, rets, scans = pf.run_lines(ref, "ref_bin16_", luma, **run)
        path = os.path.join(HERE, "pcm16front_" + name + ".npz")
        np.savez_compressed(path, input_sha256=hashlib.sha256(luma.tobytes()).hexdigest(), recs=recs.view(np.uint8).reshape(len(recs), 36),
                            rets=rets, scans=scans)
        print(f"{name}: {len(recs)} sub-lines, {int((recs['flags'] & 64 != 0).sum())} with a valid CRC, {os.path.getsize(path)} bytes")
