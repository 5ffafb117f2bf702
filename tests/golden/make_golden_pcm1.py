#!/usr/bin/env python3
"""Generates the PCM-1 golden fixtures (pcm1_<case>.npz) by running the REAL reference (oracle/_ref/libsdvref.so:
PCM1DataStitcher::doFrameReassemble on its own thread, PCM1Line::calcCRC for the CRC known answers) on the seeded scenarios
of tests/pcm1_api.py.  Build container only (needs /root/reference).

Each fixture: sha256 of the input record stream (regenerated from the seeds by the test), the settings, and the expected
PCMSamplePair stream + FrameAsmPCM1 rows.  pcm1_crc.npz: random word sextets with the reference's CRC."""
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
import pcm1_api as p1  # noqa: E402

if __name__ == "__main__":
    ref = libs.load_ref()
    ref.ref_pcm1_crc.restype = C.c_uint16
    rng = np.random.default_rng(77)
    words = rng.integers(0, 1 << 13, size=(256, 6)).astype(np.uint16)
    words[0] = (0x1A35, 0x1248, 0x0DD9, 0x13FB, 0x1C0E, 0x09CB)      # the reference's own test line (pcmtester.cpp:14-21), CRC 0x9EB9
    words[1] = 1 << 12                                                # a silent line, CRC_SILENT (pcm1line.h)
    crc = np.array([ref.ref_pcm1_crc(np.ascontiguousarray(w).ctypes.data_as(C.POINTER(C.c_uint16))) for w in words], dtype=np.uint16)
    assert crc[0] == 0x9EB9 and crc[1] == 0xECBF
    np.savez_compressed(os.path.join(HERE, "pcm1_crc.npz"), words=words, crc=crc)
    only = sys.argv[1:]                                               # names to (re)generate; all when none is given
    for name in p1.GOLDEN:
        if only and name not in only:
            continue
        recs, st = p1.make_input(name)
        pairs, frames = p1.run_cpu(ref, "ref_", recs, st)
        path = os.path.join(HERE, "pcm1_" + name + ".npz")
        np.savez_compressed(path, input_sha256=hashlib.sha256(recs.tobytes()).hexdigest(), settings=np.frombuffer(bytes(st), dtype=np.uint8),
                            pairs=pairs.view(np.uint8).reshape(len(pairs), 12), frames=frames.view(np.uint8).reshape(len(frames), 52))
        print(f"{name}: {len(recs)} records -> {len(pairs)} sample pairs, {len(frames)} frames, {os.path.getsize(path)} bytes")
