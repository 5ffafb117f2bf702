#!/usr/bin/env python3
"""Generates the stitch-stage golden fixtures (stitch_<case>.npz) by running the REAL reference
(oracle/_ref/libsdvref.so: VideoToDigital for the line records, then STC007DataStitcher::doFrameReassemble on its own
thread) on the seeded scenarios of tests/stitch_cases.py.  Build container only (needs /root/reference).

Each fixture: sha256 of the input record stream (the stream itself is regenerated from the seeds by the test),
the stitcher settings, and the expected PCMSamplePair stream + FrameAsmSTC007 rows."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import libs  # noqa: E402
import stitch_api as sa  # noqa: E402
import stitch_cases as sc  # noqa: E402
from make_golden import run_ref  # noqa: E402

E2E = dict(n_frames=4, seed=301, noise_sigma=6.0, blur=1)      # a whole (short) file through both reference workers


def make_e2e_luma():
    from sdvpcmdecoder_amd import synth
    luma, _, _ = synth.stc007_frames(**E2E)
    luma = luma.copy()
    luma[:, 101::53] = 16                          # dropouts: lost lines for P/Q and CWD to repair
    return np.ascontiguousarray(luma)


if __name__ == "__main__":
    # end to end: video -> VideoToDigital (NEW_FILE ... filler frame, END_FILE) -> STC007DataStitcher
    luma = make_e2e_luma()
    recs, stats = run_ref(luma, 2, first=1, new_file=1, end_file=1)
    st = sa.default_settings()
    pairs, frames = sa.run_cpu(libs.load_ref(), "ref_", recs, st)
    path = os.path.join(HERE, "e2e_ntsc_file.npz")
    np.savez_compressed(path, recs_sha256=sc.digest(recs), stats=stats, settings=np.frombuffer(bytes(st), dtype=np.uint8),
                        pairs=pairs.view(np.uint8).reshape(len(pairs), 12), frames=frames.view(np.uint8).reshape(len(frames), 64))
    print(f"e2e_ntsc_file: {len(recs)} records -> {len(pairs)} sample pairs, {len(frames)} frames, {os.path.getsize(path)} bytes")
    for name in sc.GOLDEN:
        recs, st = sc.make_input(name, lambda luma: run_ref(luma, 2))
        pairs, frames = sa.run_cpu(libs.load_ref(), "ref_", recs, st)
        path = os.path.join(HERE, "stitch_" + name + ".npz")
        np.savez_compressed(path, input_sha256=sc.digest(recs), settings=np.frombuffer(bytes(st), dtype=np.uint8),
                            pairs=pairs.view(np.uint8).reshape(len(pairs), 12), frames=frames.view(np.uint8).reshape(len(frames), 64))
        print(f"{name}: {len(recs)} records -> {len(pairs)} sample pairs, {len(frames)} frames, {os.path.getsize(path)} bytes")
