#!/usr/bin/env python3
"""Generates the fixtures of the PCM-1 visualiser feeds (pcm1vis_<case>.npz) by running the REAL reference (oracle/_ref/libsdvref.so:
PCM1DataStitcher on its own thread with its newBlockProcessed / newLineProcessed signals connected) on seeded scenarios of tests/pcm1_api.py.
Build container only (needs /root/reference).

Each fixture: sha256 of the input record stream, the blocks (sdv_pcm1_block_rec, with what the reference leaves undefined taken out:
pcm1_api.comparable_blocks) and the sub-lines in the order the stitcher hands them over (sdv_pcm1_asm_line_rec)."""
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
    only = sys.argv[1:]
    for name in p1.VIS_GOLDEN:
        if only and name not in only:
            continue
        recs, st = p1.make_input(name)
        pairs, frames, blocks, lines = p1.run_cpu_vis(ref, "ref_", recs, st)
        np.savez_compressed(os.path.join(HERE, "pcm1vis_" + name + ".npz"), input_sha256=hashlib.sha256(recs.tobytes()).hexdigest(),
                            blocks=p1.comparable_blocks(blocks), lines=lines)
        print(name, len(blocks), "blocks", len(lines), "sub-lines")
