#!/usr/bin/env python3
"""Generates the PCM-16x0 back half golden fixtures by running the REAL reference (oracle/_ref/libsdvref.so:
PCM16X0DataStitcher::doFrameReassemble on its own thread, PCM16X0Deinterleaver::processBlock) on the seeded scenarios of
tests/pcm16_api.py.  Build container only (needs /root/reference).

pcm16_<case>.npz: sha256 of the input sub-line stream (regenerated from the seeds by the test), the settings, and the expected
PCMSamplePair stream + FrameAsmPCM16x0 rows.  pcm16_blocks.npz: data blocks of damaged SI and EI queues under every
combination of the deinterleaver's switches."""
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

ONLY = sys.argv[1:]            # names of the scenarios to (re)generate; none: everything

if __name__ == "__main__":
    ref = libs.load_ref()
    blocks = {}
    for key, (recs, kw) in p16.block_inputs().items():
        blocks[key] = p16.run_blocks(ref, "ref_", recs, **kw).view(np.uint8)
    if not ONLY:
        np.savez_compressed(os.path.join(HERE, "pcm16_blocks.npz"), **blocks)
    print("blocks:", len(blocks), "runs,", os.path.getsize(os.path.join(HERE, "pcm16_blocks.npz")), "bytes")
    for name in p16.GOLDEN:
        if ONLY and name not in ONLY:
            continue
        recs, st = p16.make_input(name)
        pairs, frames = p16.run_cpu(ref, "ref_", recs, st)
        path = os.path.join(HERE, "pcm16_" + name + ".npz")
        np.savez_compressed(path, input_sha256=hashlib.sha256(recs.tobytes()).hexdigest(), settings=np.frombuffer(bytes(st), dtype=np.uint8),
                            pairs=pairs.view(np.uint8).reshape(len(pairs), 12), frames=frames.view(np.uint8).reshape(len(frames), 56))
        print(f"{name}: {len(recs)} records -> {len(pairs)} sample pairs, {len(frames)} frames, {os.path.getsize(path)} bytes")

# the whole PCM-16x0 path - video through the real VideoToDigital worker (TYPE_PCM16X0), its sub-lines through the real
# PCM16X0DataStitcher - for one file from NEW_FILE to END_FILE in either interleave format: e2e_pcm16x0_<si|ei>.npz holds what the
# luma (regenerated from the seed by the test) has to decode to
def make_e2e_luma(ei):
    from sdvpcmdecoder_amd import synth
    luma, audio, _ = synth.pcm16x0_tape_frames(5, seed=61 + int(ei), ei=ei, noise_sigma=4.0)
    return luma, audio


if __name__ == "__main__" and not ONLY:
    import pcm16_frames_api as fa
    for ei in (False, True):
        luma, audio = make_e2e_luma(ei)
        recs, stats = fa.run_cpu(ref, "ref_", luma, 2, dict(new_file=True, end_file=True))
        st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
        pairs, frames = p16.run_cpu(ref, "ref_", recs, st)
        path = os.path.join(HERE, "e2e_pcm16x0_%s.npz" % ("ei" if ei else "si"))
        np.savez_compressed(path, luma_sha256=hashlib.sha256(luma.tobytes()).hexdigest(), recs_sha256=hashlib.sha256(recs.tobytes()).hexdigest(),
                            pairs=pairs.view(np.uint8).reshape(len(pairs), 12), frames=frames.view(np.uint8).reshape(len(frames), 56))
        print(f"e2e {'EI' if ei else 'SI'}: {len(recs)} records -> {len(pairs)} pairs, {len(frames)} frames, audio recovered:",
              bool((pairs['audio_word'][pairs['service_type'] == 0] == audio).all()), os.path.getsize(path), "bytes")
