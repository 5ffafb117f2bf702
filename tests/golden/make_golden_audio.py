#!/usr/bin/env python3
"""Generates the AudioProcessor golden fixtures (audio_<case>.npz) by running the REAL reference (oracle/_ref/libsdvref.so: the
AudioProcessor's own processAudio loop on its own thread, fed in the bursts of the case; SamplesToWAV writing its files into a
temporary directory) on the seeded scenarios of tests/audio_api.py.  Build container only (needs /root/reference).

Each fixture: sha256 of the input PCMSamplePair stream (regenerated from the seeds by the test), the mode, the burst ends and the
stop flag, and what the reference put out: the pairs, PCMSample::index of each, the positions of its newSource signals, the sum of
its guiAddMask reports, and the bytes of the WAV files it wrote."""
import hashlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
import audio_api as A  # noqa: E402


def read_wavs(d):
    files = {}
    for f in sorted(os.listdir(d)):
        assert f.startswith("src") and f.endswith("_v0.99.7.wav"), f
        files[int(f[3:].split("_")[0])] = np.frombuffer(open(os.path.join(d, f), "rb").read(), dtype=np.uint8)
    return files


ONLY = sys.argv[1:]            # names of the scenarios to (re)generate; none: everything

if __name__ == "__main__":
    ref = libs.load_ref()
    for name in A.GOLDEN:
        if ONLY and name not in ONLY:
            continue
        pairs, mode, ends, stop = A.make_input(name)
        with tempfile.TemporaryDirectory() as d:
            out, idx, pur, masked, _ = A.run_cpu(ref, "ref_", pairs, mode, ends, stop, wav_dir=d)
            wavs = read_wavs(d)
        path = os.path.join(HERE, "audio_" + name + ".npz")
        extra = {"wav%d" % k: v for k, v in wavs.items()}
        np.savez_compressed(path, input_sha256=hashlib.sha256(pairs.tobytes()).hexdigest(), mode=mode, ends=ends, stop=stop,
                            pairs=out.view(np.uint8).reshape(len(out), 12), index=idx, purges=pur, masked=masked, **extra)
        print(f"{name}: {len(pairs)} pairs in -> {len(out)} out, {len(pur)} purges, {masked} masked, wav files {sorted(wavs)}, {os.path.getsize(path)} bytes")

    if ONLY:
        sys.exit(0)
    # end to end: the PCMSamplePair stream the real VideoToDigital + STC007DataStitcher made of the synthetic NTSC file (e2e_ntsc_file.npz,
    # tests/golden/make_golden_stitch.py) through the real AudioProcessor and into the real SamplesToWAV
    from stitch_api import PAIR_DTYPE
    z = np.load(os.path.join(HERE, "e2e_ntsc_file.npz"))
    pairs = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    ends = np.array([len(pairs)], dtype=np.uint64)
    with tempfile.TemporaryDirectory() as d:
        out, idx, pur, masked, _ = A.run_cpu(ref, "ref_", pairs, A.DROP_INTER_LIN_WORD, ends, 1, wav_dir=d)
        wavs = read_wavs(d)
    path = os.path.join(HERE, "e2e_ntsc_file_audio.npz")
    np.savez_compressed(path, input_sha256=hashlib.sha256(pairs.tobytes()).hexdigest(), pairs=out.view(np.uint8).reshape(len(out), 12), purges=pur, masked=masked,
                        **{"wav%d" % k: v for k, v in wavs.items()})
    print(f"e2e_ntsc_file: {len(pairs)} pairs in -> {len(out)} out, {masked} masked, wav files {sorted(wavs)}, {os.path.getsize(path)} bytes")
