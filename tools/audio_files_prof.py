"""The audio stage on a burst of many files: F files of n/F frames each (every window holds an invalid word), one call.  The stretches between
tags are planned by a wave each, so the in-order part shrinks with the number of files.   tools/audio_files_prof.py [frames] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import audio_api as A
from sdvpcmdecoder_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = Engine(0)
eng.set_audio_masking(A.DROP_INTER_LIN_WORD)
data = A.audio(n * 1470, 3, tone=False, p_bad=0.01)
for files in (1, 4, 16, 64, 256):
    per = len(data) // files
    parts = []
    for f in range(files):
        parts += ["N", data[f * per:(f + 1) * per], "E"]
    tape = A.tape(parts)
    d = torch.from_numpy(tape.view(np.uint8).reshape(len(tape), 12)).cuda()
    out = torch.empty((len(tape) + 1024, 12), dtype=torch.uint8, device='cuda')
    pur = torch.empty((2 * files + 4, 16), dtype=torch.uint8, device='cuda')
    for it in range(reps):
        eng.reset_audio()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o, p, m = eng.audio_process(d, stop=True, out_pairs=out, out_purges=pur)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{files} files: {dt*1e3:.3f} ms per {n} frames, {n/dt/1e6:.2f} M frames/s, masked {m}, purges {p.shape[0]}", flush=True)
