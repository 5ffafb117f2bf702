"""How the lean/full kernel schedule behaves when lines need the general path: noisy tapes, dropouts."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = 2000
for sigma, drop in ((4.0, 0), (4.0, 97), (4.0, 997), (20.0, 0), (35.0, 0)):
    luma, _ = synth.stc007_frames_torch(n, seed=3, device='cuda', noise_sigma=sigma, cyclic=True)
    if drop:
        flat = luma.view(-1, 720)
        flat[drop::drop * 7] = 16          # a lost line now and then
    eng = Engine(0); eng.set_profiling(True)
    eng.binarize_frames(luma, first_frame_no=1, new_file=True)
    res = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.binarize_frames(luma, first_frame_no=1 + (it + 1) * n, new_file=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        i = eng.run_info()
        res.append((round(dt, 2), round(i.kernel_ms, 2), i.rounds, i.frames_launched, i.frames_general))
    print(f"sigma {sigma} dropout-every {drop}: (wall ms, kernel ms, rounds, launched, by full kernel) {res[-1]}", flush=True)
