"""Cost of lost lines in the binarize stage: a 10 000-frame continuing tape with D dropouts (one black line each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
luma0, _ = synth.stc007_frames_torch(n, seed=3, device='cuda', noise_sigma=4.0, cyclic=True)
for D in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,4,16,64".split(","))]:
    luma = luma0.clone()
    if D:
        rng = np.random.default_rng(D)
        for f in sorted(rng.choice(np.arange(50, n - 50), size=D, replace=False)):
            luma[int(f), int(rng.integers(40, 440))] = 16
    eng = Engine(0); eng.set_profiling(True)
    eng.binarize_frames(luma, first_frame_no=1, new_file=True)
    res = []
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.binarize_frames(luma, first_frame_no=1 + (it + 1) * n, new_file=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        i = eng.run_info()
        res.append((round(dt, 2), round(i.kernel_ms, 2), i.rounds, i.frames_launched, i.frames_general))
    print(f"dropouts {D}: (wall ms, kernel ms, rounds, launched, by full kernel) {res[-1]}", flush=True)
