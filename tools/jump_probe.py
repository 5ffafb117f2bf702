"""Cost of a jump of the data window in the middle of a batch (binarize stage): a 10 000-frame continuing tape whose picture moves
sideways by a few pixels from frame n/2 on, and a tape where that happens J times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
luma0, _ = synth.stc007_frames_torch(n, seed=3, device='cuda', noise_sigma=4.0, cyclic=True)
for J in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,4,16".split(","))]:
    luma = luma0.clone()
    rng = np.random.default_rng(J)
    at = 0                                                  # where the window is, relative to the first call's (kept within +-8 px)
    for f in sorted(rng.choice(np.arange(50, n - 50), size=J, replace=False)):
        to = at
        while to == at: to = int(rng.integers(-8, 9))
        luma[int(f):] = torch.roll(luma0[int(f):], to, dims=2); at = to
    eng = Engine(0); eng.set_profiling(True)
    eng.binarize_frames(luma0, first_frame_no=1, new_file=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.binarize_frames(luma, first_frame_no=1 + n, new_file=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    i = eng.run_info()
    print(f"jumps {J}: (wall ms, kernel ms, rounds, launched, by full kernel, sweeps) {(round(dt, 2), round(i.kernel_ms, 2), i.rounds, i.frames_launched, i.frames_general, i.sweeps)}", flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.binarize_frames(luma, first_frame_no=1 + 2 * n, new_file=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    i = eng.run_info()
    print(f"   once more: {(round(dt, 2), round(i.kernel_ms, 2), i.rounds, i.frames_launched, i.frames_general, i.sweeps)}", flush=True)
