import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = 10000
luma, _ = synth.stc007_frames_torch(n, seed=2, device="cuda", noise_sigma=4.0, cyclic=True)
rng = np.random.default_rng(16)
if len(sys.argv) > 1 and sys.argv[1] == "bench":       # bench.py's window_jumps tape: its generator has drawn the lost-lines case first
    for f in sorted(rng.choice(np.arange(50, n - 50), size=16, replace=False)):
        rng.integers(40, 440)
lum = luma.clone(); at = 0; js = []
for f in sorted(rng.choice(np.arange(50, n - 50), size=16, replace=False)):
    to = at
    while to == at: to = int(rng.integers(-8, 9))
    lum[int(f):] = torch.roll(luma[int(f):], to, dims=2); at = to; js.append((int(f), to))
print("jumps", js, flush=True)
eng = Engine(0); eng.set_profiling(True); eng.setBinarizationMode(2)
eng.binarize_frames(luma, first_frame_no=1, new_file=True)
eng.binarize_frames(luma, first_frame_no=1 + n)
os.environ["SDV_SCHED_TRACE"] = "1"
eng.binarize_frames(lum, first_frame_no=1 + 2 * n)
i = eng.run_info(); print(i.rounds, i.kernel_ms)
