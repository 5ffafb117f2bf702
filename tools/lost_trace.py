"""Scheduler trace (developer build, SDV_SCHED_TRACE=1) of the bench's lost-lines tape: 16 lost lines per 10 000 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = 10000
luma, _ = synth.stc007_frames_torch(n, seed=2, device="cuda", noise_sigma=4.0, cyclic=True)
rng = np.random.default_rng(16)
lum = luma.clone(); ev = []
for f in sorted(rng.choice(np.arange(50, n - 50), size=16, replace=False)):
    r = int(rng.integers(40, 440)); lum[int(f), r] = 16; ev.append((int(f), r))
print("lost", ev, flush=True)
eng = Engine(0); eng.set_profiling(True); eng.setBinarizationMode(2)
eng.binarize_frames(luma, first_frame_no=1, new_file=True)
eng.binarize_frames(luma, first_frame_no=1 + n)
os.environ["SDV_SCHED_TRACE"] = "1"
eng.binarize_frames(lum, first_frame_no=1 + 2 * n)
i = eng.run_info(); print(i.rounds, i.kernel_ms, i.frames_launched)
