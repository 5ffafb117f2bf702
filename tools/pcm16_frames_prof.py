"""PCM-16x0 frame driver on the GPU (sdv_pcm16x0_binarize_frames): frames per second per Binarizer mode on a tape that plays.
usage: pcm16_frames_prof.py [frames] [reps]"""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
base, words = synth.pcm16x0_frames(8, seed=530, height=486, noise_sigma=4.0)
luma = torch.from_numpy(np.tile(base, ((n + 7) // 8, 1, 1))[:n]).to("cuda:0")
eng = Engine(0)
eng.set_profiling(True)
ol = torch.empty((n * 1461 + 1, 36), dtype=torch.uint8, device="cuda:0"); os_ = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
for mode in (0, 1, 2):
    eng.setBinarizationMode(mode); eng.reset_stream()
    eng.pcm16x0_binarize_frames(luma, first_frame_no=1, new_file=True, out_lines=ol, out_stats=os_)
    torch.cuda.synchronize()
    i0 = eng.run_info()
    t0 = time.perf_counter(); ms = 0.0; rounds = 0
    for r in range(reps):
        eng.pcm16x0_binarize_frames(luma, first_frame_no=1 + (r + 1) * n, out_lines=ol[1:], out_stats=os_)
        i = eng.run_info(); ms += i.kernel_ms; rounds += i.rounds
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    rec = ol[1:1 + n * 1461].cpu().numpy().reshape(-1).view(np.dtype([("f", "<u4"), ("l", "<u2"), ("w", "<u2", (4,)), ("rest", "u1", (22,))]))
    valid = int(((rec["rest"][:, 18] & 64) != 0).sum())
    print(f"mode {mode}: first call {i0.rounds} rounds {i0.kernel_ms:.2f} ms; continuing {dt * 1e3:.2f} ms wall, {ms / reps:.2f} ms device, {rounds / reps:.1f} rounds -> {n / dt:,.0f} frames/s, "
          f"{n * (486 * 720 + 1461 * 36) / dt / 1e9:.1f} GB/s algorithmic; valid lines {valid}/{n * 1458}")
