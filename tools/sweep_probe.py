"""Cost of lines that do not read (a bit cell inverted: the CRC fails at every reference level) in the STC-007 frame kernels.
   usage: sweep_probe.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
luma, _ = synth.stc007_frames_torch(n, seed=2, device="cuda", noise_sigma=4.0, cyclic=True)
for mode in (2, 1):
    for per_frame in (1, 4, 16):
        lum = luma.clone()
        g = torch.Generator(device="cuda"); g.manual_seed(per_frame)
        rows = torch.randint(20, 460, (n, per_frame), generator=g, device="cuda")
        xs = 12 + (torch.randint(4, 132, (n, per_frame), generator=g, device="cuda") * (720 - 24)) // 137
        fi = torch.arange(n, device="cuda")[:, None].expand(n, per_frame)
        for dx in range(5):
            lum[fi, rows, xs + dx] = (230 - lum[fi, rows, xs + dx].to(torch.int16)).clamp_(0, 255).to(torch.uint8)
        eng = Engine(0); eng.set_profiling(True); eng.setBinarizationMode(mode)
        eng.binarize_frames(luma, first_frame_no=1, new_file=True)
        eng.binarize_frames(luma, first_frame_no=1 + n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.binarize_frames(lum, first_frame_no=1 + 2 * n)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        i = eng.run_info()
        print(f"mode {mode}, {per_frame} unreadable line(s) per frame, {n} frames: {dt:.1f} ms wall, {i.kernel_ms:.1f} ms kernels, {i.rounds} rounds, {i.frames_launched} decodes, {i.frames_general} full", flush=True)
