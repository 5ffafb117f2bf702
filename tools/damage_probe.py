"""Binarize stage on damaged tapes (the workloads of bench.py's damaged_tape / pal_stage objects, one line each):
   usage: damage_probe.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = "cuda"
def run(name, clean, lum, reps=2):
    eng = Engine(0); eng.set_profiling(True); eng.setBinarizationMode(2)
    eng.binarize_frames(clean, first_frame_no=1, new_file=True)
    for r in range(reps):
        eng.binarize_frames(clean, first_frame_no=1 + (2 * r + 1) * n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.binarize_frames(lum, first_frame_no=1 + (2 * r + 2) * n)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        i = eng.run_info()
    print(f"{name}: {dt:.2f} ms wall, {i.kernel_ms:.2f} ms kernels, {i.rounds} rounds, {i.frames_launched} frame decodes, {i.frames_general} by the full kernel -> {n / dt * 1e3:,.0f} frames/s", flush=True)
luma, _ = synth.stc007_frames_torch(n, seed=2, device=dev, noise_sigma=4.0, cyclic=True)
rng = np.random.default_rng(16)
for per in (16, 160):
    lum = luma.clone()
    for f in sorted(rng.choice(np.arange(50, n - 50), size=per * n // 10000, replace=False)):
        lum[int(f), int(rng.integers(40, 440))] = 16
    run(f"NTSC {per} lost lines per 10 000 frames", luma, lum)
for per in (1, 16):
    lum = luma.clone(); at = 0
    for f in sorted(rng.choice(np.arange(50, n - 50), size=max(1, per * n // 10000), replace=False)):
        to = at
        while to == at: to = int(rng.integers(-8, 9))
        lum[int(f):] = torch.roll(luma[int(f):], to, dims=2); at = to
    run(f"NTSC {per} window jumps per 10 000 frames", luma, lum)
lum = luma.clone(); lum[:, 96::97, :] = 16
run("NTSC every 97th line of every frame lost", luma, lum, reps=1)
del lum, luma
pal, _ = synth.stc007_frames_torch(n, seed=7, device=dev, width=720, height=576, lines_per_field=294, noise_sigma=4.0, cyclic=True)
lum = pal.clone(); lum[:, 96::97, :] = 16
flat = lum.view(-1, 720)
g = torch.Generator(device=dev); g.manual_seed(53)
rows = torch.arange(0, flat.shape[0], 53, device=dev)
xs = 12 + (torch.randint(4, 132, rows.shape, generator=g, device=dev) * (720 - 24)) // 137
for dx in range(5):
    flat[rows, xs + dx] = (230 - flat[rows, xs + dx].to(torch.int16)).clamp_(0, 255).to(torch.uint8)
run("PAL clean", pal, pal)
run("PAL every 97th line lost + a cell inverted per 53 lines", pal, lum, reps=1)
