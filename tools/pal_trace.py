"""Scheduler trace (developer build, SDV_SCHED_TRACE=1) of the PAL tape of SURVEY 8d C3: usage pal_trace.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
what = sys.argv[2] if len(sys.argv) > 2 else "both"
dev = "cuda"
pal, _ = synth.stc007_frames_torch(n, seed=7, device=dev, width=720, height=576, lines_per_field=294, noise_sigma=4.0, cyclic=True)
lum = pal.clone()
if what in ("both", "lost"):
    lum[:, 96::97, :] = 16
if what in ("both", "cells"):
    flat = lum.view(-1, 720)
    g = torch.Generator(device=dev); g.manual_seed(53)
    rows = torch.arange(0, flat.shape[0], 53, device=dev)
    xs = 12 + (torch.randint(4, 132, rows.shape, generator=g, device=dev) * (720 - 24)) // 137
    for dx in range(5):
        flat[rows, xs + dx] = (230 - flat[rows, xs + dx].to(torch.int16)).clamp_(0, 255).to(torch.uint8)
eng = Engine(0); eng.set_profiling(True); eng.setBinarizationMode(2)
eng.binarize_frames(pal, first_frame_no=1, new_file=True)
eng.binarize_frames(pal, first_frame_no=1 + n)
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.binarize_frames(lum, first_frame_no=1 + 2 * n)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
i = eng.run_info()
print(f"{what}: {dt:.2f} ms wall, {i.kernel_ms:.2f} ms kernels, {i.rounds} rounds, {i.frames_launched} frame decodes, {i.frames_general} by the full kernel ({i.frames_met} met their last pass), {i.sweeps} sweeps", flush=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.binarize_frames(lum, first_frame_no=1 + (3 + rep) * n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    i = eng.run_info()
    print(f"  again: {dt:.2f} ms wall, {i.kernel_ms:.2f} ms kernels, {i.rounds} rounds, {i.frames_launched} frame decodes, {i.frames_general} by the full kernel ({i.frames_met} met their last pass), {i.sweeps} sweeps -> {n / dt * 1e3:.0f} frames/s", flush=True)
