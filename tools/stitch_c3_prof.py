"""Stitch stage on the PAL tape of SURVEY 8d C3 (every 97th line lost, a cell inverted on one line in 53): usage stitch_c3_prof.py [frames] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda"
lum, _ = synth.stc007_frames_torch(n, seed=7, device=dev, width=720, height=576, lines_per_field=294, noise_sigma=4.0, cyclic=True)
lum[:, 96::97, :] = 16
flat = lum.view(-1, 720)
g = torch.Generator(device=dev); g.manual_seed(53)
rows = torch.arange(0, flat.shape[0], 53, device=dev)
xs = 12 + (torch.randint(4, 132, rows.shape, generator=g, device=dev) * (720 - 24)) // 137
for dx in range(5):
    flat[rows, xs + dx] = (230 - flat[rows, xs + dx].to(torch.int16)).clamp_(0, 255).to(torch.uint8)
eng = Engine(0); eng.setBinarizationMode(2)
out_p = torch.empty((n * 1764 + 65536, 12), dtype=torch.uint8, device=dev)
out_f = torch.empty((n + 64, 64), dtype=torch.uint8, device=dev)
fn = 1
for it in range(reps + 1):
    lines, _ = eng.binarize_frames(lum, first_frame_no=fn, new_file=(it == 0))
    fn += n
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p, f = eng.stitch_frames(lines, out_pairs=out_p, out_frames=out_f)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    info = eng.stitch_info()
    print(f"n={n} it={it}: wall {dt*1e3:.2f} ms device {info.device_ms:.2f} ms steps {info.steps} rounds {info.rounds} piped {info.pipelined} launched {info.steps_launched} pairs {p.shape[0]}", flush=True)
