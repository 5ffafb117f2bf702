"""BASELINE configs[4] on one GPU without the rest of bench.py: one N-frame NTSC stream, NEW_FILE .. END_FILE, through sdv_decode_frames in one call.
usage: configs4_probe.py [frames] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = Engine(0); eng.setBinarizationMode(2); eng.setPCMType(2)
chunks = []
luma = torch.empty((n, 486, 720), dtype=torch.uint8, device="cuda")
step = 10000
for lo in range(0, n, step):
    hi = min(n, lo + step)
    luma[lo:hi] = synth.stc007_frames_torch(n, seed=5, device="cuda", noise_sigma=4.0, frame_range=(lo, hi))[0]
out_p = torch.empty(((n + 2) * 1800 + 8192, 12), dtype=torch.uint8, device="cuda")
out_f = torch.empty((n + 64, 64), dtype=torch.uint8, device="cuda")
out_s = torch.empty((n + 1, 32), dtype=torch.uint8, device="cuda")
for r in range(reps):
    eng.reset_stream(); eng.reset_stitcher()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p, f, st = eng.decode_frames(2, luma, first_frame_no=1, new_file=True, end_file=True, out_pairs=out_p, out_frames=out_f, out_stats=out_s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    i = eng.stitch_info()
    print(f"rep {r}: {dt:.2f} ms = {n / dt / 1e3:.2f} M frames/s; pairs {p.shape[0]}, last span: pipelined {i.pipelined}, direct frames {i.direct_frames}, rounds {i.rounds}", flush=True)
