"""One file of n frames (NEW_FILE .. END_FILE) through the fused entry in one call and in calls of `chunk` frames: what feeding a long file in parts is worth.
usage: chunk_probe.py [frames] [chunk]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
luma, _ = synth.stc007_frames_torch(n, seed=3, device='cuda', noise_sigma=4.0, cyclic=True)
eng = Engine(0); eng.setPCMType(2)
fp = torch.empty(((n + 2) * 1800 + 8192, 12), dtype=torch.uint8, device='cuda')
ff = torch.empty((n + 64, 64), dtype=torch.uint8, device='cuda')
fs = torch.empty((n + 1, 32), dtype=torch.uint8, device='cuda')
for it in range(3):
    eng.reset_stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p, f, st = eng.decode_frames(2, luma, first_frame_no=1, new_file=True, end_file=True, out_pairs=fp, out_frames=ff, out_stats=fs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    whole = p.clone() if it == 0 else whole
    si = eng.stitch_info(); ri = eng.run_info()
    print(f"one call: {dt * 1e3:.2f} ms, {p.shape[0]} pairs; binarize rounds {ri.rounds} frames launched {ri.frames_launched}; stitch rounds {si.rounds} turns launched {si.steps_launched} of {si.steps}", flush=True)
for it in range(3):
    eng.reset_stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    np_ = 0; nf = 0
    for k in range(0, n, chunk):
        m = min(chunk, n - k)
        p, f, st = eng.decode_frames(2, luma[k:k + m], first_frame_no=1 + k, new_file=k == 0, end_file=k + m == n, out_pairs=fp[np_:], out_frames=ff[nf:], out_stats=fs[k:])
        np_ += p.shape[0]; nf += f.shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    same = np_ == whole.shape[0] and bool((fp[:np_] == whole).all())
    print(f"calls of {chunk}: {dt * 1e3:.2f} ms, {np_} pairs, same as one call: {same}", flush=True)
