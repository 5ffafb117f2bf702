"""Profiling driver of the fused entry: N synthetic NTSC frames, a continuing tape, `reps` calls of sdv_decode_frames (frames -> PCMSamplePair)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = Engine(0); eng.setPCMType(2)
luma, _ = synth.stc007_frames_torch(n, seed=2, device='cuda', noise_sigma=4.0, cyclic=True)
fp = torch.empty(((n + 2) * 1800 + 8192, 12), dtype=torch.uint8, device='cuda')
ff = torch.empty((n + 64, 64), dtype=torch.uint8, device='cuda')
fs = torch.empty((n, 32), dtype=torch.uint8, device='cuda')
fn = 1
for it in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p, f, st = eng.decode_frames(2, luma, first_frame_no=fn, new_file=it == 0, out_pairs=fp, out_frames=ff, out_stats=fs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fn += n
    print(f"n={n} it={it}: wall {dt*1e3:.3f} ms pairs {p.shape[0]} pipelined {eng.stitch_info().pipelined} direct {eng.stitch_info().direct_frames}", flush=True)
