"""Timing driver of the PCM-1 back half: N synthetic PCM-1 frames (a 100-frame damaged tape tiled, frame numbers continued)
-> sdv_pcm1_stitch_frames, `reps` timed calls.  Prints wall time per call and the algorithmic-bytes rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import pcm1_api as p1
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
recs = synth.pcm1_tape(n)
d = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 32)).cuda()
eng = Engine(0)
out_p = torch.empty((n * 1470 + 64, 12), dtype=torch.uint8, device='cuda')
out_f = torch.empty((n + 64, 52), dtype=torch.uint8, device='cuda')
alg = len(recs) * 32 + n * (1470 * 12 + 52)
for it in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p, f = eng.pcm1_stitch_frames(d, out_pairs=out_p, out_frames=out_f)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"n={n} it={it}: wall {dt*1e3:.3f} ms, {n/dt/1e6:.2f} M frames/s, {alg/dt/1e9:.0f} GB/s algorithmic, pairs {p.shape[0]} frames {f.shape[0]}", flush=True)
