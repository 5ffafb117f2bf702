"""Timing driver of the PCM-16x0 back half: N synthetic PCM-1630 frames (a 50-frame damaged tape tiled, frame numbers continued)
-> sdv_pcm16x0_stitch_frames, `reps` timed calls per interleave format.  Prints wall time per call and the algorithmic-bytes rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
formats = sys.argv[3].split(',') if len(sys.argv) > 3 else ['si', 'ei']
for fmt in formats:
    ei = fmt == 'ei'
    recs = synth.pcm16x0_tape(n, ei=ei)
    d = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 36)).cuda()
    eng = Engine(0)
    st = eng.default_pcm16x0_stitch_settings()
    st.format = 2 if ei else 1
    out_p = torch.empty((n * 1470 + 64, 12), dtype=torch.uint8, device='cuda')
    out_f = torch.empty((n + 64, 56), dtype=torch.uint8, device='cuda')
    alg = len(recs) * 36 + n * (1470 * 12 + 56)
    for it in range(reps):
        eng.set_pcm16x0_stitch_settings(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p, f = eng.pcm16x0_stitch_frames(d, out_pairs=out_p, out_frames=out_f)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = int(((f.cpu().numpy()[:, 54] & 32) != 0).sum())
        print(f"{fmt} n={n} it={it}: wall {dt*1e3:.3f} ms, {n/dt/1e3:.1f} k frames/s, {alg/dt/1e9:.1f} GB/s algorithmic, pairs {p.shape[0]} frames {f.shape[0]} padding ok {ok}", flush=True)
