"""Profiling driver of the stitch stage: N synthetic NTSC frames -> binarize -> stitch (3 timed calls)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cont = len(sys.argv) > 3 and sys.argv[3] == "cont"      # a continuing (seamless) tape instead of a cold start every time
eng = Engine(0)
luma, _ = synth.stc007_frames_torch(n, seed=2, device='cuda', noise_sigma=4.0, cyclic=True)
lines, _ = eng.binarize_frames(luma, first_frame_no=1, new_file=True)
torch.cuda.synchronize()
eng.set_profiling(True)
out_p = torch.empty((n * 1470 + 20000, 12), dtype=torch.uint8, device='cuda')
out_f = torch.empty((n + 64, 64), dtype=torch.uint8, device='cuda')
fn = 1 + n
if len(sys.argv) > 4 and sys.argv[4] == "blocks":        # with the visualiser's block output on
    blk = torch.empty((n * 490 + 4096, 72), dtype=torch.uint8, device='cuda'); eng.set_stitch_block_output(blk)
for it in range(reps):
    if cont and it > 0:
        lines2, _ = eng.binarize_frames(luma, first_frame_no=fn, new_file=False)
        fn += n
        src = lines2
    else:
        eng.reset_stitcher()
        src = lines
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p, f = eng.stitch_frames(src, out_pairs=out_p, out_frames=out_f)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    info = eng.stitch_info()
    print(f"n={n} it={it}: wall {dt*1e3:.2f} ms device {info.device_ms:.2f} ms steps {info.steps} rounds {info.rounds} piped {info.pipelined} launched {info.steps_launched} pairs {p.shape[0]}", flush=True)
