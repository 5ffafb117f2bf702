"""Timing driver of the visualiser canvases: N synthetic STC-007 NTSC frames -> sdv_binarize_frames -> sdv_vis_render_lines, `reps` timed
calls of the latter.  Prints wall time per call and the rate of canvas bytes written (4 B per pixel, 685 x 650 per frame)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
from sdvpcmdecoder_amd.engine import VIS_STC007_LINES
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
luma = synth.stc007_frames_torch(n, seed=3, device="cuda")[0]
eng = Engine(0)
recs, _ = eng.binarize_frames(luma, new_file=True)
recs = recs.contiguous()
w, h = eng.vis_canvas_size(VIS_STC007_LINES)
for it in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.vis_render_lines(VIS_STC007_LINES, recs, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"n={n} it={it}: wall {dt*1e3:.3f} ms, {n/dt/1e3:.1f} K frames/s, {n*w*h*4/dt/1e9:.0f} GB/s of canvas written, canvases {tuple(out.shape)}", flush=True)
    del out
# the data blocks window: the stitch stage with its block output on, then the block canvases
import numpy as np
from sdvpcmdecoder_amd.engine import VIS_STC007_BLOCKS_NTSC
buf = torch.empty((n * 490 + 1024, 72), dtype=torch.uint8, device="cuda")
eng.stitch_frames(recs)                       # warm: the stream's first call
recs2, _ = eng.binarize_frames(luma, first_frame_no=1 + n)
for on in (False, True):
    eng.set_stitch_block_output(buf if on else None)
    for it in range(2):
        r, _ = eng.binarize_frames(luma, first_frame_no=1 + (2 + it + 2 * on) * n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pairs, frames = eng.stitch_frames(r.contiguous())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"stitch_frames, block output {'on' if on else 'off'}: {dt*1e3:.3f} ms per {n} frames, {eng.stitch_block_count()} blocks", flush=True)
nb = eng.stitch_block_count()
fr = frames.cpu().numpy()
per = np.frombuffer(fr.tobytes(), dtype=np.uint8).reshape(-1, 64)
w2, h2 = eng.vis_canvas_size(VIS_STC007_BLOCKS_NTSC)
per_frame = np.full(nb // 490, 490, dtype=np.uint32)
for it in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.vis_render_blocks(VIS_STC007_BLOCKS_NTSC, buf[:nb], per_frame)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"blocks n={len(per_frame)} it={it}: wall {dt*1e3:.3f} ms, {len(per_frame)/dt/1e3:.1f} K frames/s, {len(per_frame)*w2*h2*4/dt/1e9:.0f} GB/s of canvas written", flush=True)
    del out
