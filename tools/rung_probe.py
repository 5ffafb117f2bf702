"""A tape that sits a pixel or two beside the coordinates the binarizer holds (every line reads, but on another rung of the shift ladder): the steady call of
sdv_binarize_frames over 10 000 such frames, per shift.  usage: rung_probe.py [frames] [shifts, comma separated]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth, LINE_DTYPE
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
shifts = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,-1,2,-2".split(","))]
luma0, _ = synth.stc007_frames_torch(n, seed=3, device='cuda', noise_sigma=4.0, cyclic=True)
for sh in shifts:
    luma = torch.roll(luma0, sh, dims=2) if sh else luma0
    eng = Engine(0); eng.set_profiling(True); eng.setBinarizationMode(2)
    eng.binarize_frames(luma0, first_frame_no=1, new_file=True)            # the chain is tuned to the tape where it was
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lines, _ = eng.binarize_frames(luma, first_frame_no=1 + (1 + rep) * n)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    i = eng.run_info()
    r = lines.cpu().numpy().view(LINE_DTYPE).reshape(-1)
    r = r[r["service_type"] == 0]
    stages = np.bincount(r["shift_stage"], minlength=5)
    print(f"shift {sh:+d}: calls {[round(t, 2) for t in ts]} ms; last: {i.rounds} rounds, {i.frames_launched} frame decodes, {i.frames_general} by the full kernel, kernels {i.kernel_ms:.2f} ms; lines per shift stage {stages.tolist()}", flush=True)
