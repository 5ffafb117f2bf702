"""Developer aid: where the general path of the full frame kernel spends its cycles on the tape of SURVEY 8d C3 (library built with
-DSDV_K1_STAMPS, passed in SDVPCM_LIB): usage slow_stamps.py [frames] [both|cells|lost]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
what = sys.argv[2] if len(sys.argv) > 2 else "both"
dev = "cuda"
if what == "jumps":     # the window-jump tape of bench.py (damaged_tape.window_jumps): NTSC, the picture moves sideways at 16 places per 10 000 frames
    import numpy as np
    pal, _ = synth.stc007_frames_torch(n, seed=2, device=dev, noise_sigma=4.0, cyclic=True)
    lum = pal.clone()
    rng_d = np.random.default_rng(16); at = 0
    for f_ in sorted(rng_d.choice(np.arange(50, n - 50), size=max(1, n // 625), replace=False)):
        to = at
        while to == at:
            to = int(rng_d.integers(-8, 9))
        lum[int(f_):] = torch.roll(pal[int(f_):], to, dims=2); at = to
else:
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
eng.binarize_frames(lum, first_frame_no=1 + n)
if what == "jumps":
    eng.binarize_frames(pal, first_frame_no=1 + n)      # back to the clean tape's state
out = (C.c_ulonglong * 24)()
eng.lib.sdv_debug_k1_cycles(out, 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.binarize_frames(lum, first_frame_no=1 + 2 * n)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
eng.lib.sdv_debug_k1_cycles(out, 0)
i = eng.run_info()
print(f"{what}: {dt:.2f} ms wall, {i.kernel_ms:.2f} ms kernels, {i.rounds} rounds, {i.frames_launched} frame decodes, {i.frames_general} by the full kernel, {i.sweeps} sweeps")
v = list(out)
fr = max(1, i.frames_general)
names = {0: "frame total (sum)", 1: "batch loops", 2: "batch_finish", 3: "end of frame", 8: "slow_line total", 9: "find_black_white", 10: "find_coordinates (INPUT_LEVEL)",
         11: "read_pcm_data (INPUT_ALL)", 12: "sweep lookup + apply", 13: "post_line + record", 14: "fast_line attempts that failed",
         20: "general-path detour incl. context copies", 21: "row staging ahead of sequential lines", 4: "fast_line calls that took the line", 6: "fast_decode (ladder of reads)",
         16: "fast_line: dup check", 17: "fast_line: 9-line window + key", 18: "fast_line: damper", 19: "fast_line: counters + record"}
print(f"lines through fast_line one by one (full kernel): {v[5]} ({v[5] / fr:.1f} per frame), {v[4] / max(1, v[5]):.0f} cycles each, of which the ladder of reads {v[6] / max(1, v[5]):.0f}")
print(f"   bookkeeping of a line taken by fast_line (lean + full, per call): dup check {v[16] / max(1, v[5]):.0f}, 9-line window + key {v[17] / max(1, v[5]):.0f}, damper {v[18] / max(1, v[5]):.0f}, counters + record {v[19] / max(1, v[5]):.0f}")
print(f"lines that measured black and white ahead of the fast read (fast_line<true>): {v[23]} ({v[23] / fr:.1f} per frame), {v[22] / max(1, v[23]):.0f} cycles each for findBlackWhite")
print(f"slow lines: {v[15]} ({v[15] / fr:.1f} per full-kernel frame); the slowest frame: index {v[7] & 0x3FFF}, {v[7] >> 24} cycles with {(v[7] >> 14) & 0x3FF} slow lines")
import numpy as np
from sdvpcmdecoder_amd import LINE_DTYPE
fs = int(v[7] & 0x3FFF)
lines, _ = eng.binarize_frames(lum, first_frame_no=1 + 3 * n)
hh = lum.shape[1] + 3
r = lines.cpu().numpy().view(LINE_DTYPE).reshape(-1)[fs * hh:(fs + 1) * hh]
r = r[r["service_type"] == 0]
import collections
print("straggler frame records: flags histogram", collections.Counter(r["flags"].tolist()).most_common(8))
print("ref levels", collections.Counter(r["ref_level"].tolist()).most_common(8))
print("first 40 (line, flags, ref, black, white, start, stop):", [(int(x["line_number"]), int(x["flags"]), int(x["ref_level"]), int(x["black_level"]), int(x["white_level"]), int(x["data_start"]), int(x["data_stop"])) for x in r[:40]])
for k, nm in names.items():
    print(f"  {nm:34s} {v[k] / fr:12.0f} cycles per full-kernel frame" + (f"  {v[k] / max(1, v[15]):10.0f} per slow line" if k >= 8 else ""))
