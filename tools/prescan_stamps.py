"""Developer aid: where a prescan line (sdv_k_pcm1_prescan) spends its cycles - library built with -DSDV_K1_STAMPS, passed in SDVPCM_LIB.
usage: prescan_stamps.py [frames] [pcm1|pcm16x0]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fmt = sys.argv[2] if len(sys.argv) > 2 else "pcm1"
base, words = (synth.pcm1_frames if fmt == "pcm1" else synth.pcm16x0_frames)(8, seed=530, height=486, noise_sigma=4.0)
luma = torch.from_numpy(np.tile(base, ((n + 7) // 8, 1, 1))[:n]).to("cuda:0")
eng = Engine(0); eng.setBinarizationMode(2)
ol = torch.empty((n * (489 if fmt == "pcm1" else 3 * 486 + 3) + 1, 40 if fmt == "pcm1" else 36), dtype=torch.uint8, device="cuda:0"); os_ = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
run = eng.pcm1_binarize_frames if fmt == "pcm1" else eng.pcm16x0_binarize_frames
run(luma, first_frame_no=1, new_file=True, out_lines=ol, out_stats=os_)
out = (C.c_ulonglong * 24)()
eng.lib.sdv_debug_k1_cycles(out, 1)
run(luma, first_frame_no=1 + n, out_lines=ol[1:], out_stats=os_)
torch.cuda.synchronize()
eng.lib.sdv_debug_k1_cycles(out, 1)
names = {16: "row staged", 17: "black / white levels", 18: "search: where to start (PCM-16x0: rows with a valid read)", 19: "search: candidate reads", 20: "search: votes", 21: "search: last candidate once more",
         22: "the read with the coordinates found", 23: "whole line"}
lines = 4 * n
for i in range(16, 24):
    print(f"{names[i]:40s} {out[i] / lines:10.0f} cycles per line  {100.0 * out[i] / max(1, out[23]):5.1f} %")
