"""Cycle sums of the PCM-16x0 frame kernel's stages (library built with -DSDV_P16_STAMPS=1, handed over as SDVPCM_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
base, _ = synth.pcm16x0_frames(8, seed=530, height=486, width=720, noise_sigma=4.0)
fl = torch.from_numpy(np.tile(base, ((n + 7) // 8, 1, 1))[:n]).cuda()
eng = Engine(0)
eng.setPCMType(1); eng.setBinarizationMode(mode)
eng.pcm16x0_binarize_frames(fl, first_frame_no=1, new_file=True)
lines, stats = eng.pcm16x0_binarize_frames(fl, first_frame_no=1 + n)
st = stats.cpu().numpy().view(np.uint64).reshape(n, 4).astype(np.float64)
m = st[10:].mean(axis=0)
print("cycles per frame: rows staged %.0f, parts decoded %.0f, per-part bookkeeping %.0f, records stored %.0f; per part: %.0f / %.0f / %.0f" % (m[0], m[1], m[2], m[3], m[1] / 1458, m[2] / 1458, m[3] / 1458))
