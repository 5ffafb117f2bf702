"""One cold call of sdv_pcm16x0_binarize_frames (for kernel-time comparisons of alternative builds under rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
base, _ = synth.pcm16x0_frames(8, seed=530, height=486, width=720, noise_sigma=4.0)
fl = torch.from_numpy(np.tile(base, ((n + 7) // 8, 1, 1))[:n]).cuda()
eng = Engine(0)
eng.setPCMType(1); eng.setBinarizationMode(2)
try:
    eng.pcm16x0_binarize_frames(fl, first_frame_no=1, new_file=True)
except Exception as ex:
    print("call failed:", ex)
print("frames/s n/a")
