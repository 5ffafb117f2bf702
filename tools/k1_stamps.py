"""Developer aid: cycles per part of a frame in the lean frame kernel (needs a library built with -DSDV_K1_STAMPS, passed in SDVPCM_LIB)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = 10000
eng = Engine(0)
luma, _ = synth.stc007_frames_torch(n, seed=2, device='cuda', noise_sigma=4.0, cyclic=True)
eng.binarize_frames(luma, first_frame_no=1, new_file=True)
out = (C.c_ulonglong * 24)()
eng.lib.sdv_debug_k1_cycles(out, 1)
eng.binarize_frames(luma, first_frame_no=1 + n, new_file=False)
torch.cuda.synchronize()
eng.lib.sdv_debug_k1_cycles(out, 0)
v = [x / n for x in out]
print(f"cycles per frame: total {v[0]:.0f}  batch loops {v[1]:.0f}  batch_finish {v[2]:.0f}  end of frame {v[3]:.0f}  "
      f"capture (all) {v[4]:.0f} of which solve {v[5]:.0f}, field-0 batch_finish {v[6]:.0f}, gather loop + field 1 {v[4]-v[5]-v[6]:.0f}  rest {v[0]-v[1]-v[2]-v[3]-v[4]:.0f}")
