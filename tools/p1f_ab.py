"""A/B of two builds of the library on the frame drivers: same records? usage: p1f_ab.py <other.so> [frames]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdvpcmdecoder_amd import Engine, synth, load_library
other = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for name, gen, call, off in (("pcm1", synth.pcm1_frames, "pcm1_binarize_frames", 36), ("pcm16x0", synth.pcm16x0_frames, "pcm16x0_binarize_frames", 34)):
    base, _ = gen(8, seed=530, height=486, noise_sigma=4.0)
    d = torch.from_numpy(np.tile(base, ((n + 7) // 8, 1, 1))[:n]).to("cuda:0")
    for mode in (0, 1, 2):
        outs = []
        for lib in (None, other):
            eng = Engine(0, lib=load_library(lib)) if lib else Engine(0)
            eng.setBinarizationMode(mode)
            res = []
            for c in range(2):
                lines, stats = getattr(eng, call)(d, first_frame_no=1 + c * n, new_file=(c == 0))
                torch.cuda.synchronize()
                res.append((lines.cpu().numpy().copy(), stats.cpu().numpy().copy(), eng.run_info().rounds))
            outs.append(res)
        for c in range(2):
            a, b = outs[0][c], outs[1][c]
            fl = a[0].reshape(len(a[0]), -1)[:, off]
            print(name, "mode", mode, "call", c, "same" if a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes() else "DIFF", "valid", int(((fl & 64) != 0).sum()), "of", len(fl), "rounds", a[2], b[2], flush=True)
