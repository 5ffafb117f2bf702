"""Experiment driver: steady-state timing of an alternative build of the library whose frame kernel has no general path
(build/variants/lib_*.so): the stream is started by the product library, its chain state is handed to the variant."""
import sys, time, glob
sys.path.insert(0, '.')
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth, load_library, LINE_DTYPE
n = 10000
luma, w9 = synth.stc007_frames_torch(n, seed=2, device='cuda', noise_sigma=4.0, cyclic=True)
H = 486
out = torch.empty((n * (H + 3) + 1, 48), dtype=torch.uint8, device='cuda'); st = torch.empty((n, 32), dtype=torch.uint8, device='cuda')
base = Engine(0)
base.binarize_frames(luma, first_frame_no=1, new_file=True, out_lines=out, out_stats=st)
state = base.get_chain_state()
ref = None
for path in [None] + sorted(glob.glob('build/variants/lib_*.so')):
    eng = Engine(0, lib=load_library(path)) if path else Engine(0)
    eng.set_chain_state(state)
    eng.set_profiling(True)
    ms = []
    for it in range(6):
        eng.set_chain_state(state)
        eng.binarize_frames(luma, first_frame_no=1 + n, new_file=False, out_lines=out[1:], out_stats=st)
        ms.append(eng.run_info().kernel_ms)
    got = out[1:1 + n * (H + 3)].cpu().numpy().tobytes()
    if ref is None: ref = got
    print(path or 'product', 'kernel ms', ['%.3f' % m for m in ms[1:]], 'rounds', eng.run_info().rounds, 'same as product', got == ref, flush=True)
