import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n=int(sys.argv[1]) if len(sys.argv)>1 else 200
eng=Engine(0)
luma,_=synth.stc007_frames_torch(n, seed=2, device='cuda', noise_sigma=4.0, cyclic=True)
H=486
out_lines=torch.empty((n*(H+3)+1,48),dtype=torch.uint8,device='cuda'); out_stats=torch.empty((n,32),dtype=torch.uint8,device='cuda')
eng.binarize_frames(luma, first_frame_no=1, new_file=True, out_lines=out_lines, out_stats=out_stats)
p,f=eng.stitch_frames(out_lines)
i=eng.stitch_info(); print('first', p.shape[0], f.shape[0], i.steps, i.rounds, i.steps_launched)
fn=1+n
for it in range(4):
    eng.binarize_frames(luma, first_frame_no=fn, new_file=False, out_lines=out_lines[1:], out_stats=out_stats)
    p,f=eng.stitch_frames(out_lines[1:1+n*(H+3)])
    i=eng.stitch_info(); print('cont', p.shape[0], f.shape[0], i.steps, i.rounds, i.steps_launched)
    fn+=n
