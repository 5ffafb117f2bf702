"""Do the frame kernel (HBM-bound) and the stitch kernels (latency / issue-bound) overlap when they run on two streams?  Two engines, two host
threads: engine A binarizes a 10 000-frame NTSC tape over and over, engine B stitches the records of such a tape over and over.
Prints ms per call of each alone and of both when they run at the same time.   usage: overlap_probe.py [frames] [calls]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = 720, 486
luma, _ = synth.stc007_frames_torch(n, seed=2, device="cuda", width=W, height=H, noise_sigma=4.0, cyclic=True)
ea, eb = Engine(0), Engine(0)
for e in (ea, eb):
    e.setBinarizationMode(2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
# records for the stitcher: a continuing tape, decoded once
fno = [1]
def bin_call(e, s, new_file=False):
    out = e.binarize_frames(luma, first_frame_no=fno[0], new_file=new_file, stream=s)
    fno[0] += n
    return out
lines0, _ = bin_call(eb, sb, True)
lines1, _ = bin_call(eb, sb)
lines1 = lines1.clone()
torch.cuda.synchronize()
bin_call(ea, sa, True); bin_call(ea, sa)
# stitcher warm-up: the same records again and again would repeat frame numbers - renumber per call on the device
def stitch_call(k):
    recs = lines1.clone()
    v = recs.view(-1, 48)[:, 0:4].contiguous().view(torch.int32)
    v += k * n
    recs.view(-1, 48)[:, 0:4] = v.view(torch.uint8).view(-1, 4)
    return recs
pre = [stitch_call(k) for k in range(1, 2 * calls + 4)]
torch.cuda.synchronize()
eb.stitch_frames(lines0, stream=sb); 
kk = [0]
def stitch_once():
    eb.stitch_frames(pre[kk[0]], stream=sb); kk[0] += 1
stitch_once(); stitch_once()
torch.cuda.synchronize()
def loop_bin(res):
    t0 = time.perf_counter()
    for _ in range(calls): bin_call(ea, sa)
    sa.synchronize(); res.append((time.perf_counter() - t0) / calls * 1e3)
def loop_st(res):
    t0 = time.perf_counter()
    for _ in range(calls): stitch_once()
    sb.synchronize(); res.append((time.perf_counter() - t0) / calls * 1e3)
ra, rb = [], []
loop_bin(ra); loop_st(rb)
print("alone:    binarize %.3f ms per call, stitch %.3f ms per call (sum %.3f)" % (ra[0], rb[0], ra[0] + rb[0]))
ra2, rb2 = [], []
ta = threading.Thread(target=loop_bin, args=(ra2,)); tb = threading.Thread(target=loop_st, args=(rb2,))
t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); wall = (time.perf_counter() - t0) / calls * 1e3
print("together: binarize %.3f ms per call, stitch %.3f ms per call, wall %.3f ms per pair of calls" % (ra2[0], rb2[0], wall))
print("stitch info:", eb.stitch_info().rounds, "rounds")
