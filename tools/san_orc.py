import sys, os
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes as C
import libs
libs.load_oracle = lambda: _lib
_lib = C.CDLL(os.path.join(ROOT, 'build', 'san') + '/liborc.so')
# bind like libs does
orig = libs.__dict__
try:
    libs._bind_bin_api(_lib, "orc_")
except Exception as e:
    print("bind", e)
import numpy as np
import pcm1_api as p1, stitch_api as sa, stitch_cases as sc
from oracle_run import oracle_binarize
for name in list(p1.CASES):
    recs, st = p1.make_input(name)
    p1.run_cpu(_lib, "orc_", recs, st)
print("pcm1 ok")
for name in ("ntsc_bad5", "ntsc_burst300", "pal_bad5", "f1_16bit_bad5", "ntsc_ctrlblk", "ntsc_drift", "ntsc_noisy_video"):
    recs, st = sc.make_input(name, lambda luma: oracle_binarize(luma, mode=2))
    sa.run_cpu(_lib, "orc_", recs, st)
print("stitch ok")

import pcm1_front_api as pf
for name in sorted(pf.CASES):
    luma, run = pf.make_case(name)
    pf.run_lines(_lib, "orc_bin1_", luma, **run)
print("pcm1 front ok")

# round 2: PCM-16x0 lines, both frame drivers (MODE_INSANE included), the PCM-16x0 back half
import pcm16_front_api as p16f
for name in sorted(p16f.CASES):
    luma, run = p16f.make_case(name)
    p16f.run_lines(_lib, "orc_bin16_", luma, **run)
print("pcm16x0 front ok")
import pcm1_frames_api as f1, pcm16_frames_api as f16
for pf in (f1, f16):
    for name in sorted(pf.CASES):
        if name == "ntsc_full":
            continue
        luma, mode, st = pf.make_input(name)
        pf.run_cpu(_lib, "orc_", luma, mode, st)
print("frame drivers ok")
import pcm16_api as p16
for name in sorted(p16.CASES):
    recs, st = p16.make_input(name)
    p16.run_cpu(_lib, "orc_", recs, st)
print("pcm16x0 stitch ok")
# round 2, SURVEY 8f: AudioProcessor / SamplesToWAV
import audio_api as au
for name in sorted(au.CASES):
    pairs, mode, ends, stop = au.make_input(name)
    out, idx, pur, masked, hit = au.run_cpu(_lib, "orc_", pairs, mode, ends, stop)
    au.wav_files(_lib, "orc_", out, pur)
print("audio ok")
