import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes as C, numpy as np
import libs, engine_api as ea, pcm1_api as p1, stitch_api as sa, stitch_cases as sc
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import synth
lib = ea.bind(C.CDLL(os.path.join(ROOT, 'build', 'san') + '/libsdvpcm_emu.so'))
orc = libs.load_oracle()
for name in list(p1.CASES):
    recs, st = p1.make_input(name)
    wp, wf = p1.run_cpu(orc, "orc_", recs, st)
    eng = lib.sdv_engine_create(0)
    rc, p, f = ea.emu_pcm1_stitch(lib, eng, recs, st)
    assert rc == 0 and p.tobytes() == wp.tobytes() and f.tobytes() == wf.tobytes(), name
    lib.sdv_engine_destroy(eng)
print("pcm1 emu ok")
for name in ("ntsc_bad5", "ntsc_burst300", "f1_16bit_bad5", "ntsc_drift"):
    recs, st = sc.make_input(name, lambda luma: oracle_binarize(luma, mode=2))
    wp, wf = sa.run_cpu(orc, "orc_", recs, st)
    eng = lib.sdv_engine_create(0)
    rc, p, f = ea.emu_stitch(lib, eng, recs, st)
    assert rc == 0 and p.tobytes() == wp.tobytes() and f.tobytes() == wf.tobytes(), name
    lib.sdv_engine_destroy(eng)
print("stitch emu ok")
luma, _, _ = synth.stc007_frames(3, seed=21, height=96, noise_sigma=8.0, blur=1)
luma = luma.copy(); luma[:, 33::17] = 16
want, wst = oracle_binarize(luma, mode=2, first_frame_no=1, new_file=True)
eng = lib.sdv_engine_create(0)
lib.sdv_set_mode(eng, 2)
rc, recs, stats = ea.emu_binarize(lib, eng, luma)
assert rc == 0 and recs.tobytes() == want.tobytes()
print("binarize emu ok")
