#!/usr/bin/env python3
"""Dev tool: HIP kernel source under the CPU SIMT emulator vs the oracle."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from sdvpcmdecoder_amd import synth
import libs, engine_api
from cmp_v2d import run_lib

ROOT = os.path.join(os.path.dirname(__file__), "..")
emu = engine_api.bind(C.CDLL(os.path.join(ROOT, "tests", "emu", "libsdvpcm_emu.so")))

def cmp(luma, mode, tag):
    ro, so, to = run_lib(libs.load_oracle(), "orc_", luma, mode)
    eng = C.c_void_p(emu.sdv_engine_create(0)); emu.sdv_set_mode(eng, mode)
    t0 = time.perf_counter(); rc, re, se = engine_api.emu_binarize(emu, eng, luma); te = time.perf_counter() - t0
    info = engine_api.RunInfo(); emu.sdv_get_run_info(eng, C.byref(info)); emu.sdv_engine_destroy(eng)
    assert rc == 0, rc
    bad = np.unique(np.nonzero(ro.view(np.uint8).reshape(len(ro), -1) != re.view(np.uint8).reshape(len(re), -1))[0])
    sbad = (so != se.view(np.uint8).reshape(len(se), 32)).any(axis=1).sum()
    ok = (ro["flags"] & 64 != 0).sum()
    print(f"{tag} mode={mode}: recs={len(ro)} crc_ok={ok} rec_mismatch={len(bad)} stats_mismatch={sbad} rounds={info.rounds} launched={info.frames_launched} t_orc={to:.3f}s t_emu={te:.3f}s")
    for i in bad[:3]:
        print("  orc", ro[i]); print("  emu", re[i])
    return len(bad) + sbad

if __name__ == "__main__":
    tot = 0
    modes = [int(x) for x in sys.argv[1:]] or [2, 1]
    for mode in modes:
        luma, _, _ = synth.stc007_frames(3, seed=1, height=64)
        tot += cmp(luma, mode, "clean64")
        luma, _, _ = synth.stc007_frames(2, seed=2, noise_sigma=10.0, blur=2, height=60)
        tot += cmp(luma, mode, "noisy60")
        luma, _, _ = synth.stc007_frames(3, seed=3, height=50, lines_per_field=25, ctrl_block=True)
        tot += cmp(luma, mode, "ctrlblk50")
        luma, _, _ = synth.stc007_frames(2, seed=4, noise_sigma=25.0, blur=3, height=40)
        luma[1, 10:14] = 16
        tot += cmp(luma, mode, "rough40")
    print("TOTAL", tot)
