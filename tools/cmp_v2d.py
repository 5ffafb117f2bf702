#!/usr/bin/env python3
"""Dev tool: oracle vs real reference at the VideoToDigital level on synthetic frames."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from sdvpcmdecoder_amd import synth
import libs

def run_lib(lib, prefix, luma, mode, first=1, new_file=1):
    n, h, w = luma.shape
    new = getattr(lib, prefix + "v2d_new"); new.restype = C.c_void_p
    hnd = C.c_void_p(new())
    getattr(lib, prefix + "v2d_set_mode")(hnd, mode)
    run = getattr(lib, prefix + "v2d_run"); run.restype = C.c_long
    run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    recs = np.zeros(n * (h + 3) + new_file, dtype=libs.LINE_DTYPE)
    stats = np.zeros((n, 32), dtype=np.uint8)
    t0 = time.perf_counter()
    got = run(hnd, luma.ctypes.data, w, w, h, n, first, new_file, 0, recs.ctypes.data, stats.ctypes.data)
    dt = time.perf_counter() - t0
    dele = getattr(lib, prefix + ("v2d_delete")); dele.argtypes = [C.c_void_p]; dele(hnd)
    assert got == len(recs)
    return recs, stats, dt

def cmp(luma, mode, tag):
    ro, so, to = run_lib(libs.load_oracle(), "orc_", luma, mode)
    rr, sr, tr = run_lib(libs.load_ref(), "ref_", luma, mode)
    bad = np.nonzero(ro.view(np.uint8).reshape(len(ro), -1) != rr.view(np.uint8).reshape(len(rr), -1))[0]
    bad = np.unique(bad)
    sbad = (so != sr).any(axis=1).sum()
    ok = (rr["flags"] & 64 != 0).sum()
    print(f"{tag} mode={mode}: recs={len(ro)} crc_ok={ok} rec_mismatch={len(bad)} stats_mismatch={sbad} t_orc={to:.3f}s t_ref={tr:.3f}s")
    for i in bad[:3]:
        print("  orc", ro[i]); print("  ref", rr[i])
    return len(bad) + sbad

if __name__ == "__main__":
    tot = 0
    for mode in (2, 1, 0, 3):
        luma, _, _ = synth.stc007_frames(4, seed=1)
        tot += cmp(luma, mode, "clean")
        luma, _, _ = synth.stc007_frames(3, seed=2, noise_sigma=10.0, blur=2)
        tot += cmp(luma, mode, "noisy")
        luma, _, _ = synth.stc007_frames(3, seed=3, height=490, ctrl_block=True)
        tot += cmp(luma, mode, "ctrlblk490")
        luma, _, _ = synth.stc007_frames(3, seed=4, noise_sigma=25.0, blur=3)
        luma[1, 100:140] = 16
        luma[2, ::7] = luma[2, 1::7]
        tot += cmp(luma, mode, "rough")
    print("TOTAL", tot)
