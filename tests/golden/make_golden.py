#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by running the REAL reference
(oracle/_ref/libsdvref.so = Fagear/SDVPCMdecoder sources compiled by oracle/Makefile.ref) on seeded
synthetic frames.  Only runs in the build container (needs /root/reference); the .npz outputs are
committed and are what travels to the GPU box.

Each fixture: inputs (generator arguments, so the luma is regenerated bit-identically by
sdvpcmdecoder_amd.synth) + the expected STC007Line records and FrameBinDescriptor rows produced by the
reference's VideoToDigital worker loop."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs  # noqa: E402
from sdvpcmdecoder_amd import synth  # noqa: E402

CASES = {
    # name: (mode, generator kwargs, post-edit)
    "ntsc_clean_normal": (2, dict(n_frames=3, seed=101), None),
    "ntsc_noisy_normal": (2, dict(n_frames=2, seed=102, noise_sigma=10.0, blur=2), None),
    "ntsc_ctrlblk_fast": (1, dict(n_frames=2, seed=103, height=490, ctrl_block=True), None),
    "pal_clean_normal": (2, dict(n_frames=2, seed=104, height=576, lines_per_field=294), None),
    "ntsc_dropouts_normal": (2, dict(n_frames=2, seed=105, noise_sigma=6.0, blur=1), "dropouts"),
    "ntsc_rough_insane": (3, dict(n_frames=1, seed=106, noise_sigma=22.0, blur=3, height=120), None),
    "ntsc_rough_draft": (0, dict(n_frames=2, seed=107, noise_sigma=22.0, blur=3, height=120), None),
    "ntsc_silent_normal": (2, dict(n_frames=2, seed=108, silent=True, height=120), None),
}


def make_luma(kw, edit):
    luma, _, _ = synth.stc007_frames(**kw)
    if edit == "dropouts":
        luma = luma.copy()
        luma[:, 97::97] = 16                      # whole-line dropouts
        luma[1, 200:204, 300:420] = 235           # a white scratch
        luma[0, 50] = luma[0, 48]                 # duplicated line (VTR dropout compensator)
    return luma


def run_ref(luma, mode, first=1, new_file=1, end_file=0):
    lib = libs.load_ref()
    lib.ref_v2d_new.restype = C.c_void_p
    lib.ref_v2d_run.restype = C.c_long
    lib.ref_v2d_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p]
    lib.ref_v2d_delete.argtypes = [C.c_void_p]
    lib.ref_v2d_set_mode.argtypes = [C.c_void_p, C.c_int]
    h = C.c_void_p(lib.ref_v2d_new())
    lib.ref_v2d_set_mode(h, mode)
    n, hh, w = luma.shape
    recs = np.zeros(n * (hh + 3) + new_file + (hh + 4 if end_file else 0), dtype=libs.LINE_DTYPE)
    stats = np.zeros((n + (1 if end_file else 0), 32), dtype=np.uint8)
    got = lib.ref_v2d_run(h, luma.ctypes.data, w, w, hh, n, first, new_file | (2 if end_file else 0), 0, recs.ctypes.data, stats.ctypes.data)
    lib.ref_v2d_delete(h)
    assert got == len(recs)
    return recs, stats


if __name__ == "__main__":
    for name, (mode, kw, edit) in CASES.items():
        luma = make_luma(kw, edit)
        recs, stats = run_ref(np.ascontiguousarray(luma), mode)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), mode=mode, recs=recs.view(np.uint8).reshape(len(recs), 48),
                            stats=stats)
        ok = int(((recs["flags"] & 64) != 0).sum())
        print(f"{name}: {len(recs)} records, {ok} with valid CRC, {os.path.getsize(os.path.join(HERE, name + '.npz'))} bytes")
