"""PCM-1 frame driver (the PCM-1 branch of VideoToDigital::doBinarize): seeded synthetic frames and runners shared by the
oracle-vs-reference test, the golden fixture generator and the product parity tests."""
import ctypes as C

import numpy as np

import libs
from sdvpcmdecoder_amd import synth
from pcm1_front_api import BIN1_DTYPE
from engine_api import STATS_DTYPE

# name: (frames, height, generator kwargs, mode, settings)
CASES = {
    "clean_normal": (3, 64, dict(seed=401), 2, {}),
    "clean_fast": (3, 64, dict(seed=402), 1, {}),
    "clean_draft": (3, 64, dict(seed=403), 0, {}),
    "noisy_normal": (3, 64, dict(seed=404, noise_sigma=14.0, blur=1), 2, {}),
    "noisy_fast": (3, 64, dict(seed=405, noise_sigma=18.0, blur=1), 1, {}),
    "cut_bits_normal": (2, 48, dict(seed=406, x0=-9, x1=726, noise_sigma=3.0), 2, {}),
    "no_header_top_blank": (3, 64, dict(seed=407, header=0, top_blank=3, noise_sigma=4.0), 2, {}),
    "jitter_draft": (4, 64, dict(seed=408, jitter=3, noise_sigma=4.0), 0, {}),
    "jitter_normal": (3, 64, dict(seed=409, jitter=2, noise_sigma=4.0), 2, {}),
    "dropouts_fast": (4, 64, dict(seed=410, p_dropout=0.08, noise_sigma=5.0), 1, {}),
    "dropouts_draft": (4, 64, dict(seed=411, p_dropout=0.10, noise_sigma=5.0), 0, {}),
    "dup_lines": (3, 64, dict(seed=412, dup_every=7, noise_sigma=3.0), 2, {}),
    "dup_lines_nocheck": (2, 64, dict(seed=413, dup_every=5), 1, dict(check_line_dup=0)),
    "silence": (3, 48, dict(seed=414, silent_from=1), 2, {}),
    "low_contrast": (2, 48, dict(seed=415, black=60, white=95, noise_sigma=2.0), 2, {}),
    "narrow_window_fast": (3, 64, dict(seed=416, x0=30, x1=690, noise_sigma=3.0), 1, {}),      # out of the search's reach: no line reads
    "file_marks": (3, 48, dict(seed=417, noise_sigma=3.0), 2, dict(new_file=True, end_file=True)),
    "forced_coords": (2, 48, dict(seed=418, x0=30, x1=690, noise_sigma=3.0), 2, dict(force=(30, 29))),       # offsets from the line ends (binarizer.cpp:631-641)
    "first_line_dup_off": (2, 48, dict(seed=419, header=0), 1, dict(first_line_dup=0)),
    "wide_1440": (2, 32, dict(seed=420, width=1440, x0=8, x1=1432, noise_sigma=3.0), 1, dict(doubled=True)),
    "garbage": (2, 32, dict(seed=421, white=34, noise_sigma=30.0), 1, {}),
    "ntsc_full": (2, 486, dict(seed=422, noise_sigma=4.0), 2, {}),
    # MODE_INSANE: lines that do not read from what was handed on run the reference level sweep
    "insane_noisy": (2, 16, dict(seed=735, noise_sigma=11.0, blur=1, black=50, white=100), 3, {}),
    "insane_jitter": (2, 16, dict(seed=734, jitter=2, black=40, white=90, noise_sigma=4.0), 3, {}),
    # min_valid_crcs above min_contrast: the Binarizer's sticky sweep flag then decides whether levels 50 apart count as levels
    "insane_flag_matters": (3, 16, dict(seed=736, black=50, white=100, noise_sigma=8.0, blur=1, p_dropout=0.1), 3, dict(preset=dict(min_valid_crcs=60))),
    "insane_flag_matters_wide": (1, 12, dict(seed=737, black=40, white=100, noise_sigma=8.0, blur=1), 3, dict(preset=dict(min_valid_crcs=50))),
}
GOLDEN = ("noisy_normal", "jitter_draft", "dropouts_fast", "dup_lines", "file_marks", "cut_bits_normal", "insane_noisy")


def make_input(name):
    n, h, kw, mode, st = CASES[name]
    kw = dict(kw)
    luma, _ = synth.pcm1_frames(n, height=h, **kw)
    return luma, mode, st


def _preset(st):
    p = libs.default_preset()
    if "force" in st:
        p.en_force_coords = 1
        p.horiz_start, p.horiz_stop = st["force"]
    if "first_line_dup" in st:
        p.en_first_line_dup = st["first_line_dup"]
    for k, v in st.get("preset", {}).items():
        setattr(p, k, v)
    return p


def n_records(n, h, st):
    return n * (h + 3) + (1 if st.get("new_file") else 0) + (h + 4 if st.get("end_file") else 0)


def run_cpu(lib, prefix, luma, mode, st, first_frame_no=1, handle=None, keep=False):
    """orc_v2d1_run / ref_v2d1_run: the frames through the PCM-1 worker of the oracle (`orc_`) or of the real reference (`ref_`)."""
    g = lambda name: getattr(lib, prefix + "v2d1_" + name)
    g("new").restype = C.c_void_p
    for nm in ("delete", "set_mode", "set_check_line_dup", "set_preset"):
        g(nm).restype = None
    g("delete").argtypes = [C.c_void_p]
    g("set_mode").argtypes = [C.c_void_p, C.c_int]
    g("set_check_line_dup").argtypes = [C.c_void_p, C.c_int]
    g("set_preset").argtypes = [C.c_void_p, C.POINTER(libs.BinPreset)]
    g("run").restype = C.c_long
    g("run").argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    h = handle or C.c_void_p(g("new")())
    if handle is None:
        if "force" in st or "first_line_dup" in st or "preset" in st:
            g("set_preset")(h, C.byref(_preset(st)))
        g("set_mode")(h, mode)
        if "check_line_dup" in st:
            g("set_check_line_dup")(h, st["check_line_dup"])
    luma = np.ascontiguousarray(luma)
    n, hh, w = luma.shape
    nrec = n_records(n, hh, st)
    recs = np.zeros(nrec, dtype=BIN1_DTYPE)
    stats = np.zeros(n + (1 if st.get("end_file") else 0), dtype=STATS_DTYPE)
    flags = (1 if st.get("new_file") else 0) | (2 if st.get("end_file") else 0)
    got = g("run")(h, luma.ctypes.data, w, w, hh, n, first_frame_no, flags, 1 if st.get("doubled") else 0, recs.ctypes.data, stats.ctypes.data)
    assert got == nrec, (got, nrec)
    if keep:
        return recs, stats, h
    g("delete")(h)
    return recs, stats


def run_engine(lib, eng, luma, mode, st, first_frame_no=1, configure=True):
    """sdv_pcm1_binarize_frames on host buffers (the emulator build): one call for all frames."""
    f = lib.sdv_pcm1_binarize_frames
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint, C.c_void_p, C.c_size_t,
                  C.c_void_p, C.c_size_t, C.c_void_p]
    if configure:
        lib.sdv_set_mode.argtypes = [C.c_void_p, C.c_int]
        lib.sdv_set_mode(eng, mode)
        lib.sdv_set_bin_preset.argtypes = [C.c_void_p, C.POINTER(libs.BinPreset)]
        lib.sdv_set_bin_preset(eng, C.byref(_preset(st)))
        lib.sdv_set_check_line_dup.argtypes = [C.c_void_p, C.c_int]
        lib.sdv_set_check_line_dup(eng, st.get("check_line_dup", 1))
    luma = np.ascontiguousarray(luma)
    n, h, w = luma.shape
    recs = np.zeros(n_records(n, h, st), dtype=BIN1_DTYPE)
    stats = np.zeros(n + (1 if st.get("end_file") else 0), dtype=STATS_DTYPE)
    flags = (1 if st.get("new_file") else 0) | (2 if st.get("doubled") else 0) | (4 if st.get("end_file") else 0)
    rc = f(eng, luma.ctypes.data, w, w * h, w, h, n, first_frame_no, flags, recs.ctypes.data, len(recs), stats.ctypes.data, len(stats), None)
    return rc, recs, stats
