"""PCM-16x0 frame driver (the PCM-16x0 branch of VideoToDigital::doBinarize): seeded synthetic frames and runners shared by the
oracle-vs-reference test, the golden fixture generator and the product parity tests."""
import ctypes as C

import numpy as np

import libs
from sdvpcmdecoder_amd import synth
from pcm16_front_api import BIN16_DTYPE
from engine_api import STATS_DTYPE

# name: (frames, height, generator kwargs, mode, settings)
CASES = {
    "clean_normal": (3, 48, dict(seed=701), 2, {}),
    "clean_fast": (3, 48, dict(seed=702), 1, {}),
    "clean_draft": (3, 48, dict(seed=703), 0, {}),
    "noisy_normal": (3, 48, dict(seed=704, noise_sigma=10.0, blur=1), 2, {}),
    "noisy_fast": (3, 48, dict(seed=705, noise_sigma=12.0, blur=1), 1, {}),
    "cut_bits_normal": (2, 40, dict(seed=706, x0=-4, x1=723, noise_sigma=3.0), 2, {}),
    "top_blank": (3, 48, dict(seed=707, top_blank=3, noise_sigma=4.0), 2, {}),
    "jitter_draft": (4, 48, dict(seed=708, jitter=2, noise_sigma=4.0), 0, {}),
    "jitter_normal": (3, 48, dict(seed=709, jitter=2, noise_sigma=4.0), 2, {}),
    "dropouts_fast": (4, 48, dict(seed=710, p_dropout=0.08, noise_sigma=5.0), 1, {}),
    "dropouts_draft": (4, 48, dict(seed=711, p_dropout=0.10, noise_sigma=5.0), 0, {}),
    "dup_lines": (3, 48, dict(seed=712, dup_every=7, noise_sigma=3.0), 2, {}),
    "dup_lines_nocheck": (2, 48, dict(seed=713, dup_every=5), 1, dict(check_line_dup=0)),
    "silence": (3, 40, dict(seed=714, silent_from=1), 2, {}),
    "low_contrast": (2, 40, dict(seed=715, black=60, white=95, noise_sigma=2.0), 2, {}),
    "control_bits": (3, 40, dict(seed=716, control="random", noise_sigma=3.0), 1, {}),
    "file_marks": (3, 40, dict(seed=717, noise_sigma=3.0), 2, dict(new_file=True, end_file=True)),
    "forced_coords": (2, 40, dict(seed=718, x0=30, x1=690, noise_sigma=3.0), 2, dict(force=(30, 29))),
    "first_line_dup_off": (2, 40, dict(seed=719), 1, dict(first_line_dup=0)),
    "wide_1440": (2, 32, dict(seed=720, width=1440, x0=8, x1=1432, noise_sigma=3.0), 1, dict(doubled=True)),
    "garbage": (2, 24, dict(seed=721, white=34, noise_sigma=30.0), 1, {}),
    "smeared_parts_normal": (3, 48, dict(seed=722, smear=(3, 250, 330), noise_sigma=3.0), 2, {}),
    "smeared_parts_fast": (3, 48, dict(seed=723, smear=(2, 500, 600), noise_sigma=3.0), 1, {}),
    "smeared_parts_draft": (3, 48, dict(seed=724, smear=(4, 20, 120), noise_sigma=3.0), 0, {}),
    "ntsc_full": (2, 486, dict(seed=725, noise_sigma=4.0), 2, {}),
    # MODE_INSANE: passes that do not read from what was handed on run the reference level sweep
    "insane_smeared": (2, 10, dict(seed=732, smear=(3, 250, 330), black=50, white=100, noise_sigma=3.0), 3, {}),
    "insane_jitter": (2, 16, dict(seed=734, jitter=2, black=40, white=90, noise_sigma=4.0), 3, {}),
    # min_valid_crcs above min_contrast: the Binarizer's sticky sweep flag then decides whether levels 50 apart count as levels
    "insane_flag_matters": (3, 12, dict(seed=736, black=50, white=100, noise_sigma=4.0, smear=(3, 250, 330)), 3, dict(preset=dict(min_valid_crcs=60))),
    "insane_flag_matters_wide": (1, 10, dict(seed=737, black=40, white=100, noise_sigma=4.0, smear=(3, 250, 330)), 3, dict(preset=dict(min_valid_crcs=50))),
}
GOLDEN = ("noisy_normal", "jitter_draft", "dropouts_fast", "dup_lines", "file_marks", "cut_bits_normal", "smeared_parts_normal", "control_bits", "insane_smeared")


def make_input(name):
    n, h, kw, mode, st = CASES[name]
    luma, _ = synth.pcm16x0_frames(n, height=h, **dict(kw))
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
    return n * (3 * h + 3) + (1 if st.get("new_file") else 0) + (h + 4 if st.get("end_file") else 0)


def run_cpu(lib, prefix, luma, mode, st, first_frame_no=1, handle=None, keep=False):
    """orc_v2d16_run / ref_v2d16_run: the frames through the PCM-16x0 worker of the oracle (`orc_`) or of the real reference (`ref_`)."""
    g = lambda name: getattr(lib, prefix + "v2d16_" + name)
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
    recs = np.zeros(nrec, dtype=BIN16_DTYPE)
    stats = np.zeros(n + (1 if st.get("end_file") else 0), dtype=STATS_DTYPE)
    flags = (1 if st.get("new_file") else 0) | (2 if st.get("end_file") else 0)
    got = g("run")(h, luma.ctypes.data, w, w, hh, n, first_frame_no, flags, 1 if st.get("doubled") else 0, recs.ctypes.data, stats.ctypes.data)
    assert got == nrec, (got, nrec)
    if keep:
        return recs, stats, h
    g("delete")(h)
    return recs, stats


def run_engine(lib, eng, luma, mode, st, first_frame_no=1, configure=True):
    """sdv_pcm16x0_binarize_frames on host buffers (the emulator build): one call for all frames."""
    f = lib.sdv_pcm16x0_binarize_frames
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
    recs = np.zeros(n_records(n, h, st), dtype=BIN16_DTYPE)
    stats = np.zeros(n + (1 if st.get("end_file") else 0), dtype=STATS_DTYPE)
    flags = (1 if st.get("new_file") else 0) | (2 if st.get("doubled") else 0) | (4 if st.get("end_file") else 0)
    rc = f(eng, luma.ctypes.data, w, w * h, w, h, n, first_frame_no, flags, recs.ctypes.data, len(recs), stats.ctypes.data, len(stats), None)
    return rc, recs, stats
