"""PCM-1 back half (PCM1DataStitcher): PODs, seeded record streams and runners shared by the oracle-vs-reference test,
the golden fixture generator (tests/golden/make_golden_pcm1.py) and the product parity tests."""
import ctypes as C
import hashlib

import numpy as np

import libs
from sdvpcmdecoder_amd import synth
from stitch_api import PAIR_DTYPE

LINE1_DTYPE = synth.PCM1_LINE_DTYPE
FRASM1_DTYPE = np.dtype([("frame_number", "<u4"),
                         ("odd_std_lines", "<u2"), ("even_std_lines", "<u2"), ("odd_data_lines", "<u2"), ("even_data_lines", "<u2"),
                         ("odd_valid_lines", "<u2"), ("even_valid_lines", "<u2"),
                         ("odd_top_data", "<u2"), ("odd_bottom_data", "<u2"), ("even_top_data", "<u2"), ("even_bottom_data", "<u2"),
                         ("odd_sample_rate", "<u2"), ("even_sample_rate", "<u2"),
                         ("blocks_total", "<u2"), ("blocks_drop", "<u2"), ("samples_drop", "<u2"),
                         ("odd_top_padding", "<u2"), ("odd_bottom_padding", "<u2"), ("even_top_padding", "<u2"), ("even_bottom_padding", "<u2"),
                         ("blocks_fix_bp", "<u2"),
                         ("field_order", "u1"), ("odd_ref", "u1"), ("even_ref", "u1"), ("service_type", "u1"), ("flags", "u1"), ("_pad", "u1", (3,))])
assert LINE1_DTYPE.itemsize == 32 and FRASM1_DTYPE.itemsize == 52, (LINE1_DTYPE.itemsize, FRASM1_DTYPE.itemsize)

SRV_NEW_FILE, SRV_END_FILE, SRV_FILLER, SRV_END_FIELD, SRV_END_FRAME, SRV_HEADER = 1, 2, 3, 4, 5, 6
LF_BW_SET, LF_FORCED_BAD = 8, 32


class Pcm1Settings(C.Structure):
    _fields_ = [("field_order", C.c_uint8), ("auto_offset", C.c_uint8), ("use_ecc", C.c_uint8), ("odd_offset", C.c_int8),
                ("even_offset", C.c_int8), ("_pad", C.c_uint8 * 3)]


assert C.sizeof(Pcm1Settings) == 8


def default_settings(**kw):
    st = Pcm1Settings(1, 1, 1, 0, 0)
    for k, v in kw.items():
        setattr(st, k, v)
    return st


crc_words = synth.pcm1_crc_words
make_stream = synth.pcm1_line_stream


# name: (frames, generator kwargs, settings overrides)
CASES = {
    "clean": (3, dict(seed=301), {}),
    "header_emph": (3, dict(seed=302, header=4, footer=3), {}),
    "bad5": (4, dict(seed=303, p_bad=0.05), {}),
    "bad30_noecc": (3, dict(seed=304, p_bad=0.30, p_nobw=0.1), dict(use_ecc=0)),
    "bff": (3, dict(seed=305, p_bad=0.03), dict(field_order=2)),
    "short_fields": (3, dict(seed=306, lines=(240, 238), p_bad=0.02), {}),
    "short_fields_header": (3, dict(seed=307, lines=(236, 241), header=2, p_bad=0.02), {}),
    "long_fields": (3, dict(seed=308, lines=(250, 247), p_bad=0.02), {}),
    "noise_outside": (4, dict(seed=309, noise_lines=6, p_bad=0.25), {}),
    "picked_forced": (3, dict(seed=310, p_bad=0.02, p_picked=0.08, p_forced=0.03), {}),
    "fillers": (3, dict(seed=311, p_bad=0.02, p_filler=0.03), {}),
    "manual_offsets": (3, dict(seed=312, p_bad=0.02, first=(5, 8)), dict(auto_offset=0, odd_offset=2, even_offset=-3)),
    "manual_offsets_big": (3, dict(seed=313, p_bad=0.02, lines=(250, 250)), dict(auto_offset=0, odd_offset=-4, even_offset=5)),
    "file_marks": (4, dict(seed=314, p_bad=0.03, header=3, new_file=True, end_file=True), {}),
    "empty_frames": (4, dict(seed=315, p_bad=0.02, empty=(1,), one_field=(2,)), {}),
    "burst": (4, dict(seed=316, burst=(300, 200)), {}),
    "very_long_fields": (3, dict(seed=317, lines=(345, 350), p_bad=0.02, header=2), {}),      # > 672 records: the kernel's global-memory path
    # what the reference's per-frame queue walk (pcm1datastitcher.cpp:1392-1473) drops or cuts: see MANGLE
    "stale_new_file": (4, dict(seed=318, p_bad=0.03, header=3, new_file=True, end_file=True), {}),
    "stale_end_file": (4, dict(seed=319, p_bad=0.03, new_file=True, end_file=True), {}),
    "stale_tag_inside": (4, dict(seed=320, p_bad=0.03, header=2), {}),
    # manual line offsets over fields that are longer than 245 lines by their numbers and have lost lines: the stitcher is told to read its field
    # buffers past what the frame wrote and puts out what earlier frames left there (pcm1datastitcher.cpp:896-909, :952-1016)
    "manual_lost_lines": (6, dict(seed=323, p_bad=0.03, lines=(252, 250)), dict(auto_offset=0, odd_offset=-2, even_offset=1)),
    "manual_lost_many": (8, dict(seed=324, p_bad=0.05, lines=(258, 261), header=2), dict(auto_offset=0, odd_offset=3, even_offset=-5)),
    "manual_lost_first_frame": (5, dict(seed=325, p_bad=0.02, lines=(250, 255)), dict(auto_offset=0, odd_offset=0, even_offset=0)),
    "manual_lost_file_marks": (7, dict(seed=326, p_bad=0.03, lines=(251, 249), new_file=True, end_file=True), dict(auto_offset=0, odd_offset=-1, even_offset=2, use_ecc=0)),
    "manual_lost_long_frame": (6, dict(seed=327, p_bad=0.03, lines=(300, 296), header=1), dict(auto_offset=0, odd_offset=-3, even_offset=2)),     # > 544 records: the global-memory path
    # manual line offsets at the ends of their range (setOddLineOffset / setEvenLineOffset take an int8) over fields that hold far fewer than 245 lines,
    # some of them lost: the stitcher pads 128 lines above a field / skips 127 lines of it and still hands PCM1Deinterleaver a queue of exactly one
    # field (pcm1datastitcher.cpp:809-923) - its DI_RET_NO_DATA (pcm1deinterleaver.cpp:104, 119) cannot be reached through the stitcher
    "manual_short_fields": (5, dict(seed=331, p_bad=0.03, lines=(120, 96)), dict(auto_offset=0, odd_offset=-128, even_offset=127)),
    "manual_short_fields_lost": (6, dict(seed=332, p_bad=0.04, lines=(131, 245), header=1), dict(auto_offset=0, odd_offset=127, even_offset=-128)),
    "overlong_frame": (3, dict(seed=321, lines=(990, 985), p_bad=0.02), {}),
    "overlong_frame_tagged": (3, dict(seed=322, lines=(990, 985), p_bad=0.02, new_file=True, end_file=True), {}),
}
GOLDEN = ("header_emph", "bad5", "noise_outside", "file_marks", "manual_offsets", "empty_frames", "stale_tag_inside", "overlong_frame_tagged",
          "manual_lost_many", "manual_lost_first_frame", "manual_short_fields", "manual_short_fields_lost")


def _older_number(recs, srv, which=0):
    at = np.nonzero(recs["service_type"] == srv)[0][which]
    recs["frame_number"][at] -= 1
    return recs


def _tag_inside(recs):
    """A NEW_FILE and an END_FILE tag of the frame before in the middle of frame 2, and a NEW_FILE of its own in frame 3."""
    ends = np.nonzero(recs["service_type"] == SRV_END_FRAME)[0]
    out = []
    for at, srv, d in ((ends[0] + 40, SRV_NEW_FILE, -1), (ends[0] + 300, SRV_END_FILE, -1), (ends[1] + 120, SRV_NEW_FILE, 0)):
        r = recs[at:at + 1].copy()
        r["service_type"] = srv
        r["frame_number"] = int(recs["frame_number"][at]) + d
        out.append((at, r))
    parts, last = [], 0
    for at, r in out:
        parts += [recs[last:at], r]
        last = at
    parts.append(recs[last:])
    return np.concatenate(parts)


def _lose_lines(share, seed, first_frame_share=None):
    """Data lines lost on the way (the records are simply not there); first_frame_share: another share for the first frame."""
    def f(recs):
        rng = np.random.default_rng(seed)
        p = np.full(len(recs), share)
        if first_frame_share is not None:
            p[recs["frame_number"] == recs["frame_number"][0]] = first_frame_share
        return recs[~((recs["service_type"] == 0) & (rng.random(len(recs)) < p))]
    return f


# name: what is done to the generated stream
MANGLE = {
    "manual_lost_lines": _lose_lines(0.04, 1),
    "manual_lost_many": _lose_lines(0.15, 2),
    "manual_lost_first_frame": _lose_lines(0.03, 3, first_frame_share=0.5),      # places no frame has written yet: default-constructed sub-lines
    "manual_lost_file_marks": _lose_lines(0.08, 4),
    "manual_lost_long_frame": _lose_lines(0.06, 5),
    "manual_short_fields_lost": _lose_lines(0.10, 6),
    "stale_new_file": lambda r: _older_number(r, SRV_NEW_FILE),
    "stale_end_file": lambda r: _older_number(r, SRV_END_FILE),
    "stale_tag_inside": _tag_inside,
}


def make_input(name):
    n, kw, st_kw = CASES[name]
    recs = make_stream(n, **kw)
    if name in MANGLE:
        recs = MANGLE[name](recs.copy())
    return recs, default_settings(**st_kw)


def run_cpu(lib, prefix, recs, st, pair_cap=None, frame_cap=None, overflow_ok=False):
    f = getattr(lib, prefix + "pcm1_stitch_run")
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Pcm1Settings), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    recs = np.ascontiguousarray(recs)
    nfr = int((recs["service_type"] == SRV_END_FRAME).sum()) + 2
    pair_cap = pair_cap or nfr * 1472 + 16
    frame_cap = frame_cap or nfr + 8
    pairs = np.zeros(pair_cap, dtype=PAIR_DTYPE)
    frames = np.zeros(frame_cap, dtype=FRASM1_DTYPE)
    nf = C.c_size_t(0)
    n = f(recs.ctypes.data, len(recs), C.byref(st), pairs.ctypes.data, pair_cap, frames.ctypes.data, frame_cap, C.byref(nf))
    if n < 0 and overflow_ok:
        return None, nf.value
    assert n >= 0, "pair buffer too small"
    return pairs[:n], frames[:min(nf.value, frame_cap)]


# ---- what the stitcher hands to the visualiser (sdv_set_pcm1_stitch_block_output / _line_output) ----------------------------------------------
BLOCK1_DTYPE = np.dtype([("frame_number", "<u4"), ("start_line", "<u2"), ("stop_line", "<u2"), ("interleave_num", "u1"), ("flags", "u1"), ("sample_rate", "<u2"),
                         ("words", "<u2", (184,)), ("word_flags", "u1", (184,)), ("_pad", "u1", (12,))])
ASM1_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (2,)), ("picked_bits_left", "u1"), ("picked_bits_right", "u1"),
                       ("line_part", "u1"), ("flags", "u1"), ("_pad", "u1", (2,))])
assert BLOCK1_DTYPE.itemsize == 576 and ASM1_DTYPE.itemsize == 16
P1S_SKIP = 0x80


def run_cpu_vis(lib, prefix, recs, st):
    """(pairs, frames, blocks, sub-lines): the stitcher's run with the two feeds of the visualiser switched on."""
    f = getattr(lib, prefix + "pcm1_stitch_run_vis")
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Pcm1Settings), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    recs = np.ascontiguousarray(recs)
    nfr = int((recs["service_type"] == SRV_END_FRAME).sum()) + 2
    pairs = np.zeros(nfr * 1472 + 16, dtype=PAIR_DTYPE)
    frames = np.zeros(nfr + 8, dtype=FRASM1_DTYPE)
    blocks = np.zeros(nfr * 16, dtype=BLOCK1_DTYPE)
    lines = np.zeros(nfr * 1470, dtype=ASM1_DTYPE)
    nf, nb, nl = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    n = f(recs.ctypes.data, len(recs), C.byref(st), pairs.ctypes.data, len(pairs), frames.ctypes.data, len(frames), C.byref(nf),
          blocks.ctypes.data, len(blocks), C.byref(nb), lines.ctypes.data, len(lines), C.byref(nl))
    assert n >= 0 and nb.value <= len(blocks) and nl.value <= len(lines)
    return pairs[:n], frames[:nf.value], blocks[:nb.value], lines[:nl.value]


VIS_GOLDEN = ("bad5", "picked_forced", "header_emph", "manual_lost_lines", "bff")


def comparable_blocks(blocks, stale_frames=False):
    """Blocks with what the reference leaves undefined taken out: the stop line of a field's last block (read one sub-line past the end of the queue)."""
    b = blocks.copy()
    b["stop_line"][b["interleave_num"] == 7] = 0
    if stale_frames:
        b["frame_number"] = 0
    return b


def digest(pairs, frames):
    return hashlib.sha256(pairs.tobytes() + frames.tobytes()).hexdigest()
