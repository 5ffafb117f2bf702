"""PCM-16x0 back half (PCM16X0Deinterleaver, PCM16X0DataStitcher): PODs, seeded sub-line streams and runners shared by the
oracle-vs-reference test, the golden fixture generator (tests/golden/make_golden_pcm16.py) and the product parity tests."""
import ctypes as C
import hashlib

import numpy as np

import libs
from sdvpcmdecoder_amd import synth
from stitch_api import PAIR_DTYPE

SUB_DTYPE = synth.PCM16X0_BIN_DTYPE
FRASM16_DTYPE = np.dtype([("frame_number", "<u4"),
                          ("odd_std_lines", "<u2"), ("even_std_lines", "<u2"), ("odd_data_lines", "<u2"), ("even_data_lines", "<u2"),
                          ("odd_valid_lines", "<u2"), ("even_valid_lines", "<u2"),
                          ("odd_top_data", "<u2"), ("odd_bottom_data", "<u2"), ("even_top_data", "<u2"), ("even_bottom_data", "<u2"),
                          ("odd_sample_rate", "<u2"), ("even_sample_rate", "<u2"),
                          ("blocks_total", "<u2"), ("blocks_drop", "<u2"), ("samples_drop", "<u2"),
                          ("odd_top_padding", "<u2"), ("odd_bottom_padding", "<u2"), ("even_top_padding", "<u2"), ("even_bottom_padding", "<u2"),
                          ("blocks_broken", "<u2"), ("blocks_fix_bp", "<u2"), ("blocks_fix_p", "<u2"), ("blocks_fix_cwd", "<u2"),
                          ("field_order", "u1"), ("odd_ref", "u1"), ("even_ref", "u1"), ("service_type", "u1"), ("flags", "u1"), ("_pad", "u1")])
BLOCK16_DTYPE = np.dtype([("frame_number", "<u4"), ("start_line", "<u2"), ("stop_line", "<u2"), ("queue_order", "<u2"),
                          ("start_part", "u1"), ("stop_part", "u1"), ("words", "<u2", (3, 3)), ("word_crc", "u1", (3, 3)), ("word_valid", "u1", (3, 3)),
                          ("picked_left", "u1", (3,)), ("picked_crc", "u1", (3,)), ("audio_state", "u1", (3,)), ("order_even", "u1"), ("ret", "u1"), ("_pad", "u1")])
assert SUB_DTYPE.itemsize == 36 and FRASM16_DTYPE.itemsize == 56 and BLOCK16_DTYPE.itemsize == 60, (FRASM16_DTYPE.itemsize, BLOCK16_DTYPE.itemsize)

SRV_NEW_FILE, SRV_END_FILE, SRV_FILLER, SRV_END_FIELD, SRV_END_FRAME = 1, 2, 3, 4, 5
FORMAT_SI, FORMAT_EI = 1, 2
FA16_SILENCE, FA16_PADDING_OK, FA16_EI = 16, 32, 64


class Pcm16Settings(C.Structure):
    _fields_ = [("format", C.c_uint8), ("field_order", C.c_uint8), ("p_correction", C.c_uint8), ("use_ecc", C.c_uint8),
                ("mask_seams", C.c_uint8), ("broke_mask", C.c_uint8), ("sample_rate_preset", C.c_uint16)]


assert C.sizeof(Pcm16Settings) == 8


def default_settings(**kw):
    st = Pcm16Settings(FORMAT_SI, 1, 1, 1, 1, 81, 1)
    for k, v in kw.items():
        setattr(st, k, v)
    return st


make_stream = synth.pcm16x0_sub_stream

# name: (frames, generator kwargs, settings overrides)
CASES = {
    "si_clean": (4, dict(seed=401), {}),
    "si_cut_top": (5, dict(seed=402, cut=(6, 9), p_bad=0.01), {}),
    "si_cut_both": (5, dict(seed=403, cut=(12, 3), tail_cut=(4, 10), p_bad=0.02), {}),
    "si_rate_emph": (5, dict(seed=404, cut=(5, 5), rate_44100=True, emphasis=True, p_bad=0.02), {}),
    "si_code": (4, dict(seed=405, cut=(3, 8), code=True, rate_44100=True), {}),
    "si_bad10": (5, dict(seed=406, cut=(7, 7), p_bad=0.10, rate_44100=True), {}),
    "si_bad35": (4, dict(seed=407, cut=(2, 4), p_bad=0.35), {}),
    "si_noise_rows": (5, dict(seed=408, cut=(8, 6), lead=(3, 2), trail=(2, 4), p_bad=0.03, rate_44100=True), {}),
    "si_picked_forced": (5, dict(seed=409, cut=(4, 4), p_bad=0.04, p_picked=0.10, p_forced=0.03, rate_44100=True), {}),
    "si_silence": (6, dict(seed=410, cut=(6, 6), silent=(1, 2), quiet=(4,), p_bad=0.01), {}),
    "si_wander": (8, dict(seed=411, cut=(4, 6), wander=(2, 5), p_bad=0.02, rate_44100=True), {}),
    "si_burst": (5, dict(seed=412, cut=(5, 5), burst=(2, 0, 60, 50), rate_44100=True), {}),
    "si_bff": (4, dict(seed=413, cut=(5, 7), bff=True, p_bad=0.02), dict(field_order=2)),
    "si_no_p": (4, dict(seed=414, cut=(5, 7), p_bad=0.05), dict(p_correction=0)),
    "si_no_ecc": (4, dict(seed=415, cut=(5, 7), p_bad=0.05, p_nobw=0.05), dict(use_ecc=0)),
    "si_no_mask": (4, dict(seed=416, cut=(9, 2), p_bad=0.30), dict(mask_seams=0, broke_mask=0)),
    "si_rate_preset": (3, dict(seed=417, cut=(5, 5), rate_44100=True), dict(sample_rate_preset=44056)),
    "si_file_marks": (5, dict(seed=418, cut=(5, 7), p_bad=0.03, new_file=True, end_file=True, rate_44100=True), {}),
    "si_empty_frames": (6, dict(seed=419, cut=(5, 7), p_bad=0.02, empty=(1,), one_field=(3,)), {}),
    "si_short_fields": (4, dict(seed=420, cut=(100, 150), tail_cut=(120, 70), p_bad=0.02), {}),
    "si_long_lead": (4, dict(seed=421, cut=(0, 0), lead=(20, 25), p_bad=0.02), {}),
    "ei_clean": (4, dict(seed=431, ei=True), dict(format=FORMAT_EI)),
    "ei_cut": (5, dict(seed=432, ei=True, cut=(6, 9), tail_cut=(3, 2), p_bad=0.01, rate_44100=True), dict(format=FORMAT_EI)),
    "ei_bad10": (5, dict(seed=433, ei=True, cut=(7, 5), tail_cut=(4, 6), p_bad=0.10, rate_44100=True), dict(format=FORMAT_EI)),
    "ei_wander": (8, dict(seed=434, ei=True, cut=(4, 6), tail_cut=(5, 5), wander=(2, 4), p_bad=0.02), dict(format=FORMAT_EI)),
    "ei_silence": (6, dict(seed=435, ei=True, cut=(6, 6), tail_cut=(3, 3), silent=(1, 2), p_bad=0.01), dict(format=FORMAT_EI)),
    "ei_bff": (4, dict(seed=436, ei=True, cut=(5, 7), tail_cut=(2, 2), bff=True, p_bad=0.02), dict(format=FORMAT_EI, field_order=2)),
    "ei_noise_short": (5, dict(seed=437, ei=True, cut=(40, 30), tail_cut=(160, 150), lead=(2, 2), p_bad=0.05), dict(format=FORMAT_EI)),
    "ei_file_marks": (5, dict(seed=438, ei=True, cut=(5, 7), tail_cut=(3, 3), p_bad=0.03, new_file=True, end_file=True, emphasis=True), dict(format=FORMAT_EI)),
    "ei_picked": (5, dict(seed=439, ei=True, cut=(4, 4), tail_cut=(4, 4), p_bad=0.04, p_picked=0.10, p_forced=0.03), dict(format=FORMAT_EI)),
    "si_tape_as_ei": (4, dict(seed=440, cut=(5, 5), tail_cut=(3, 3), p_bad=0.02), dict(format=FORMAT_EI)),
    "ei_tape_as_si": (4, dict(seed=441, ei=True, cut=(5, 5), p_bad=0.02), {}),
    # lines that reach the stitcher with a sub-line missing or doubled: a field no longer adds up to 735 sub-lines, the frame is queued with
    # the "WRONG COUNT" (pcm16x0datastitcher.cpp:4699-4703) and what does not fill an interleave round waits in conv_queue for the next frame
    "si_lost_sublines": (8, dict(seed=451, cut=(5, 7), tail_cut=(3, 3), p_bad=0.02, rate_44100=True), {}, dict(seed=1, drop=3)),
    "si_doubled_sublines": (8, dict(seed=452, cut=(6, 4), tail_cut=(2, 5), p_bad=0.02), {}, dict(seed=2, dup=4)),
    "si_lost_many": (9, dict(seed=453, cut=(4, 4), tail_cut=(3, 3), p_bad=0.05, emphasis=True), {}, dict(seed=3, drop=40, dup=10)),
    "si_lost_file_marks": (9, dict(seed=454, cut=(5, 7), p_bad=0.03, new_file=True, end_file=True), {}, dict(seed=4, drop=5, dup=3)),
    "ei_lost_sublines": (8, dict(seed=455, ei=True, cut=(5, 7), tail_cut=(3, 3), p_bad=0.02, rate_44100=True), dict(format=FORMAT_EI), dict(seed=5, drop=3)),
    "ei_doubled_sublines": (8, dict(seed=456, ei=True, cut=(6, 4), tail_cut=(2, 5), p_bad=0.02), dict(format=FORMAT_EI), dict(seed=6, dup=4)),
    "ei_lost_many": (9, dict(seed=457, ei=True, cut=(4, 4), tail_cut=(3, 3), p_bad=0.05), dict(format=FORMAT_EI), dict(seed=7, drop=300, dup=10)),
    "ei_lost_file_marks": (9, dict(seed=458, ei=True, cut=(5, 7), tail_cut=(2, 2), p_bad=0.03, new_file=True, end_file=True), dict(format=FORMAT_EI), dict(seed=8, drop=6, dup=2)),
    "ei_lost_bff_no_p": (7, dict(seed=459, ei=True, cut=(5, 7), tail_cut=(2, 2), bff=True, p_bad=0.1), dict(format=FORMAT_EI, field_order=2, p_correction=0), dict(seed=9, drop=7)),
    # long enough for the remainder to grow into a whole extra interleave round: a frame that puts out 525 (SI) / 980 (EI) data blocks
    "si_lost_long": (30, dict(seed=460, cut=(5, 7), tail_cut=(3, 3), p_bad=0.02), {}, dict(seed=10, drop=400)),
    "ei_lost_every_field": (372, dict(seed=462, ei=True, cut=(5, 7), tail_cut=(3, 3)), dict(format=FORMAT_EI), dict(every_field=True)),
}
# records that carry the number of a frame already gone, in the middle of later frames: the reference pops its queue only while the head
# carries the frame's own number (pcm16x0datastitcher.cpp:5711-5732), stops at the stranger and assembles what is left of the same frame
# again and again - it never ends.  The product refuses such a stream.
LIVELOCK = {
    "si_stale_tags": (6, dict(seed=463, cut=(5, 7), tail_cut=(3, 3), p_bad=0.03, new_file=True, end_file=True), {}, dict(stale_tags=True)),
    "ei_stale_tags": (6, dict(seed=464, ei=True, cut=(5, 7), tail_cut=(3, 3), p_bad=0.03, new_file=True, end_file=True), dict(format=FORMAT_EI), dict(stale_tags=True)),
}
WRONG_COUNT = tuple(n for n, c in CASES.items() if len(c) > 3)
GOLDEN = ("si_cut_both", "si_bad10", "si_picked_forced", "si_wander", "si_file_marks", "ei_cut", "ei_bad10", "ei_noise_short",
          "si_lost_sublines", "si_lost_file_marks", "ei_doubled_sublines", "ei_lost_many", "si_lost_long")


def mangle(recs, seed=0, drop=0, dup=0, every_field=False, stale_tags=False):
    """Sub-line records lost / delivered twice at random places of the stream (service records stay); every_field: the middle
    sub-line of one line of every field is lost; stale_tags: the NEW_FILE tag of the stream numbered one frame early, and an END_FILE
    and a NEW_FILE tag numbered one frame early in the middle of frames 2 and 3, a NEW_FILE tag of its own in the middle of frame 4."""
    if stale_tags:
        recs = recs.copy()
        recs["frame_number"][np.nonzero(recs["service_type"] == SRV_NEW_FILE)[0][0]] -= 1
        ends = np.nonzero(recs["service_type"] == SRV_END_FRAME)[0]
        parts, last = [], 0
        for at, srv, d in ((ends[0] + 500, SRV_END_FILE, -1), (ends[1] + 900, SRV_NEW_FILE, -1), (ends[2] + 700, SRV_NEW_FILE, 0)):
            tag = recs[at:at + 1].copy()
            for nm in tag.dtype.names:
                if nm not in ("frame_number", "line_number"):
                    tag[nm] = 0
            tag["service_type"] = srv
            tag["frame_number"] = int(recs["frame_number"][at]) + d
            parts += [recs[last:at], tag]
            last = at
        parts.append(recs[last:])
        return np.concatenate(parts)
    if every_field:
        return recs[~((recs["service_type"] == 0) & (recs["line_number"] >= 200) & (recs["line_number"] <= 201) & (recs["line_part"] == 1))]
    rng = np.random.default_rng(seed)
    data = np.nonzero(recs["service_type"] == 0)[0]
    sel = rng.choice(data, size=drop + dup, replace=False)
    times = np.ones(len(recs), dtype=np.int64)
    times[sel[:drop]] = 0
    times[sel[drop:]] = 2
    return np.repeat(recs, times)


def make_input(name):
    case = CASES[name] if name in CASES else LIVELOCK[name]
    n, kw, st_kw = case[:3]
    recs, _ = make_stream(n, **kw)
    if len(case) > 3:
        recs = mangle(recs, **case[3])
    return recs, default_settings(**st_kw)


def run_cpu(lib, prefix, recs, st, pair_cap=None, frame_cap=None, overflow_ok=False):
    f = getattr(lib, prefix + "pcm16x0_stitch_run")
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Pcm16Settings), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    recs = np.ascontiguousarray(recs)
    nfr = int((recs["service_type"] == SRV_END_FRAME).sum()) + 2
    pair_cap = pair_cap or nfr * 2400 + 16
    frame_cap = frame_cap or nfr + 8
    pairs = np.zeros(pair_cap, dtype=PAIR_DTYPE)
    frames = np.zeros(frame_cap, dtype=FRASM16_DTYPE)
    nf = C.c_size_t(0)
    n = f(recs.ctypes.data, len(recs), C.byref(st), pairs.ctypes.data, pair_cap, frames.ctypes.data, frame_cap, C.byref(nf))
    if n < 0 and overflow_ok:
        return None, nf.value
    assert n >= 0, "pair buffer too small"
    return pairs[:n], frames[:min(nf.value, frame_cap)]


# ---- what the stitcher hands to the visualiser (sdv_set_pcm16x0_stitch_block_output): sdv_pcm16x0_block_rec --------------------------------------
VBLOCK16_DTYPE = np.dtype([("words", "<u2", (3, 3)), ("word_crc", "<u2"), ("word_valid", "<u2"), ("picked_left", "u1"), ("picked_crc", "u1"),
                           ("audio_state", "u1", (3,)), ("flags", "u1"), ("sample_rate", "<u2"), ("_pad", "u1", (2,))])
assert VBLOCK16_DTYPE.itemsize == 32
VIS_GOLDEN = ("si_bad10", "si_picked_forced", "ei_bad10", "si_file_marks")


def run_cpu_vis(lib, prefix, recs, st):
    """(pairs, frames, blocks): the stitcher's run with the visualiser's block feed switched on."""
    f = getattr(lib, prefix + "pcm16x0_stitch_run_vis")
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Pcm16Settings), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    recs = np.ascontiguousarray(recs)
    nfr = int((recs["service_type"] == SRV_END_FRAME).sum()) + 2
    pairs = np.zeros(nfr * 2400 + 16, dtype=PAIR_DTYPE)
    frames = np.zeros(nfr + 8, dtype=FRASM16_DTYPE)
    blocks = np.zeros(nfr * 800 + 16, dtype=VBLOCK16_DTYPE)
    nf, nb = C.c_size_t(0), C.c_size_t(0)
    n = f(recs.ctypes.data, len(recs), C.byref(st), pairs.ctypes.data, len(pairs), frames.ctypes.data, len(frames), C.byref(nf),
          blocks.ctypes.data, len(blocks), C.byref(nb))
    assert n >= 0 and nb.value <= len(blocks)
    return pairs[:n], frames[:nf.value], blocks[:nb.value]


def run_cpu_feeds(lib, prefix, recs, st):
    """(pairs, frames, blocks, lines): ... and with the assembled sub-lines (sdv_set_pcm16x0_stitch_line_output): records of the input's own type."""
    f = getattr(lib, prefix + "pcm16x0_stitch_run_feeds")
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Pcm16Settings), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    recs = np.ascontiguousarray(recs)
    nfr = int((recs["service_type"] == SRV_END_FRAME).sum()) + 2
    pairs = np.zeros(nfr * 2400 + 16, dtype=PAIR_DTYPE)
    frames = np.zeros(nfr + 8, dtype=FRASM16_DTYPE)
    blocks = np.zeros(nfr * 800 + 16, dtype=VBLOCK16_DTYPE)
    lines = np.zeros(nfr * 3000 + 16, dtype=recs.dtype)
    nf, nb, nl = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    n = f(recs.ctypes.data, len(recs), C.byref(st), pairs.ctypes.data, len(pairs), frames.ctypes.data, len(frames), C.byref(nf),
          blocks.ctypes.data, len(blocks), C.byref(nb), lines.ctypes.data, len(lines), C.byref(nl))
    assert n >= 0 and nb.value <= len(blocks) and nl.value <= len(lines)
    return pairs[:n], frames[:nf.value], blocks[:nb.value], lines[:nl.value]


def run_blocks(lib, prefix, recs, n_blocks, ei=False, force=True, p_code=True, ignore_crc=False, first_shift=0, first_even=False):
    f = getattr(lib, prefix + "pcm16x0_deint_blocks")
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    recs = np.ascontiguousarray(recs)
    out = np.zeros(n_blocks, dtype=BLOCK16_DTYPE)
    f(recs.ctypes.data, len(recs), int(ei), int(force), int(p_code), int(ignore_crc), first_shift, int(first_even), out.ctypes.data, n_blocks)
    return out


def digest(pairs, frames):
    return hashlib.sha256(pairs.tobytes() + frames.tobytes()).hexdigest()


def block_inputs():
    """name -> (queue of data sub-lines, run_blocks keywords): damaged SI and EI queues under every switch combination."""
    out = {}
    tapes = [dict(p_bad=0.3, p_picked=0.2, p_forced=0.05), dict(p_bad=0.15, p_picked=0.4, p_nobw=0.1)]
    for t, kw in enumerate(tapes):
        for ei in (False, True):
            recs, _ = make_stream(1, seed=900 + t, ei=ei, **kw)
            data = recs[recs["service_type"] == 0]
            for sw in range(8):
                force, p_code, ign = sw & 1, (sw >> 1) & 1, (sw >> 2) & 1
                out[f"t{t}_{'ei' if ei else 'si'}_{sw}"] = (data, dict(n_blocks=160, ei=ei, force=force, p_code=p_code, ignore_crc=ign,
                                                                    first_shift=0, first_even=bool(t & 1)))
    return out
