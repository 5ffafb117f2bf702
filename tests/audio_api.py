"""AudioProcessor / SamplesToWAV: PODs, seeded PCMSamplePair streams with dropouts and runners shared by the
oracle-vs-reference test, the golden fixture generator (tests/golden/make_golden_audio.py) and the product parity tests."""
import ctypes as C
import hashlib

import numpy as np

from stitch_api import PAIR_DTYPE

PURGE_DTYPE = np.dtype([("first_pair", "<u8"), ("tag_index", "<u4"), ("kind", "u1"), ("_pad", "u1", (3,))])
assert PAIR_DTYPE.itemsize == 12 and PURGE_DTYPE.itemsize == 16

SF_BLOCK_OK, SF_WORD_VALID, SF_WORD_FIXED, SF_WORD_MASKED = 1, 2, 4, 8
SRV_NEW_FILE, SRV_END_FILE = 1, 2
DROP_IGNORE, DROP_MUTE_BLOCK, DROP_MUTE_WORD, DROP_HOLD_BLOCK, DROP_HOLD_WORD, DROP_INTER_LIN_BLOCK, DROP_INTER_LIN_WORD = range(7)
PURGE_NEW_FILE, PURGE_END_FILE, PURGE_STOP = 1, 2, 3


def tag(kind):
    t = np.zeros(1, dtype=PAIR_DTYPE)
    t["service_type"] = kind
    t["sample_rate"] = 44056
    return t


def audio(n, seed, rate=44056, amp=12000, runs=(), p_bad=0.0, p_block=0.0, tone=True, emphasis=0):
    """n data pairs: a tone plus noise, all valid; `runs` = (start, length, channels) dropouts (channels: 0 left, 1 right, 2 both),
    p_bad = single invalid words at random, p_block = whole data blocks (three pairs) that did not decode (block and word flags off,
    in one channel or in both)."""
    rng = np.random.default_rng(seed)
    a = np.zeros(n, dtype=PAIR_DTYPE)
    t = np.arange(n)
    for ch in range(2):
        w = rng.integers(-(amp // 8) - 1, amp // 8 + 1, n)
        if tone:
            w = w + (amp * np.sin(2 * np.pi * t * (0.003 + 0.002 * ch) + ch)).astype(np.int64)
        a["audio_word"][:, ch] = np.clip(w, -32768, 32767)
    a["sample_flags"] = SF_BLOCK_OK | SF_WORD_VALID
    a["sample_rate"] = rate
    a["emphasis"] = emphasis
    for s, ln, chs in runs:
        for ch in ((0, 1) if chs == 2 else (chs,)):
            a["sample_flags"][s:s + ln, ch] &= ~SF_WORD_VALID & 0xFF
            a["audio_word"][s:s + ln, ch] = rng.integers(-amp - 1, amp + 1, len(a["audio_word"][s:s + ln, ch]))     # what a bad word holds
    if p_bad:
        m = rng.random((n, 2)) < p_bad
        a["sample_flags"][m] &= ~SF_WORD_VALID & 0xFF
    if p_block:
        nb = (n + 2) // 3
        hit = rng.random(nb) < p_block
        which = rng.integers(0, 3, nb)                      # 0 left, 1 right, 2 both
        for ch in range(2):
            sel = np.repeat(hit & ((which == ch) | (which == 2)), 3)[:n]
            a["sample_flags"][sel, ch] &= ~(SF_WORD_VALID | SF_BLOCK_OK) & 0xFF
        # a block the deinterleaver gave up on although some of its words had a good CRC: word flags stay on
        keep = np.repeat(hit & (rng.random(nb) < 0.3), 3)[:n]
        a["sample_flags"][keep & ((a["sample_flags"][:, 0] & SF_BLOCK_OK) == 0), 0] |= SF_WORD_VALID
    # words the error correction repaired: valid, flagged as fixed
    fx = (rng.random((n, 2)) < 0.01) & ((a["sample_flags"] & SF_WORD_VALID) != 0)
    a["sample_flags"][fx] |= SF_WORD_FIXED
    return a


def tape(parts):
    """parts: arrays and 'N' / 'E' for the NEW_FILE / END_FILE tags."""
    return np.concatenate([tag(SRV_NEW_FILE) if isinstance(p, str) and p == "N" else tag(SRV_END_FILE) if isinstance(p, str) else p for p in parts])


def _c_clean():
    return tape(["N", audio(3000, 1), "E"])


def _c_short_runs():
    return tape(["N", audio(4000, 2, runs=[(100, 5, 0), (300, 17, 1), (640, 30, 2), (1500, 1, 0), (1502, 1, 0), (2600, 60, 2), (3990, 4, 1)]), "E"])


def _c_long_runs():
    return tape(["N", audio(6000, 3, runs=[(200, 230, 0), (700, 225, 1), (1200, 226, 2), (1700, 224, 0), (2100, 500, 2), (3100, 1400, 0), (5000, 33, 1), (5100, 32, 1)]), "E"])


def _c_window_edges():
    # runs placed around the places where windows start and end (509-pair stride from the first pair of the file)
    runs = [(509 - 230, 10, 0), (509 * 2 - 226, 40, 1), (509 * 3 - 225, 300, 2), (509 * 5 - 3, 8, 0), (509 * 6 - 1, 3, 1), (509 * 7 + 284, 2, 2), (509 * 8 + 287, 250, 0), (509 * 10 + 500, 20, 1)]
    return tape(["N", audio(509 * 12 + 77, 4, runs=runs), "E"])


def _c_random_words():
    return tape(["N", audio(5000, 5, p_bad=0.02), "E"])


def _c_random_heavy():
    return tape(["N", audio(5000, 6, p_bad=0.3, runs=[(2000, 700, 2)]), "E"])


def _c_blocks():
    return tape(["N", audio(6000, 7, p_block=0.05, p_bad=0.01), "E"])


def _c_tail_bad():
    return tape(["N", audio(2500, 8, runs=[(2300, 200, 2)]), "E"])


def _c_tail_bad_short():
    return tape(["N", audio(1800, 9, runs=[(1790, 10, 0), (1700, 100, 1)]), "E"])


def _c_head_bad():
    return tape(["N", audio(2000, 10, runs=[(0, 40, 2), (60, 300, 0)]), "E"])


def _c_all_bad():
    return tape(["N", audio(1500, 11, runs=[(0, 1500, 0), (10, 1480, 1)]), "E"])


def _c_two_files():
    return tape(["N", audio(1300, 12, runs=[(600, 50, 2)]), "E", "N", audio(2100, 13, rate=44100, runs=[(20, 400, 1), (2050, 50, 0)]), "E"])


def _c_no_end_tag():
    # a new file without the end of the one before: what waited goes out as it is; the last file has no end either (stop)
    return tape(["N", audio(1400, 14, runs=[(1000, 100, 0), (1350, 30, 1)]), "N", audio(900, 15, runs=[(500, 80, 2)])])


def _c_tiny_files():
    return tape(["N", audio(3, 16), "E", "N", audio(2, 17, runs=[(1, 1, 0)]), "E", "N", audio(230, 18, runs=[(100, 20, 2)]), "E", "N", audio(226, 19, runs=[(5, 200, 2)]), "E",
                 "N", "E", "N", audio(1, 28), "E", "N", audio(700, 29, runs=[(300, 30, 1)]), "E"])


def _c_tiny_files_bad():
    # END_FILE tags that find one or two pairs in the window (no purge: the pairs are scanned as the end of a file and stay), with invalid words among them,
    # the very first pairs of a stream without a tag included
    return tape([audio(2, 40, runs=[(1, 1, 2)]), "E", audio(1, 41, runs=[(0, 1, 0)]), "E", audio(300, 42, runs=[(100, 20, 2)]), "E",
                 "N", audio(1, 43, runs=[(0, 1, 1)]), "E", "E", audio(1, 44), "E", audio(400, 45, runs=[(0, 3, 2), (200, 230, 0)]), "E", "N", audio(1, 46, runs=[(0, 1, 2)]), "E"])


def _c_stalled_start():
    # a stream that starts with invalid samples and no NEW_FILE tag: the window fills up, nothing can ever leave, the worker stops taking input
    return tape([audio(2000, 77, runs=[(0, 5, 2)]), "E", "N", audio(500, 78), "E"])


def _c_tiny_then_long():
    # ... the same behind an END_FILE that did not purge (one pair in the window, invalid in one channel)
    return tape(["N", audio(900, 79, runs=[(300, 40, 2)]), "E", "N", audio(1, 80, runs=[(0, 1, 1)]), "E", audio(1, 83, runs=[(0, 1, 1)]), audio(1500, 81, runs=[(0, 2, 1), (700, 10, 2)]), "N", audio(100, 82), "E"])


def _c_small_files():
    return tape(["N", audio(3, 16), "E", "N", audio(2, 17, runs=[(1, 1, 0)]), "E", "N", audio(230, 18, runs=[(100, 20, 2)]), "E", "N", audio(226, 19, runs=[(5, 200, 2)]), "E",
                 "N", audio(9, 30, runs=[(3, 6, 2)]), "E", "N", audio(700, 29, runs=[(300, 30, 1)]), "E"])


def _c_exact_windows():
    # the data of a file ends exactly where a window is full: the end tag gets a turn of its own
    return tape(["N", audio(511, 20, runs=[(400, 111, 0)]), "E", "N", audio(511 + 509, 21, runs=[(900, 120, 1)]), "E", "N", audio(511 + 2 * 509, 22), "E"])


def _c_masked_zero_start():
    # a long run that is still open when its window ends: the next scan starts from a zero an earlier scan put there
    return tape(["N", audio(4000, 23, runs=[(250, 900, 0), (1300, 20, 0), (1500, 1000, 1), (2600, 40, 2)]), "E"])


def _c_silence():
    a = audio(3000, 24, amp=0, tone=False, runs=[(500, 300, 2), (1200, 20, 0)])
    a["audio_word"] = 0
    return tape(["N", a, "E"])


def _c_extreme_levels():
    a = audio(3000, 25, runs=[(300, 100, 0), (600, 226, 1), (1400, 30, 2)])
    a["audio_word"][299, 0] = 32767; a["audio_word"][400, 0] = -32768
    a["audio_word"][599, 1] = -32768; a["audio_word"][826, 1] = 32767
    a["audio_word"][1399] = (32767, -32768); a["audio_word"][1430] = (-32768, 32767)
    return tape(["N", a, "E"])


def _c_input_masked_flag():
    a = audio(2500, 26, runs=[(700, 40, 0), (1200, 300, 1)])
    a["sample_flags"][699, 0] |= SF_WORD_MASKED; a["audio_word"][699, 0] = 0       # looks like a zero an earlier scan put there
    a["sample_flags"][1199, 1] |= SF_WORD_MASKED
    return tape(["N", a, "E"])


def _c_long_tape():
    return tape(["N", audio(40000, 27, p_bad=0.001, runs=[(9000, 400, 2), (20000, 3000, 0), (30000, 20, 1)]), "E"])


def _c_worn_tape():
    # an invalid word in almost every window: the windows are 509 pairs apart until one ends on an invalid pair
    return tape(["N", audio(70000, 33, p_bad=0.01, runs=[(30000, 3, 2), (50897, 5, 0)]), "E"])


def _c_no_first_tag():
    # a stream that starts without NEW_FILE: no silent pair in front, the index starts at 0
    return tape([audio(1500, 31, runs=[(300, 40, 2)]), "E", "N", audio(800, 32), "E"])


# name: (builder, mask mode, bursts (fractions of the stream) or None, stop)
CASES = {
    "clean": (_c_clean, DROP_INTER_LIN_WORD, None, 1),
    "short_runs_lin": (_c_short_runs, DROP_INTER_LIN_WORD, None, 1),
    "short_runs_hold": (_c_short_runs, DROP_HOLD_WORD, None, 1),
    "short_runs_mute": (_c_short_runs, DROP_MUTE_WORD, None, 1),
    "long_runs_lin": (_c_long_runs, DROP_INTER_LIN_WORD, None, 1),
    "long_runs_hold": (_c_long_runs, DROP_HOLD_WORD, None, 1),
    "long_runs_mute": (_c_long_runs, DROP_MUTE_WORD, None, 1),
    "window_edges": (_c_window_edges, DROP_INTER_LIN_WORD, None, 1),
    "random_words": (_c_random_words, DROP_INTER_LIN_WORD, None, 1),
    "random_heavy": (_c_random_heavy, DROP_INTER_LIN_WORD, None, 1),
    "random_heavy_hold": (_c_random_heavy, DROP_HOLD_WORD, None, 1),
    "blocks_by_block": (_c_blocks, DROP_INTER_LIN_BLOCK, None, 1),
    "blocks_by_word": (_c_blocks, DROP_INTER_LIN_WORD, None, 1),
    "blocks_mute_block": (_c_blocks, DROP_MUTE_BLOCK, None, 1),
    "blocks_hold_block": (_c_blocks, DROP_HOLD_BLOCK, None, 1),
    "ignore": (_c_blocks, DROP_IGNORE, None, 1),
    "ignore_no_end": (_c_no_end_tag, DROP_IGNORE, None, 1),
    "tail_bad": (_c_tail_bad, DROP_INTER_LIN_WORD, None, 1),
    "tail_bad_mute": (_c_tail_bad, DROP_MUTE_WORD, None, 1),
    "tail_bad_short": (_c_tail_bad_short, DROP_HOLD_WORD, None, 1),
    "head_bad": (_c_head_bad, DROP_INTER_LIN_WORD, None, 1),
    "all_bad": (_c_all_bad, DROP_INTER_LIN_WORD, None, 1),
    "two_files": (_c_two_files, DROP_INTER_LIN_WORD, None, 1),
    "no_end_tag_stop": (_c_no_end_tag, DROP_INTER_LIN_WORD, None, 1),
    "no_end_tag_open": (_c_no_end_tag, DROP_INTER_LIN_WORD, None, 0),
    "no_first_tag": (_c_no_first_tag, DROP_INTER_LIN_WORD, None, 1),
    "tiny_files": (_c_tiny_files, DROP_INTER_LIN_WORD, None, 1),
    "small_files": (_c_small_files, DROP_INTER_LIN_WORD, None, 1),
    "tiny_files_bad": (_c_tiny_files_bad, DROP_INTER_LIN_WORD, None, 1),
    "tiny_files_bad_ignore": (_c_tiny_files_bad, DROP_IGNORE, None, 0),
    "tiny_files_bad_block_bursts": (_c_tiny_files_bad, DROP_HOLD_BLOCK, (0.001, 0.003, 0.4, 0.43, 1.0), 1),
    "stalled_start": (_c_stalled_start, DROP_INTER_LIN_WORD, None, 1),
    "stalled_start_open": (_c_stalled_start, DROP_MUTE_WORD, None, 0),
    "stalled_start_bursts": (_c_stalled_start, DROP_INTER_LIN_WORD, (0.1, 0.2, 0.5, 0.9, 1.0), 1),
    "tiny_then_long": (_c_tiny_then_long, DROP_INTER_LIN_WORD, None, 1),
    "exact_windows": (_c_exact_windows, DROP_INTER_LIN_WORD, None, 1),
    "masked_zero_start": (_c_masked_zero_start, DROP_INTER_LIN_WORD, None, 1),
    "masked_zero_start_hold": (_c_masked_zero_start, DROP_HOLD_WORD, None, 1),
    "silence": (_c_silence, DROP_INTER_LIN_WORD, None, 1),
    "extreme_levels": (_c_extreme_levels, DROP_INTER_LIN_WORD, None, 1),
    "input_masked_flag": (_c_input_masked_flag, DROP_INTER_LIN_WORD, None, 1),
    "long_tape": (_c_long_tape, DROP_INTER_LIN_WORD, None, 1),
    "worn_tape": (_c_worn_tape, DROP_INTER_LIN_WORD, None, 1),
    "worn_tape_hold_bursts": (_c_worn_tape, DROP_HOLD_WORD, (0.3, 0.31, 0.77, 1.0), 1),
    # the queue runs dry in between: bursts
    "bursts_long_runs": (_c_long_runs, DROP_INTER_LIN_WORD, (0.13, 0.5, 0.51, 0.9, 1.0), 1),
    "bursts_two_files": (_c_two_files, DROP_HOLD_WORD, (0.2, 0.4, 0.6, 1.0), 0),
    "bursts_small": (_c_short_runs, DROP_INTER_LIN_WORD, (0.02, 0.05, 0.1, 0.16, 0.3, 1.0), 1),
}
GOLDEN = ("short_runs_lin", "long_runs_hold", "window_edges", "blocks_by_block", "two_files", "no_end_tag_stop", "masked_zero_start", "bursts_long_runs", "ignore",
          "tiny_files", "tiny_files_bad", "stalled_start", "tiny_then_long")
DEAD_ENDS = ("tiny_files", "tiny_files_bad", "tiny_files_bad_ignore", "tiny_files_bad_block_bursts", "stalled_start", "stalled_start_open", "stalled_start_bursts",
             "tiny_then_long")      # the worker's two dead ends (an END_FILE that does not purge, a window nothing can leave): reproduced as they are


def make_input(name):
    build, mode, bursts, stop = CASES[name]
    pairs = build()
    n = len(pairs)
    ends = np.array([n] if bursts is None else sorted(set(min(n, max(1, int(round(f * n)))) for f in bursts) | {n}), dtype=np.uint64)
    return pairs, mode, ends, stop


def run_cpu(lib, prefix, pairs, mode, ends, stop, wav_dir=None):
    """-> (out pairs, index of each, purges, masked count, hit_unsupported); purges are positions only for the reference"""
    pairs = np.ascontiguousarray(pairs)
    ends = np.ascontiguousarray(ends, dtype=np.uint64)
    cap = len(pairs) + 1024
    out = np.zeros(cap, dtype=PAIR_DTYPE)
    idx = np.zeros(cap, dtype=np.uint64)
    ntags = int((pairs["service_type"] != 0).sum()) + 2
    npur, nm = C.c_size_t(0), C.c_uint64(0)
    f = getattr(lib, prefix + "audio_run")
    f.restype = C.c_long
    if prefix == "ref_":
        pur = np.zeros(ntags, dtype=np.uint64)
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                      C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_char_p]
        r = f(pairs.ctypes.data, len(pairs), ends.ctypes.data, len(ends), mode, stop, out.ctypes.data, idx.ctypes.data, cap, pur.ctypes.data, ntags,
              C.byref(npur), C.byref(nm), wav_dir.encode() if wav_dir else None)
        assert r >= 0
        return out[:r], idx[:r], pur[:npur.value], nm.value, None
    pur = np.zeros(ntags, dtype=PURGE_DTYPE)
    hit = C.c_int(0)
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                  C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    r = f(pairs.ctypes.data, len(pairs), ends.ctypes.data, len(ends), mode, stop, out.ctypes.data, idx.ctypes.data, cap, pur.ctypes.data, ntags,
          C.byref(npur), C.byref(nm), C.byref(hit))
    assert r >= 0
    return out[:r], idx[:r], pur[:npur.value], nm.value, hit.value


def expected_index(n_out, purge_positions):
    """PCMSample::index of the pairs of an output stream: it counts from 0 behind every purge (and from 0 at the start)."""
    idx = np.arange(n_out, dtype=np.uint64)
    for fp in purge_positions:
        idx[int(fp):] = np.arange(n_out - int(fp), dtype=np.uint64)
    return idx


def wav_files(lib, prefix, out, purges):
    """The files SamplesToWAV leaves for an output stream: one per NEW_FILE purge that is followed by at least one pair,
    as (number of the NEW_FILE tag, bytes)."""
    hdr_f = getattr(lib, prefix + "wav_header")
    hdr_f.argtypes = [C.c_void_p, C.c_uint64, C.c_uint16]
    hdr_f.restype = None
    files = []
    k = 0
    for i, p in enumerate(purges):
        if p["kind"] != PURGE_NEW_FILE:
            continue
        a = int(p["first_pair"])
        b = int(purges[i + 1]["first_pair"]) if i + 1 < len(purges) else len(out)
        if b > a:
            hdr = np.zeros(44, dtype=np.uint8)
            hdr_f(hdr.ctypes.data, b - a, int(out["sample_rate"][b - 1]))
            files.append((k, hdr.tobytes() + np.ascontiguousarray(out["audio_word"][a:b]).astype("<i2").tobytes()))
        k += 1
    return files


def digest(out, purges, masked):
    return hashlib.sha256(out.tobytes() + np.ascontiguousarray(purges).tobytes() + str(masked).encode()).hexdigest()


def bind_product(lib):
    """argtypes of the audio entry points of the C-ABI (product library or its emulator build)."""
    lib.sdv_set_audio_masking.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_reset_audio.argtypes = [C.c_void_p]
    lib.sdv_audio_pending.restype = C.c_size_t
    lib.sdv_audio_pending.argtypes = [C.c_void_p]
    lib.sdv_audio_next_index.restype = C.c_uint64
    lib.sdv_audio_next_index.argtypes = [C.c_void_p]
    lib.sdv_audio_process.restype = C.c_int
    lib.sdv_audio_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_void_p]
    lib.sdv_wav_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.sdv_wav_header.argtypes = [C.c_void_p, C.c_uint64, C.c_uint16]
    lib.sdv_wav_header.restype = None
    return lib


def emu_audio(lib, eng, pairs, stop, out_cap=None, purges_cap=None):
    """Host-memory call (emulator build only): one sdv_audio_process call over `pairs` -> (rc, out, purges, masked)."""
    pairs = np.ascontiguousarray(pairs)
    out_cap = len(pairs) + 1024 if out_cap is None else out_cap
    purges_cap = int((pairs["service_type"] != 0).sum()) + 2 if purges_cap is None else purges_cap
    out = np.zeros(max(out_cap, 1), dtype=PAIR_DTYPE)
    pur = np.zeros(max(purges_cap, 1), dtype=PURGE_DTYPE)
    n_out, n_pur, nm = C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
    rc = lib.sdv_audio_process(eng, pairs.ctypes.data if len(pairs) else None, len(pairs), stop, out.ctypes.data, out_cap, C.byref(n_out),
                               pur.ctypes.data, purges_cap, C.byref(n_pur), C.byref(nm), None)
    return rc, out[:min(n_out.value, out_cap)], pur[:min(n_pur.value, purges_cap)], nm.value, n_out.value, n_pur.value


def emu_run(lib, pairs, mode, ends, stop):
    """The bursts of a case through a fresh engine of the emulator build, like run_cpu: -> (out, purges, masked)."""
    eng = lib.sdv_engine_create(0)
    assert lib.sdv_set_audio_masking(eng, mode) == 0
    outs, purs, masked, a, got = [], [], 0, 0, 0
    try:
        for k, b in enumerate(ends):
            b = int(b)
            rc, o, p, m, _, _ = emu_audio(lib, eng, pairs[a:b], 1 if (stop and k + 1 == len(ends)) else 0)
            assert rc == 0, lib.sdv_last_error(eng)
            p = p.copy()
            p["first_pair"] += got
            p["tag_index"] += a
            outs.append(o.copy()); purs.append(p); masked += m
            got += len(o); a = b
    finally:
        lib.sdv_engine_destroy(eng)
    return np.concatenate(outs), np.concatenate(purs), masked
