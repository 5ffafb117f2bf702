"""PCM-1 back half (PCM1DataStitcher): PODs, seeded record streams and runners shared by the oracle-vs-reference test,
the golden fixture generator (tests/golden/make_golden_pcm1.py) and the product parity tests."""
import ctypes as C
import hashlib

import numpy as np

import libs
from stitch_api import PAIR_DTYPE

LINE1_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (7,)), ("calc_crc", "<u2"),
                        ("ref_level", "u1"), ("picked_bits_left", "u1"), ("picked_bits_right", "u1"), ("service_type", "u1"),
                        ("flags", "u1"), ("_pad", "u1", (5,))])
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


def crc_words(words6):
    """PCM1Line::calcCRC over rows of six 13-bit words (vectorised restatement used only to BUILD inputs; the oracle's own
    orc_pcm1_crc_words and the reference's calcCRC are what the tests pin against each other)."""
    w = (~np.asarray(words6, dtype=np.uint32)) & 0x1FFF
    crc = np.full(w.shape[0], 0xFFFF, dtype=np.uint32)
    for i in range(6):
        for b in range(12, -1, -1):
            bit = (w[:, i] >> b) & 1
            top = ((crc >> 15) & 1) ^ bit
            crc = ((crc << 1) & 0xFFFF) ^ (top * 0x1021)
    return ((~crc) & 0xFFFF).astype(np.uint16)


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
}
GOLDEN = ("header_emph", "bad5", "noise_outside", "file_marks", "manual_offsets", "empty_frames")


def make_input(name):
    n, kw, st_kw = CASES[name]
    return make_stream(n, **kw), default_settings(**st_kw)


def make_stream(n_frames, seed=0, lines=(245, 245), first=(1, 2), header=0, footer=0, p_bad=0.0, p_nobw=0.0, p_picked=0.0, p_forced=0.0,
                p_filler=0.0, noise_lines=0, new_file=False, end_file=False, empty=(), one_field=(), burst=None, first_frame=1):
    """The PCM1Line stream of a synthetic PCM-1 tape as the PCM-1 VideoToDigital branch would queue it: per frame the odd rows,
    END_FIELD, the even rows, END_FIELD, END_FRAME (videotodigital.cpp:1189-1383); header/footer rows as HEADER service lines."""
    rng = np.random.default_rng(seed)
    out = []

    def rec(frame, line, srv=0):
        r = np.zeros(1, dtype=LINE1_DTYPE)
        r["frame_number"] = frame
        r["line_number"] = line
        r["service_type"] = srv
        if srv:        # a cleared line (PCM1Line::clear): silent words, CRC_SILENT against its inverse
            r["words"][0, :6] = 1 << 12
            r["calc_crc"] = 0xECBF
            r["words"][0, 6] = 0xECBF ^ 0xFFFF
        return r

    for fi in range(n_frames):
        frame = first_frame + fi
        last_line = 0
        if new_file and fi == 0:
            out.append(rec(frame, 0, SRV_NEW_FILE))
        for field in (0, 1):
            if fi in empty or (fi in one_field and field == 1):
                cnt = 0
            else:
                cnt = lines[field]
            ln = first[field]
            for _ in range(noise_lines if fi % 2 == 0 else 0):         # garbage rows above the data: B/W found, CRC bad
                r = rec(frame, ln)
                r["words"][0, :6] = rng.integers(0, 1 << 13, size=6)
                r["calc_crc"] = crc_words(r["words"][:, :6])
                r["words"][0, 6] = r["calc_crc"][0] ^ np.uint16(rng.integers(1, 1 << 16))
                r["flags"] = LF_BW_SET
                r["ref_level"] = rng.integers(40, 200)
                out.append(r)
                ln += 2
            for _ in range(header):
                out.append(rec(frame, ln, SRV_HEADER))
                ln += 2
            block = np.zeros(cnt, dtype=LINE1_DTYPE)
            block["frame_number"] = frame
            block["line_number"] = ln + 2 * np.arange(cnt)
            block["words"][:, :6] = rng.integers(0, 1 << 13, size=(cnt, 6))
            small = rng.random((cnt, 6)) < 0.5                          # half of the words in the fine range (range bit clear)
            block["words"][:, :6] = np.where(small, block["words"][:, :6] & 0x0FFF, block["words"][:, :6])
            block["calc_crc"] = crc_words(block["words"][:, :6]) if cnt else 0
            block["words"][:, 6] = block["calc_crc"]
            block["flags"] = LF_BW_SET
            block["ref_level"] = rng.integers(60, 180, size=cnt)
            bad = rng.random(cnt) < p_bad
            block["words"][bad, 6] ^= rng.integers(1, 1 << 16, size=int(bad.sum())).astype(np.uint16)
            nobw = rng.random(cnt) < p_nobw
            block["flags"][nobw] = 0
            pk = rng.random(cnt) < p_picked
            block["picked_bits_left"][pk] = rng.integers(0, 4, size=int(pk.sum()))
            block["picked_bits_right"][pk] = rng.integers(0, 3, size=int(pk.sum()))
            fb = rng.random(cnt) < p_forced
            block["flags"][fb] |= LF_FORCED_BAD
            fil = rng.random(cnt) < p_filler
            for i in np.nonzero(fil)[0]:
                block[i] = rec(frame, block["line_number"][i], SRV_FILLER)[0]
            if burst and fi == 1 and field == 0 and cnt:
                s, length = burst
                s = min(s, 2 * cnt) // 2
                block["words"][s:s + length // 2, 6] ^= 0x5A5A
            out.append(block)
            ln += 2 * cnt
            for _ in range(footer):
                out.append(rec(frame, ln, SRV_HEADER))
                ln += 2
            out.append(rec(frame, ln, SRV_END_FIELD))
            last_line = max(last_line, ln)
        out.append(rec(frame, last_line + 2, SRV_END_FRAME))
    if end_file:
        frame = first_frame + n_frames
        for field in (0, 1):
            for ln in range(1 + field, 491, 2):
                out.append(rec(frame, ln, SRV_FILLER))
            out.append(rec(frame, 491 + field, SRV_END_FIELD))
        out.append(rec(frame, 494, SRV_END_FILE))
        out.append(rec(frame, 496, SRV_END_FRAME))
    return np.concatenate(out)


def run_cpu(lib, prefix, recs, st, pair_cap=None, frame_cap=None):
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
    assert n >= 0, "pair buffer too small"
    return pairs[:n], frames[:min(nf.value, frame_cap)]


def digest(pairs, frames):
    return hashlib.sha256(pairs.tobytes() + frames.tobytes()).hexdigest()
