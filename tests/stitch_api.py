"""ctypes helpers for the stitch stage (STC007DataStitcher): oracle / reference / product share the PODs."""
import ctypes as C
import numpy as np
import libs

PAIR_DTYPE = np.dtype([("audio_word", "<i2", (2,)), ("sample_flags", "u1", (2,)), ("sample_rate", "<u2"),
                       ("emphasis", "u1"), ("service_type", "u1"), ("_pad", "<u2")])
FRASM_DTYPE = np.dtype([("frame_number", "<u4"),
                        ("odd_std_lines", "<u2"), ("even_std_lines", "<u2"), ("odd_data_lines", "<u2"), ("even_data_lines", "<u2"),
                        ("odd_valid_lines", "<u2"), ("even_valid_lines", "<u2"),
                        ("odd_top_data", "<u2"), ("odd_bottom_data", "<u2"), ("even_top_data", "<u2"), ("even_bottom_data", "<u2"),
                        ("odd_sample_rate", "<u2"), ("even_sample_rate", "<u2"),
                        ("blocks_total", "<u2"), ("blocks_drop", "<u2"), ("samples_drop", "<u2"),
                        ("inner_padding", "<u2"), ("outer_padding", "<u2"),
                        ("blocks_broken_field", "<u2"), ("blocks_broken_seam", "<u2"),
                        ("blocks_fix_p", "<u2"), ("blocks_fix_q", "<u2"), ("blocks_fix_cwd", "<u2"),
                        ("field_order", "u1"), ("odd_ref", "u1"), ("even_ref", "u1"), ("service_type", "u1"),
                        ("video_standard", "u1"), ("tff_cnt", "u1"), ("bff_cnt", "u1"), ("odd_resolution", "u1"), ("even_resolution", "u1"),
                        ("flags", "u1"), ("flags2", "u1"),
                        ("ctrl_index", "i1"), ("ctrl_hour", "i1"), ("ctrl_minute", "i1"), ("ctrl_second", "i1"), ("ctrl_field", "i1")])
assert PAIR_DTYPE.itemsize == 12 and FRASM_DTYPE.itemsize == 64, (PAIR_DTYPE.itemsize, FRASM_DTYPE.itemsize)


class StitchSettings(C.Structure):
    _fields_ = [("video_standard", C.c_uint8), ("field_order", C.c_uint8), ("enable_p", C.c_uint8), ("enable_q", C.c_uint8),
                ("enable_cwd", C.c_uint8), ("m2_format", C.c_uint8), ("resolution_preset", C.c_uint8), ("max_unch_14", C.c_uint8),
                ("max_unch_16", C.c_uint8), ("use_ecc", C.c_uint8), ("mask_seams", C.c_uint8), ("broke_mask", C.c_uint8),
                ("top_line_fix", C.c_uint8), ("_pad", C.c_uint8), ("sample_rate_preset", C.c_uint16)]


assert C.sizeof(StitchSettings) == 16


def default_settings(**kw):
    lib = libs.load_oracle()
    st = StitchSettings()
    lib.orc_default_stitch_settings(C.byref(st))
    st.enable_p = 1
    st.enable_q = 1              # what the application switches on (mainwindow defaults)
    for k, v in kw.items():
        setattr(st, k, v)
    return st


def with_end_file(recs):
    """Appends what the input plugin puts after the last frame of a file (VideoInFFMPEG::insertDummyFrame(true, false),
    vin_ffmpeg.cpp:367-523, as it passes through VideoToDigital): one frame of FILLER service lines in field order
    (rows 1,3,5.. END_FIELD, rows 2,4,6.. END_FIELD), then END_FILE and END_FRAME."""
    last = recs["frame_number"][-1]
    h = int(((recs["frame_number"] == last) & (recs["service_type"] == 0)).sum())
    tail = np.zeros(h + 4, dtype=libs.LINE_DTYPE)
    tail["frame_number"] = last + 1
    k = 0
    ln = 0
    for first in (1, 2):
        for ln in range(first, h + 1, 2):
            tail["line_number"][k] = ln
            tail["service_type"][k] = 3         # SDV_SRV_FILLER
            k += 1
        ln += 2
        tail["line_number"][k] = ln
        tail["service_type"][k] = 4             # SDV_SRV_END_FIELD
        k += 1
    ln += 2
    tail["line_number"][k] = ln
    tail["service_type"][k] = 2                 # SDV_SRV_END_FILE
    ln += 2
    tail["line_number"][k + 1] = ln
    tail["service_type"][k + 1] = 5             # SDV_SRV_END_FRAME
    assert k + 2 == len(tail)
    return np.concatenate([recs, tail])


def run_cpu(lib, prefix, recs, st, pair_cap=None, frame_cap=None):
    f = getattr(lib, prefix + "stitch_run")
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(StitchSettings), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    recs = np.ascontiguousarray(recs)
    nfr = int(recs["frame_number"].max() - recs["frame_number"].min()) + 4
    pair_cap = pair_cap or nfr * 2100 + 4096
    frame_cap = frame_cap or nfr + 8
    pairs = np.zeros(pair_cap, dtype=PAIR_DTYPE)
    frames = np.zeros(frame_cap, dtype=FRASM_DTYPE)
    nf = C.c_size_t(0)
    n = f(recs.ctypes.data, len(recs), C.byref(st), pairs.ctypes.data, pair_cap, frames.ctypes.data, frame_cap, C.byref(nf))
    assert n >= 0, "pair buffer too small"
    return pairs[:n], frames[:min(nf.value, frame_cap)]


BLOCK_DTYPE = np.dtype([("w_frame", "<u4", (8,)), ("w_line", "<u2", (8,)), ("words", "<u2", (8,)), ("line_crc", "u1"), ("cwd_fixed", "u1"), ("word_valid", "u1"),
                        ("resolution", "u1"), ("audio_state", "u1"), ("cwd_applied", "u1"), ("sample_rate", "<u2")])     # sdv_block_rec, 72 bytes


def run_cpu_blocks(lib, prefix, recs, st):
    """... and the data blocks the stitcher hands to the visualiser (newBlockProcessed): (pairs, frames, blocks)."""
    f = getattr(lib, prefix + "stitch_run_blocks")
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(StitchSettings), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    recs = np.ascontiguousarray(recs)
    nfr = int(recs["frame_number"].max() - recs["frame_number"].min()) + 4
    pair_cap, frame_cap = nfr * 2100 + 4096, nfr + 8
    pairs = np.zeros(pair_cap, dtype=PAIR_DTYPE)
    frames = np.zeros(frame_cap, dtype=FRASM_DTYPE)
    blocks = np.zeros(pair_cap // 3 + 1, dtype=BLOCK_DTYPE)
    nf, nb = C.c_size_t(0), C.c_size_t(0)
    n = f(recs.ctypes.data, len(recs), C.byref(st), pairs.ctypes.data, pair_cap, frames.ctypes.data, frame_cap, C.byref(nf), blocks.ctypes.data, len(blocks), C.byref(nb))
    assert n >= 0 and nb.value <= len(blocks)
    return pairs[:n], frames[:min(nf.value, frame_cap)], blocks[:nb.value]


ASM_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (9,)), ("calc_crc", "<u2"), ("word_crc_ok", "<u2"), ("word_valid", "<u2"),
                      ("flags", "u1"), ("_pad", "u1")])       # sdv_asm_line_rec, 32 bytes


def last_asm_lines(lib, prefix):
    """The assembled lines (newLineProcessed) of the last run_cpu_blocks call on that library: (lines, lines per stitcher turn)."""
    f = getattr(lib, prefix + "stitch_last_asm_lines")
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    nl, nt = C.c_size_t(0), C.c_size_t(0)
    f(None, 0, C.byref(nl), None, 0, C.byref(nt))
    lines, per = np.zeros(nl.value, dtype=ASM_DTYPE), np.zeros(nt.value, dtype=np.uint32)
    f(lines.ctypes.data, len(lines), C.byref(nl), per.ctypes.data, len(per), C.byref(nt))
    return lines, per
