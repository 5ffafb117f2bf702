"""ctypes helpers for the deinterleave stage (oracle / reference / emulator / product share the PODs)."""
import ctypes as C
import numpy as np

DEINT_LINE_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (8,)),
                             ("word_crc_ok", "u1"), ("flags", "u1")])
BLOCK_DTYPE = np.dtype([("w_frame", "<u4", (8,)), ("w_line", "<u2", (8,)), ("words", "<u2", (8,)),
                        ("line_crc", "u1"), ("cwd_fixed", "u1"), ("word_valid", "u1"), ("resolution", "u1"),
                        ("audio_state", "u1"), ("cwd_applied", "u1"), ("sample_rate", "<u2")])
assert DEINT_LINE_DTYPE.itemsize == 24 and BLOCK_DTYPE.itemsize == 72


class DeintSettings(C.Structure):
    _fields_ = [("res_mode", C.c_uint8), ("ignore_crc", C.c_uint8), ("force_ecc_check", C.c_uint8),
                ("en_p_code", C.c_uint8), ("en_q_code", C.c_uint8), ("en_cwd", C.c_uint8), ("_pad", C.c_uint8 * 2)]


def settings(res_mode=1, ignore_crc=0, force=1, p=1, q=1, cwd=0):
    return DeintSettings(res_mode, ignore_crc, force, p, q, cwd)


def run_cpu(lib, prefix, lines, st, n_blocks):
    f = getattr(lib, prefix + "deint_run")
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(DeintSettings), C.c_void_p, C.c_size_t]
    lines = np.ascontiguousarray(lines)
    out = np.zeros(n_blocks, dtype=BLOCK_DTYPE)
    rc = f(lines.ctypes.data, len(lines), C.byref(st), out.ctypes.data, n_blocks)
    return rc, out


def make_lines(w9, frame0=1, line0=1, rng=None, p_bad=0.0, p_corrupt_valid=0.0, p_cwd=0.0, p_corrupt_bad=0.7):
    """Turns generator line words into deinterleaver input records with injected damage:
    p_bad: line has bad CRC (all word flags false) and (p_corrupt_bad) some of its words are really damaged;
    p_corrupt_valid: a word is damaged although the CRC claims the line is fine (CRC collision -> BROKEN blocks);
    p_cwd: the line is marked as repaired by CWD."""
    n = len(w9)
    rng = rng or np.random.default_rng(0)
    lines = np.zeros(n, dtype=DEINT_LINE_DTYPE)
    lines["frame_number"] = frame0 + np.arange(n) // 490
    lines["line_number"] = line0 + (np.arange(n) % 490)
    words = w9[:, :8].astype(np.uint16).copy()
    bad = rng.random(n) < p_bad
    lines["word_crc_ok"] = np.where(bad, 0, 0xFF).astype(np.uint8)
    dmg = bad & (rng.random(n) < p_corrupt_bad)
    for i in np.nonzero(dmg)[0]:
        k = rng.integers(1, 9)
        for s in rng.choice(8, size=k, replace=False):
            words[i, s] ^= np.uint16(rng.integers(1, 1 << 14))
    sneaky = (~bad) & (rng.random(n) < p_corrupt_valid)
    for i in np.nonzero(sneaky)[0]:
        words[i, rng.integers(0, 8)] ^= np.uint16(rng.integers(1, 1 << 14))
    lines["words"] = words
    flags = np.full(n, 2, dtype=np.uint8)                      # SDV_DL_COORDS_BW_OK
    flags[rng.random(n) < 0.1] = 0
    cwd = rng.random(n) < p_cwd
    flags[cwd] |= 1
    lines["flags"] = flags
    # a line can only be "fixed by CWD" if at least one of its words failed the CRC (stc007line.cpp:628-641)
    crc = lines["word_crc_ok"]
    crc[cwd & (crc == 0xFF)] = 0xFE
    lines["word_crc_ok"] = crc
    return lines
