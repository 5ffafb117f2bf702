"""Deinterleaver + P/Q error correction: oracle pinned by the reference's own test vectors (pcmtester.cpp) and
compared block-for-block with the real reference on randomly damaged streams."""
import ctypes as C

import numpy as np
import pytest

import deint_api as da
import libs
from sdvpcmdecoder_amd import synth

COMBOS = [(1, 1, 0), (1, 1, 1), (1, 0, 0), (1, 0, 1), (0, 0, 0)]      # (P, Q, CWD) the setters allow


def test_pcmtester_ecc_vector(oracle_lib):
    """pcmtester.cpp:119-126: L0..R2 + P 0x0495 + Q 0x1DB7 is a consistent block."""
    oracle_lib.orc_q_code.restype = C.c_uint16
    oracle_lib.orc_p_code.restype = C.c_uint16
    w = (C.c_uint16 * 6)(0x3B43, 0x3FDB, 0x3B52, 0x3FDA, 0x3B5F, 0x3FDA)
    assert oracle_lib.orc_p_code(w) == 0x0495
    assert oracle_lib.orc_q_code(w) == 0x1DB7


def pcmtester_window():
    """The reference's ECC test buffer (pcmtester.cpp:110-180): 113 copies of one valid line."""
    words = np.array([0x3B43, 0x3FDB, 0x3B52, 0x3FDA, 0x3B5F, 0x3FDA, 0x0495, 0x1DB7], dtype=np.uint16)
    lines = np.zeros(113, dtype=da.DEINT_LINE_DTYPE)
    lines["words"] = words
    lines["word_crc_ok"] = 0xFF
    lines["flags"] = 2
    lines["frame_number"] = 1
    lines["line_number"] = np.arange(113) + 1
    return lines, words


@pytest.mark.parametrize("kill", [1, 2, "any"])
def test_pcmtester_ecc_property(oracle_lib, kill):
    """The reference's own randomized ECC test (pcmtester.cpp:196-369), 2048 runs per mode: corrupt words at line
    offsets idx*16, invert those lines' CRC; 1-2 errors must be repaired to the original words, more must fail."""
    rng = np.random.default_rng(7 if kill == "any" else kill)
    st = da.settings(res_mode=0, force=1, p=1, q=1, cwd=0)        # RES_MODE_14BIT, forced check
    for _ in range(2048):
        lines, words = pcmtester_window()
        n_kill = int(rng.integers(0, 9)) if kill == "any" else kill
        idxs = rng.choice(8, size=n_kill, replace=False)
        for k in idxs:
            lines["words"][k * 16, k] ^= np.uint16(rng.integers(1, 1 << 14))
            lines["word_crc_ok"][k * 16] = 0
        rc, out = da.run_cpu(oracle_lib, "orc_", lines, st, 1)
        assert rc == 3
        b = out[0]
        audio_killed = sum(1 for k in idxs if k < 6)
        valid = (b["word_valid"] & 0x3F) == 0x3F
        if n_kill <= 2:
            assert valid and (b["words"] == words).all(), (idxs, b)
        elif audio_killed > 0:
            assert not valid, (idxs, b)


@pytest.mark.ref
@pytest.mark.skipif(not libs.ref_available(), reason="real reference not built here")
@pytest.mark.parametrize("res_mode", [0, 1, 2, 3])
def test_deint_oracle_vs_live_reference(oracle_lib, res_mode):
    ref = libs.load_ref()
    rng = np.random.default_rng(100 + res_mode)
    n = 113 + 400
    audio14 = rng.integers(0, 1 << 14, size=(n, 6), dtype=np.uint32)
    audio16 = rng.integers(0, 1 << 16, size=(n, 6), dtype=np.uint32)
    audio14[200:230] = 0                                            # a silent stretch
    streams = [synth.interleave_stream(audio14), synth.interleave_stream_f1(audio16)]
    for w9 in streams:
        for (p_bad, p_sneak, p_cwd) in ((0.0, 0.0, 0.0), (0.02, 0.0, 0.0), (0.08, 0.01, 0.02), (0.3, 0.02, 0.05)):
            lines = da.make_lines(w9, rng=rng, p_bad=p_bad, p_corrupt_valid=p_sneak, p_cwd=p_cwd)
            for (p, q, cwd) in COMBOS:
                for force in (1, 0):
                    for ign in (0, 1):
                        st = da.settings(res_mode=res_mode, ignore_crc=ign, force=force, p=p, q=q, cwd=cwd)
                        rc_o, o = da.run_cpu(oracle_lib, "orc_", lines, st, 400)
                        rc_r, r = da.run_cpu(ref, "ref_", lines, st, 400)
                        assert rc_o == rc_r == 3
                        bad = np.nonzero(o.view(np.uint8).reshape(400, 72) != r.view(np.uint8).reshape(400, 72))[0]
                        assert len(bad) == 0, (res_mode, p_bad, (p, q, cwd), force, ign, o[bad[0]], r[bad[0]])


def test_deint_short_buffer(oracle_lib):
    lines, _ = pcmtester_window()
    rc, _ = da.run_cpu(oracle_lib, "orc_", lines[:112], da.settings(), 1)
    assert rc == 2          # DI_RET_NO_DATA
