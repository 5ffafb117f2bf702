"""CPU tests of the oracle (oracle/*.c): pinned against the reference's own known-answer vectors
(pcmtester.cpp, line-class headers) and against golden outputs of the real reference."""
import ctypes as C

import numpy as np
import pytest

import golden_cases
import libs
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import synth


def test_crc_kat_stc007(oracle_lib):
    # pcmtester.cpp:73-82: 14-bit words -> CRC 0xB2ED
    w = (C.c_uint16 * 8)(0x2D4B, 0x18EE, 0x152B, 0x3A7F, 0x04AB, 0x301B, 0x22F6, 0x0DD6)
    assert oracle_lib.orc_crc_stc007(w) == 0xB2ED


def test_crc_silent_line(oracle_lib):
    # stc007line.h:120 CRC_SILENT
    w = (C.c_uint16 * 8)(*([0] * 8))
    assert oracle_lib.orc_crc_stc007(w) == 0xA96A


def test_crc_check_value(oracle_lib):
    # pcmline.h:88-97: CRC-16 CCITT-FALSE, check 0x29B1 for "123456789"
    oracle_lib.orc_crc16_bytes.restype = C.c_uint16
    oracle_lib.orc_crc16_bytes.argtypes = [C.c_char_p, C.c_size_t]
    assert oracle_lib.orc_crc16_bytes(b"123456789", 9) == 0x29B1


def test_generator_crc_matches_oracle(oracle_lib):
    rng = np.random.default_rng(5)
    words = rng.integers(0, 1 << 14, size=(64, 8), dtype=np.uint32)
    crc = synth.crc16_words14(words)
    for i in range(64):
        w = (C.c_uint16 * 8)(*[int(x) for x in words[i]])
        assert oracle_lib.orc_crc_stc007(w) == int(crc[i])


def test_ecc_vector_pq_generator():
    # pcmtester.cpp:119-126: a block whose words satisfy P and Q
    a = np.array([[0x3B43, 0x3FDB, 0x3B52, 0x3FDA, 0x3B5F, 0x3FDA]], dtype=np.uint32)
    p, q = synth.pq_words(a)
    assert int(p[0]) == 0x0495 and int(q[0]) == 0x1DB7


@pytest.mark.parametrize("name", list(golden_cases.CASES))
def test_oracle_matches_reference_golden(oracle_lib, name):
    mode, luma, want, want_stats = golden_cases.load(name)
    got, got_stats = oracle_binarize(luma, mode=mode)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()


def test_service_and_empty_lines(oracle_lib):
    o = libs.BinApi(oracle_lib, "orc_")
    for srv in (1, 2, 3, 4, 5):
        ret, rec = o.process_px(None, frame=7, line=9, service=srv)
        assert ret == 0 and rec.service_type == srv and rec.frame_number == 7 and rec.line_number == 9
        assert rec.flags & 64 == 0 and rec.words[8] == 0x5695 and rec.calc_crc == 0
    ret, rec = o.process_px(None, frame=1, line=1, service=0, empty=1)
    assert ret == 0 and rec.flags & 64 == 0 and rec.calc_crc == 0xA96A
    ret, rec = o.process_px(np.zeros(100, np.uint8), frame=1, line=1)     # shorter than 137 bit cells
    assert ret == 3                                                        # LB_RET_SHORT_LINE
    o.close()


@pytest.mark.ref
@pytest.mark.skipif(not libs.ref_available(), reason="real reference not built here")
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_oracle_vs_live_reference_lines(oracle_lib, mode):
    """Per-line Binarizer::processLine: oracle vs the real reference on degraded lines, cold and chained."""
    ref = libs.load_ref()
    for seed, kw in ((1, {}), (2, dict(noise_sigma=14.0, blur=2)), (3, dict(noise_sigma=35.0, blur=3)),
                     (4, dict(black=90, white=120)), (5, dict(black=60, white=66, noise_sigma=2.0)),
                     (6, dict(x0=2, x1=716)), (7, dict(width=1440, x0=24, x1=1416)), (8, dict(width=300, x0=5, x1=295))):
        n = 12 if mode >= 2 else 40
        luma, _ = synth.random_lines(n, seed=seed, **kw)
        for chain in (0, 1):
            o = libs.BinApi(oracle_lib, "orc_")
            r = libs.BinApi(ref, "ref_")
            o.set_mode(mode)
            r.set_mode(mode)
            for i in range(n):
                ro, rr = o.process_px(luma[i], 1, i + 1), r.process_px(luma[i], 1, i + 1)
                assert ro[0] == rr[0] and libs.rec_tuple(ro[1]) == libs.rec_tuple(rr[1]), (seed, chain, i)
                if chain and (rr[1].flags & 64):
                    o.set_good_from_last()
                    r.set_good_from_last()
                elif not chain:
                    o.reset_good()
                    r.reset_good()
            o.close()
            r.close()
