"""PCM-16x0 front half: the oracle's restatement of Binarizer::processLine with a PCM16X0SubLine output
(oracle/bin_pcm16.c: black/white search over the three thirds, marker-less coordinate search over the 21 x 21 grid with the three
parts read at every pair and voted on, Bit Picker on the outer parts, control bit) against the real reference - live when
oracle/_ref is built, and through the committed fixtures (tests/golden/pcm16front_*.npz, made by make_golden_pcm16_front.py)."""
import hashlib
import os

import numpy as np
import pytest

import libs
import pcm16_front_api as pf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


REF_KAT_WORDS, REF_KAT_CRC = (0xD527, 0x9C36, 0x02A5), 0xFB40       # pcmtester.cpp:45-49


def _kat_lines():
    """Video lines whose three sub-lines all carry the reference's CRC test vector (words and CRCC as in PCMTester::testPCM16x0CRCC), clean and
    with a little noise, and the presets of a cold Binarizer for every pass."""
    from sdvpcmdecoder_amd import synth
    w4 = np.tile(np.array(REF_KAT_WORDS + (REF_KAT_CRC,), dtype=np.uint16), (4, 3, 1))
    luma = synth.render_lines(synth.pcm16x0_line_bits(w4, np.ones(4, dtype=np.int64)), width=720, x0=5, x1=714, noise_sigma=0.0)
    luma[2:] = synth.render_lines(synth.pcm16x0_line_bits(w4[2:], np.ones(2, dtype=np.int64)), width=720, x0=5, x1=714, noise_sigma=3.0, rng=np.random.default_rng(7))
    cold = np.zeros(3 * len(luma), dtype=pf.STATE_DTYPE); cold["start"], cold["stop"] = -32768, 32767
    return luma, cold


def _check_kat_records(got):
    """the first pass of every line (nothing preset: it searches the coordinates) reads the vector and calculates the reference's CRC for it"""
    first = got[0::3]
    assert (first["words"][:, :3] == np.array(REF_KAT_WORDS, dtype=np.uint16)).all(), first["words"]
    assert (first["words"][:, 3] == REF_KAT_CRC).all() and (first["calc_crc"] == REF_KAT_CRC).all() and ((first["flags"] & pf.LF_CRC_VALID) != 0).all()


def _diff(a, b, ra, rb):
    for i in range(min(len(a), len(b))):
        if a[i].tobytes() != b[i].tobytes() or ra[i] != rb[i]:
            return f"sub-line {i}:\n  got  {a[i]} ret {ra[i]}\n  want {b[i]} ret {rb[i]}"
    return f"lengths {len(a)} / {len(b)}"


def test_crc_known_answers(oracle_lib):
    """PCM16X0SubLine::calcCRC: CRC-16/CCITT-FALSE over 3 x 16 bit; the silent line's CRC is the constant the reference holds
    (pcm16x0subline.h:104, CRC_SILENT = 0x0E10); check value of the algorithm (pcmline.h:88-97)."""
    import ctypes as C
    f = oracle_lib.orc_pcm16x0_crc
    f.restype = C.c_uint16
    f.argtypes = [C.POINTER(C.c_uint16)]
    assert f((C.c_uint16 * 3)(0, 0, 0)) == 0x0E10
    # the reference's own known answer (PCMTester::testPCM16x0CRCC, pcmtester.cpp:40-56): 0xD527 0x9C36 0x02A5 -> 0xFB40
    assert f((C.c_uint16 * 3)(*REF_KAT_WORDS)) == REF_KAT_CRC
    from sdvpcmdecoder_amd import synth
    assert int(synth.pcm16x0_crc_words(np.array(REF_KAT_WORDS, dtype=np.uint32))) == REF_KAT_CRC
    if libs.ref_available():
        g0 = libs.load_ref().ref_pcm16x0_crc
        g0.restype = C.c_uint16
        g0.argtypes = [C.POINTER(C.c_uint16)]
        assert g0((C.c_uint16 * 3)(*REF_KAT_WORDS)) == REF_KAT_CRC
    rng = np.random.default_rng(1)
    w = rng.integers(0, 1 << 16, size=(64, 3), dtype=np.uint32)
    want = synth.pcm16x0_crc_words(w)
    got = [f((C.c_uint16 * 3)(*[int(x) for x in row])) for row in w]
    assert (np.array(got, dtype=np.uint16) == want).all()
    if libs.ref_available():
        g = libs.load_ref().ref_pcm16x0_crc
        g.restype = C.c_uint16
        g.argtypes = [C.POINTER(C.c_uint16)]
        assert [g((C.c_uint16 * 3)(*[int(x) for x in row])) for row in w] == got


@pytest.mark.parametrize("name", pf.GOLDEN)
def test_oracle_matches_golden(name, oracle_lib):
    luma, run = pf.make_case(name)
    g = np.load(os.path.join(GOLD, "pcm16front_" + name + ".npz"))
    assert hashlib.sha256(luma.tobytes()).hexdigest() == str(g["input_sha256"]), "the seeded input changed: regenerate the fixtures"
    want = g["recs"].reshape(-1).view(pf.BIN16_DTYPE)
    got, rets, scans = pf.run_lines(oracle_lib, "orc_bin16_", luma, **run)
    assert got.tobytes() == want.tobytes() and (rets == g["rets"]).all(), _diff(got, want, rets, g["rets"])
    assert (scans == g["scans"]).all()


@pytest.mark.skipif(not libs.ref_available(), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("name", sorted(pf.CASES))
def test_oracle_matches_live_reference(name, oracle_lib):
    ref = libs.load_ref()
    luma, run = pf.make_case(name)
    want, wrets, wscans = pf.run_lines(ref, "ref_bin16_", luma, **run)
    got, rets, scans = pf.run_lines(oracle_lib, "orc_bin16_", luma, **run)
    assert got.tobytes() == want.tobytes() and (rets == wrets).all(), _diff(got, want, rets, wrets)
    assert (scans == wscans).all()


@pytest.mark.skipif(not libs.ref_available(), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("seed", range(6))
def test_oracle_matches_live_reference_random(seed, oracle_lib):
    """Random geometry, levels, noise and cut-off; every mode but MODE_INSANE (its sweep is covered by the insane_* scenarios; too slow for random runs)."""
    from sdvpcmdecoder_amd import synth
    ref = libs.load_ref()
    rng = np.random.default_rng(950 + seed)
    width = int(rng.choice([640, 704, 720, 768]))
    x0 = int(rng.integers(-6, 9)); x1 = width - int(rng.integers(-6, 9))
    black = int(rng.integers(10, 70)); white = black + int(rng.integers(40, 170))
    luma, _ = synth.pcm16x0_random_lines(6, seed=seed, width=width, x0=x0, x1=x1, black=black, white=min(white, 250),
                                         noise_sigma=float(rng.integers(0, 12)), blur=int(rng.integers(0, 2)), control="random")
    for mode in (0, 1, 2):
        for fb in ("good", "reset"):
            want, wrets, wscans = pf.run_lines(ref, "ref_bin16_", luma, mode=mode, feedback=fb)
            got, rets, scans = pf.run_lines(oracle_lib, "orc_bin16_", luma, mode=mode, feedback=fb)
            assert got.tobytes() == want.tobytes() and (rets == wrets).all(), (mode, fb, _diff(got, want, rets, wrets))
            assert (scans == wscans).all()


def test_clean_lines_decode_to_what_was_rendered(oracle_lib):
    from sdvpcmdecoder_amd import synth
    luma, words = synth.pcm16x0_random_lines(8, seed=5, noise_sigma=2.0, control="random")
    got, rets, _ = pf.run_lines(oracle_lib, "orc_bin16_", luma, mode=1, feedback="good")
    assert (rets == 0).all() and ((got["flags"] & pf.LF_CRC_VALID) != 0).all()
    assert (got["words"].reshape(8, 3, 4) == words).all()
    assert (got["line_part"].reshape(8, 3) == np.arange(3)).all()


def test_short_line_and_insane_mode(oracle_lib):
    luma = np.zeros((1, 150), np.uint8)
    got, rets, _ = pf.run_lines(oracle_lib, "orc_bin16_", luma, mode=1)
    assert (rets == 3).all()                                    # LB_RET_SHORT_LINE: under 193 px
    from sdvpcmdecoder_amd import synth
    luma, words = synth.pcm16x0_random_lines(1, seed=9, black=50, white=100)
    got, rets, _ = pf.run_lines(oracle_lib, "orc_bin16_", luma, mode=3, feedback="none")
    # MODE_INSANE: each of the three passes gets its reference level from the sweep and reads with it
    assert (rets == 0).all() and ((got["flags"] & 1) != 0).all() and ((got["flags"] & 64) != 0).all()


# ---- the per-pass entry (sdv_pcm16x0_binarize_lines): the contract, the kernel source on the emulator, the product on the GPU --------------
import ctypes as C  # noqa: E402


def _same_but_line_number(a, b):
    """the sequential run numbers its lines with gaps where service lines were: everything but the line number"""
    a, b = a.copy(), b.copy()
    a["line_number"] = 0; b["line_number"] = 0
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize("name", sorted(pf.CASES))
def test_states_reproduce_the_sequential_run(name, oracle_lib):
    """What a pass depends on is its pixels, the mode and settings, the line's scan_done mark and what was preset on the Binarizer: preset
    pass by pass with what the sequential run had handed on, every pass comes out as in that run."""
    luma, states, seq, seq_scans, kw = pf.case_states(name, lib=oracle_lib)
    got, scans = pf.run_lines_with_states(oracle_lib, "orc_bin16_", luma, states, **kw)
    assert _same_but_line_number(got, seq), _diff(got, seq, scans, seq_scans)
    assert (scans == seq_scans).all()


@pytest.fixture(scope="module")
def emu(emu_lib):
    emu_lib.sdv_engine_create.restype = C.c_void_p
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    yield emu_lib, eng
    emu_lib.sdv_engine_destroy(eng)


@pytest.mark.parametrize("name", ["clean_fast", "cut_bits_normal", "cut_left_only", "noisy", "control_bits", "forced_coords", "no_bit_picker", "window_moves", "one_bad_part", "garbage",
                                  "flat_and_services", "wide_1440_doubled", "insane_scratch", "insane_one_bad_part", "insane_few_valid"])
def test_emu_lines_match_oracle(name, emu, oracle_lib):
    lib, eng = emu
    luma, states, seq, seq_scans, kw = pf.case_states(name, lib=oracle_lib)
    rc, got, scans = pf.run_engine_lines(lib, eng, luma, states, **kw)
    assert rc == 0
    assert _same_but_line_number(got, seq), _diff(got, seq, scans, seq_scans)
    assert (scans == seq_scans).all()


def test_reference_crc_vector_through_oracle_and_emulator(emu, oracle_lib):
    """0xD527 0x9C36 0x02A5 -> 0xFB40 (pcmtester.cpp:40-56) as pixels: the oracle's Binarizer and the kernel source on the emulator read the words and
    calculate that CRC; the live reference does when it is built."""
    luma, cold = _kat_lines()
    want, _ = pf.run_lines_with_states(oracle_lib, "orc_bin16_", luma, cold, mode=2)
    _check_kat_records(want)
    lib, eng = emu
    rc, got, _ = pf.run_engine_lines(lib, eng, luma, cold, mode=2)
    assert rc == 0 and got.tobytes() == want.tobytes()
    _check_kat_records(got)
    if libs.ref_available():
        ref, _ = pf.run_lines_with_states(libs.load_ref(), "ref_bin16_", luma, cold, mode=2)
        assert ref.tobytes() == want.tobytes()


def test_emu_lines_argument_checks(emu):
    lib, eng = emu
    luma = np.zeros((2, 720), np.uint8)
    assert pf.run_engine_lines(lib, eng, luma[:, :150])[0] == 3          # SDV_ERR_SHORT_LINE: under 193 px
    rc, got, _ = pf.run_engine_lines(lib, eng, luma)                     # nothing preset, nothing to read
    assert rc == 0 and ((got["flags"] & pf.LF_CRC_VALID) == 0).all() and (got["line_part"].reshape(2, 3) == np.arange(3)).all()


def _gpu_lines(luma, states, mode=1, coord_search=True, preset=None, doubled=False):
    import torch
    torch.zeros(1, device="cuda:0")
    from sdvpcmdecoder_amd import Engine
    from sdvpcmdecoder_amd.engine import BinPreset
    eng = Engine(0)
    eng.setBinarizationMode(mode)
    eng.setFineSettings(BinPreset.from_buffer_copy(bytes(preset if preset is not None else libs.default_preset())))
    d_luma = torch.from_numpy(np.ascontiguousarray(luma)).cuda()
    d_st = torch.from_numpy(np.ascontiguousarray(states).view(np.uint8).reshape(len(states), 10)).cuda()
    out, scans = eng.pcm16x0_binarize_lines(d_luma, d_st, frame_number=1, first_line=1, line_step=1, doubled=doubled, coord_search=coord_search, with_scan_done=True)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(-1).view(pf.BIN16_DTYPE), scans.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(pf.CASES))
def test_gpu_lines_match_oracle(name, oracle_lib):
    luma, states, seq, seq_scans, kw = pf.case_states(name, lib=oracle_lib)
    got, scans = _gpu_lines(luma, states, **kw)
    assert _same_but_line_number(got, seq), _diff(got, seq, scans, seq_scans)
    assert (scans == seq_scans).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", pf.GOLDEN)
def test_gpu_lines_match_golden_from_reference(name):
    """The fixtures hold the real Binarizer's sequential run (three passes per line on one object); the presets of every pass are rebuilt
    from the fixture's own records - no oracle in between."""
    g = np.load(os.path.join(GOLD, "pcm16front_" + name + ".npz"))
    want = g["recs"].reshape(-1).view(pf.BIN16_DTYPE)
    luma, states, seq, _, kw = pf.case_states(name, recs=want)
    assert hashlib.sha256(pf.make_case(name)[0].tobytes()).hexdigest() == str(g["input_sha256"])
    got, scans = _gpu_lines(luma, states, **kw)
    assert _same_but_line_number(got, seq), _diff(got, seq, scans, scans)
    rows, at = pf.data_rows(len(pf.make_case(name)[0]), pf.make_case(name)[1])
    idx = (at[:, None] + np.arange(3)[None, :]).reshape(-1)
    assert (scans == g["scans"][idx]).all()


@pytest.mark.gpu
def test_gpu_reference_crc_vector(oracle_lib):
    """0xD527 0x9C36 0x02A5 -> 0xFB40 (pcmtester.cpp:40-56) as pixels through sdv_pcm16x0_binarize_lines on the GPU"""
    luma, cold = _kat_lines()
    got, _ = _gpu_lines(luma, cold, mode=2)
    _check_kat_records(got)
    want, _ = pf.run_lines_with_states(oracle_lib, "orc_bin16_", luma, cold, mode=2)
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
def test_gpu_field_of_lines_three_passes(oracle_lib):
    """245 lines at once: every pass from scratch (the coordinate search on the first pass of every line), then every pass preset from a
    decoded neighbour (a tape that plays)."""
    from sdvpcmdecoder_amd import synth
    luma, words = synth.pcm16x0_random_lines(245, seed=21, x0=5, x1=713, noise_sigma=4.0, control="random")
    cold = np.zeros(3 * 245, dtype=pf.STATE_DTYPE); cold["start"], cold["stop"] = -32768, 32767
    got, scans = _gpu_lines(luma, cold, mode=2)
    want, wscans = pf.run_lines_with_states(oracle_lib, "orc_bin16_", luma, cold, mode=2)
    assert got.tobytes() == want.tobytes() and (scans == wscans).all()
    # nothing handed on: the first pass searches the coordinates and reads; the line's search has run then, the other passes have nothing to read with
    assert ((got["flags"][0::3] & pf.LF_CRC_VALID) != 0).all() and (got["words"].reshape(245, 3, 4)[:, 0] == words[:, 0]).all() and (scans == 1).all()
    warm = pf.states_from_records(np.concatenate([got[:1], got[:-1]]), mode=2)
    warm[0] = warm[1]
    got2, scans2 = _gpu_lines(luma, warm, mode=2)
    want2, wscans2 = pf.run_lines_with_states(oracle_lib, "orc_bin16_", luma, warm, mode=2)
    assert got2.tobytes() == want2.tobytes() and (scans2 == wscans2).all()
    assert ((got2["flags"] & pf.LF_BY_EXT_TUNE) != 0).sum() >= 3 * 240 and (got2["words"].reshape(245, 3, 4) == words).all()
