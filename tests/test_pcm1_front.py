"""PCM-1 front half: the oracle's restatement of Binarizer::processLine with a PCM1Line output
(oracle/bin_pcm1.c: black/white search, marker-less coordinate search over the 25 x 25 grid with CRC voting, Bit Picker, header
detection) against the real reference - live when oracle/_ref is built, and through the committed fixtures
(tests/golden/pcm1front_*.npz, made by make_golden_pcm1_front.py) everywhere; the HIP line kernel (sdv_pcm1_binarize_lines) is compared with the same fixtures in the gpu-marked tests."""
import hashlib
import os

import numpy as np
import pytest

import libs
import pcm1_front_api as pf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _diff(a, b, ra, rb):
    for i in range(len(a)):
        if a[i].tobytes() != b[i].tobytes() or ra[i] != rb[i]:
            return f"line {i}:\n  got  {a[i]} ret {ra[i]}\n  want {b[i]} ret {rb[i]}"
    return "equal"


@pytest.mark.parametrize("name", pf.GOLDEN)
def test_oracle_matches_golden(name, oracle_lib):
    luma, run = pf.make_case(name)
    g = np.load(os.path.join(GOLD, "pcm1front_" + name + ".npz"))
    assert hashlib.sha256(luma.tobytes()).hexdigest() == str(g["input_sha256"]), "the seeded input changed: regenerate the fixtures"
    want = g["recs"].reshape(-1).view(pf.BIN1_DTYPE)
    got, rets, scans = pf.run_lines(oracle_lib, "orc_bin1_", luma, **run)
    assert got.tobytes() == want.tobytes() and (rets == g["rets"]).all(), _diff(got, want, rets, g["rets"])
    assert (scans == g["scans"]).all()


@pytest.mark.skipif(not libs.ref_available(), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("name", sorted(pf.CASES))
def test_oracle_matches_live_reference(name, oracle_lib):
    ref = libs.load_ref()
    luma, run = pf.make_case(name)
    want, wrets, wscans = pf.run_lines(ref, "ref_bin1_", luma, **run)
    got, rets, scans = pf.run_lines(oracle_lib, "orc_bin1_", luma, **run)
    assert got.tobytes() == want.tobytes() and (rets == wrets).all(), _diff(got, want, rets, wrets)
    assert (scans == wscans).all()


@pytest.mark.skipif(not libs.ref_available(), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("seed", range(6))
def test_oracle_matches_live_reference_random(seed, oracle_lib):
    """Random geometry, levels, noise and cut-off; every mode but MODE_INSANE (its sweep is covered by the insane_* scenarios; too slow for random runs)."""
    from sdvpcmdecoder_amd import synth
    ref = libs.load_ref()
    rng = np.random.default_rng(900 + seed)
    width = int(rng.choice([640, 704, 720, 768]))
    x0 = int(rng.integers(-14, 16)); x1 = width - int(rng.integers(-12, 16))
    black = int(rng.integers(10, 70)); white = black + int(rng.integers(40, 170))
    luma, _ = synth.pcm1_random_lines(10, seed=seed, width=width, x0=x0, x1=x1, black=black, white=min(white, 250),
                                      noise_sigma=float(rng.integers(0, 14)), blur=int(rng.integers(0, 3)), header_every=int(rng.choice([0, 4])))
    for mode in (0, 1, 2):
        for fb in ("good", "reset"):
            want, wrets, wscans = pf.run_lines(ref, "ref_bin1_", luma, mode=mode, feedback=fb)
            got, rets, scans = pf.run_lines(oracle_lib, "orc_bin1_", luma, mode=mode, feedback=fb)
            assert got.tobytes() == want.tobytes() and (rets == wrets).all(), (mode, fb, _diff(got, want, rets, wrets))
            assert (scans == wscans).all()


def test_clean_lines_decode_to_what_was_rendered(oracle_lib):
    from sdvpcmdecoder_amd import synth
    luma, words = synth.pcm1_random_lines(12, seed=5, noise_sigma=2.0)
    got, rets, _ = pf.run_lines(oracle_lib, "orc_bin1_", luma, mode=1, feedback="good")
    assert (rets == 0).all() and ((got["flags"] & pf.LF_CRC_VALID) != 0).all()
    assert (got["words"] == words).all()
    assert (got["calc_crc"] == words[:, 6]).all()


def test_cut_off_bits_are_picked(oracle_lib):
    """The picture starts 9 px into the first word and ends inside the CRC: the Bit Picker completes both from the CRC."""
    from sdvpcmdecoder_amd import synth
    luma, words = synth.pcm1_random_lines(8, seed=6, x0=-9, x1=726, noise_sigma=2.0)
    got, _, _ = pf.run_lines(oracle_lib, "orc_bin1_", luma, mode=2, feedback="good")
    ok = (got["flags"] & pf.LF_CRC_VALID) != 0
    assert ok.all() and (got["picked_bits_left"] > 0).all()
    assert (got["words"] == words).all()


def test_short_lines_rejected(oracle_lib):
    from sdvpcmdecoder_amd import synth
    luma, _ = synth.pcm1_random_lines(2, seed=7)
    _, rets, _ = pf.run_lines(oracle_lib, "orc_bin1_", luma[:, :80], mode=1, feedback="none")
    assert (rets == 3).all()            # LB_RET_SHORT_LINE


def test_mode_insane_sweeps_the_reference_level(oracle_lib):
    """MODE_INSANE: a line that starts from nothing gets its reference level from the sweep (and reads with it)."""
    from sdvpcmdecoder_amd import synth
    luma, words = synth.pcm1_random_lines(2, seed=7, black=50, white=100)
    got, rets, _ = pf.run_lines(oracle_lib, "orc_bin1_", luma, mode=3, feedback="none")
    assert (rets == 0).all() and ((got["flags"] & 1) != 0).all() and ((got["flags"] & pf.LF_CRC_VALID) != 0).all()
    assert (got["words"][:, :6] == words[:, :6]).all()


# ---- the HIP kernel source on the CPU emulator (tests/emu) against the oracle ----------------------------------------------------
import ctypes as C


@pytest.fixture(scope="module")
def emu(emu_lib):
    emu_lib.sdv_engine_create.restype = C.c_void_p
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    yield emu_lib, eng
    emu_lib.sdv_engine_destroy(eng)


def _case_states(name, oracle_lib):
    """Rows, run options and the per-line presets the sequential run of the case would have had (from the oracle's own records)."""
    luma, run = pf.make_case(name)
    keep = np.ones(len(luma), dtype=bool)
    if "services" in run:
        keep &= run["services"] == 0
    if "empty" in run:
        keep &= run["empty"] == 0
    seq, _, _ = pf.run_lines(oracle_lib, "orc_bin1_", luma, **run)
    states = pf.states_for_run(seq, run)
    kw = dict(mode=run["mode"], coord_search=run.get("coord_search", True), preset=run["preset"], doubled=run.get("doubled", False))
    return luma[keep], states[keep], seq[keep], kw


@pytest.mark.parametrize("name", ["clean_fast", "cut_bits_draft", "cut_left_only", "noisy_header", "forced_coords", "no_bit_picker", "window_moves", "garbage"]
                         + sorted(n for n in pf.CASES if n.startswith("insane")))
def test_emu_matches_oracle(name, emu, oracle_lib):
    lib, eng = emu
    luma, states, seq, kw = _case_states(name, oracle_lib)
    want = pf.run_lines_with_states(oracle_lib, "orc_bin1_", luma, states, **kw)
    rc, got = pf.run_engine_lines(lib, eng, luma, states, **kw)
    assert rc == 0
    assert got.tobytes() == want.tobytes(), _diff(got, want, np.zeros(len(got)), np.zeros(len(got)))
    # and the batch with per-line presets is the sequential run of the case: same line numbers as the sequential run have gaps
    # where service lines were, so compare everything but the line number
    a, b = got.copy(), seq.copy()
    a["line_number"] = 0; b["line_number"] = 0
    assert a.tobytes() == b.tobytes()


def test_emu_argument_checks(emu):
    lib, eng = emu
    luma = np.zeros((2, 720), np.uint8)
    rc, _ = pf.run_engine_lines(lib, eng, luma[:, :80])
    assert rc == 3                      # SDV_ERR_SHORT_LINE


# ---- the product on the GPU, through the C-ABI -------------------------------------------------------------------------------------
def _gpu_engine():
    """torch first: it has to bring up the HIP runtime it ships before the library's own first HIP call."""
    import torch
    torch.zeros(1, device="cuda:0")
    from sdvpcmdecoder_amd import Engine
    return Engine(0)


def _gpu_run(eng, luma, states, mode=1, coord_search=True, preset=None, doubled=False):
    import torch
    from sdvpcmdecoder_amd.engine import BinPreset
    eng.setBinarizationMode(mode)
    eng.setFineSettings(BinPreset.from_buffer_copy(bytes(preset if preset is not None else libs.default_preset())))
    d_luma = torch.from_numpy(np.ascontiguousarray(luma)).cuda()
    d_st = torch.from_numpy(np.ascontiguousarray(states).view(np.uint8).reshape(len(states), 10)).cuda()
    out = eng.pcm1_binarize_lines(d_luma, d_st, frame_number=1, first_line=1, line_step=1, doubled=doubled, coord_search=coord_search)
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(-1).view(pf.BIN1_DTYPE)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(pf.CASES))
def test_gpu_matches_oracle(name, oracle_lib):
    luma, states, seq, kw = _case_states(name, oracle_lib)
    want = pf.run_lines_with_states(oracle_lib, "orc_bin1_", luma, states, **kw)
    got = _gpu_run(_gpu_engine(), luma, states, **kw)
    assert got.tobytes() == want.tobytes(), _diff(got, want, np.zeros(len(got)), np.zeros(len(got)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", pf.GOLDEN)
def test_gpu_matches_golden_from_reference(name, oracle_lib):
    """The fixtures hold the reference's sequential run; the per-line presets are rebuilt from the fixture's own records."""
    luma, run = pf.make_case(name)
    g = np.load(os.path.join(GOLD, "pcm1front_" + name + ".npz"))
    want = g["recs"].reshape(-1).view(pf.BIN1_DTYPE)
    keep = np.ones(len(luma), dtype=bool)
    if "services" in run:
        keep &= run["services"] == 0
    if "empty" in run:
        keep &= run["empty"] == 0
    states = pf.states_for_run(want, run)
    got = _gpu_run(_gpu_engine(), luma[keep], states[keep], mode=run["mode"], coord_search=run.get("coord_search", True), preset=run["preset"],
                   doubled=run.get("doubled", False))
    a, b = got.copy(), want[keep].copy()
    a["line_number"] = 0; b["line_number"] = 0
    assert a.tobytes() == b.tobytes(), _diff(a, b, np.zeros(len(a)), np.zeros(len(a)))


@pytest.mark.gpu
def test_gpu_field_of_lines_cold_and_warm(oracle_lib):
    """245 lines at once: every line from scratch (the coordinate search on every line), then every line preset from a decoded
    neighbour (the steady state of a tape that plays)."""
    from sdvpcmdecoder_amd import synth
    luma, words = synth.pcm1_random_lines(245, seed=11, x0=5, x1=713, noise_sigma=4.0)
    cold = np.zeros(245, dtype=pf.STATE_DTYPE); cold["start"], cold["stop"] = -32768, 32767
    eng = _gpu_engine()
    got = _gpu_run(eng, luma, cold, mode=2)
    want = pf.run_lines_with_states(oracle_lib, "orc_bin1_", luma, cold, mode=2)
    assert got.tobytes() == want.tobytes()
    assert ((got["flags"] & pf.LF_CRC_VALID) != 0).all() and (got["words"] == words).all()
    warm = pf.states_from_records(np.concatenate([got[:1], got[:-1]]))
    warm[0] = warm[1]
    got2 = _gpu_run(eng, luma, warm, mode=2)
    want2 = pf.run_lines_with_states(oracle_lib, "orc_bin1_", luma, warm, mode=2)
    assert got2.tobytes() == want2.tobytes()
    assert ((got2["flags"] & pf.LF_BY_EXT_TUNE) != 0).sum() >= 240


@pytest.mark.gpu
def test_gpu_long_hand_over_list(oracle_lib):
    """More lines than the full kernel has workgroups, none of them decodable from its presets: the lean kernel hands all of them on,
    the full kernel works the list off (its entries behind the grid through the shared counter) - same records as with no presets at
    all, and as the oracle's on a sample."""
    import torch
    from sdvpcmdecoder_amd import synth
    luma, _ = synth.pcm1_random_lines(245, seed=12, x0=5, x1=713, noise_sigma=4.0)
    big = np.tile(luma, (41, 1))                        # 10 045 lines
    wrong = np.zeros(len(big), dtype=pf.STATE_DTYPE)
    wrong["black"], wrong["white"], wrong["ref"], wrong["start"], wrong["stop"] = 30, 200, 115, 60, 650       # presets that read nothing
    eng = _gpu_engine()
    eng.setBinarizationMode(1)
    d_luma = torch.from_numpy(big).cuda()
    a = eng.pcm1_binarize_lines(d_luma, torch.from_numpy(wrong.view(np.uint8).reshape(len(wrong), 10)).cuda()).cpu().numpy().reshape(-1).view(pf.BIN1_DTYPE)
    want = pf.run_lines_with_states(oracle_lib, "orc_bin1_", big[:300], wrong[:300], mode=1)
    assert a[:300].tobytes() == want.tobytes()
    assert ((a["flags"] & pf.LF_CRC_VALID) != 0).all() and ((a["flags"] & pf.LF_BY_EXT_TUNE) == 0).all()
    rest = a[245:].copy(); first = np.tile(a[:245], 40)
    rest["line_number"] = 0; first["line_number"] = 0
    assert rest.tobytes() == first.tobytes()
