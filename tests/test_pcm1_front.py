"""PCM-1 front half: the oracle's restatement of Binarizer::processLine with a PCM1Line output
(oracle/bin_pcm1.c: black/white search, marker-less coordinate search over the 25 x 25 grid with CRC voting, Bit Picker, header
detection) against the real reference - live when oracle/_ref is built, and through the committed fixtures
(tests/golden/pcm1front_*.npz, made by make_golden_pcm1_front.py) everywhere.  No HIP kernel for this row yet."""
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
    """Random geometry, levels, noise and cut-off; every mode but MODE_INSANE (whose reference level sweep is not restated)."""
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


def test_mode_insane_is_reported_unsupported_and_short_lines_rejected(oracle_lib):
    from sdvpcmdecoder_amd import synth
    luma, _ = synth.pcm1_random_lines(2, seed=7)
    _, rets, _ = pf.run_lines(oracle_lib, "orc_bin1_", luma, mode=3, feedback="none")
    assert (rets == pf.RET_UNSUPPORTED).all()
    _, rets, _ = pf.run_lines(oracle_lib, "orc_bin1_", luma[:, :80], mode=1, feedback="none")
    assert (rets == 3).all()            # LB_RET_SHORT_LINE
