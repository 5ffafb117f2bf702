"""PCM-16x0 frame driver (SURVEY section 8 row a11 for PCM-16x0): VideoToDigital::doBinarize + prescanCoordinates with
PCM16X0SubLine output, three passes per video line.
  oracle/v2d_p16.c     vs  the real reference's VideoToDigital worker (live when oracle/_ref is built) and the committed
                           fixtures tests/golden/pcm16frames_*.npz (made by make_golden_pcm16_frames.py)
  HIP kernel source    vs  the oracle, on the SIMT emulator (CPU) and through the C-ABI on the GPU (-m gpu)."""
import hashlib
import os

import numpy as np
import pytest

import libs
import pcm16_frames_api as pf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _diff(a, b, sa, sb):
    for i in range(min(len(a), len(b))):
        if a[i].tobytes() != b[i].tobytes():
            return f"record {i}:\n  got  {a[i]}\n  want {b[i]}"
    for i in range(min(len(sa), len(sb))):
        if sa[i].tobytes() != sb[i].tobytes():
            return f"frame descriptor {i}:\n  got  {sa[i]}\n  want {sb[i]}"
    return f"lengths {len(a)}/{len(b)} {len(sa)}/{len(sb)}"


@pytest.mark.parametrize("name", pf.GOLDEN)
def test_oracle_matches_golden(name, oracle_lib):
    luma, mode, st = pf.make_input(name)
    g = np.load(os.path.join(GOLD, "pcm16frames_" + name + ".npz"))
    assert hashlib.sha256(luma.tobytes()).hexdigest() == str(g["input_sha256"]), "the seeded input changed: regenerate the fixtures"
    want, wstats = g["recs"].reshape(-1).view(pf.BIN16_DTYPE), g["stats"].reshape(-1).view(pf.STATS_DTYPE)
    got, stats = pf.run_cpu(oracle_lib, "orc_", luma, mode, st)
    assert got.tobytes() == want.tobytes() and stats.tobytes() == wstats.tobytes(), _diff(got, want, stats, wstats)


@pytest.mark.skipif(not libs.ref_available(), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("name", sorted(pf.CASES))
def test_oracle_matches_live_reference(name, oracle_lib):
    ref = libs.load_ref()
    luma, mode, st = pf.make_input(name)
    want, wstats = pf.run_cpu(ref, "ref_", luma, mode, st)
    got, stats = pf.run_cpu(oracle_lib, "orc_", luma, mode, st)
    assert got.tobytes() == want.tobytes() and stats.tobytes() == wstats.tobytes(), _diff(got, want, stats, wstats)


def test_clean_frames_decode_to_what_was_rendered(oracle_lib):
    from sdvpcmdecoder_amd import synth
    h = 48
    luma, words = synth.pcm16x0_frames(2, seed=7, height=h, noise_sigma=2.0)
    got, stats = pf.run_cpu(oracle_lib, "orc_", luma, 2, {})
    per = 3 * h + 3
    for f in range(2):
        fr = got[f * per:(f + 1) * per]
        odd = fr[:3 * (h // 2)].reshape(h // 2, 3)
        even = fr[3 * (h // 2) + 1:3 * h + 1].reshape(h // 2, 3)
        assert (odd["words"] == words[f * h:(f + 1) * h:2]).all() and (even["words"] == words[f * h + 1:(f + 1) * h:2]).all()
        assert (odd["line_part"] == np.arange(3)).all() and (odd["queue_order"].reshape(-1) == np.arange(3 * (h // 2))).all()
    assert (stats["lines_odd"] == 245).all() and (stats["lines_pcm_odd"] == h // 2).all()
