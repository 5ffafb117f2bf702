"""PCM-1 frame driver (SURVEY section 8 row a11 for PCM-1): VideoToDigital::doBinarize + prescanCoordinates with PCM1Line output.
  oracle/v2d_p1.c      vs  the real reference's VideoToDigital worker (live when oracle/_ref is built) and the committed
                           fixtures tests/golden/pcm1frames_*.npz (made by make_golden_pcm1_frames.py)
  HIP kernel source    vs  the oracle, on the SIMT emulator (CPU) and through the C-ABI on the GPU (-m gpu)."""
import hashlib
import os

import numpy as np
import pytest

import libs
import pcm1_frames_api as pf

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
    g = np.load(os.path.join(GOLD, "pcm1frames_" + name + ".npz"))
    assert hashlib.sha256(luma.tobytes()).hexdigest() == str(g["input_sha256"]), "the seeded input changed: regenerate the fixtures"
    want, wstats = g["recs"].reshape(-1).view(pf.BIN1_DTYPE), g["stats"].reshape(-1).view(pf.STATS_DTYPE)
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
    h = 64
    luma, words = synth.pcm1_frames(2, seed=7, height=h, noise_sigma=2.0)
    got, stats = pf.run_cpu(oracle_lib, "orc_", luma, 2, {})
    rows = got[got["service_type"] == 0]
    # rows come field by field; the first PCM line of each field is the Header (a service line)
    assert len(rows) == 2 * (h - 2)
    for f in range(2):
        fr = got[f * (h + 3):(f + 1) * (h + 3)]
        odd, even = fr[1:h // 2], fr[h // 2 + 2:h + 1]
        assert (odd["words"] == words[f * h + 2:(f + 1) * h:2]).all() and (even["words"] == words[f * h + 3:(f + 1) * h:2]).all()
        assert fr[0]["service_type"] == 6 and fr[h // 2 + 1]["service_type"] == 6        # SRVLINE_HEADER_LINE
    assert (stats["lines_odd"] == 245).all() and (stats["lines_pcm_odd"] == h // 2 - 1).all() and (stats["lines_bad_odd"] == 0).all()
