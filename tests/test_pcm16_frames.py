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


# ---- the kernels on the emulator -----------------------------------------------------------------------------------
def _info(lib, eng):
    """sdv_get_run_info through the shared binding (one argtypes declaration per library: engine_api.bind)."""
    import ctypes as C
    import engine_api
    lib.sdv_get_run_info.argtypes = [C.c_void_p, C.POINTER(engine_api.RunInfo)]
    i = engine_api.RunInfo()
    lib.sdv_get_run_info(eng, C.byref(i))
    return i


# the two heaviest MODE_INSANE scenarios (10-14 s each lane by lane) run on the GPU and oracle-vs-reference only; the other insane_* cases cover the sweep on the emulator
EMU_CASES = sorted(set(pf.CASES) - {"ntsc_full", "insane_smeared", "insane_flag_matters_wide"})


@pytest.mark.parametrize("name", EMU_CASES)
def test_emu_matches_oracle(name, emu_lib, oracle_lib):
    import ctypes as C
    luma, mode, st = pf.make_input(name)
    want, wstats = pf.run_cpu(oracle_lib, "orc_", luma, mode, st)
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    rc, got, stats = pf.run_engine(emu_lib, eng, luma, mode, st)
    info = _info(emu_lib, eng)
    emu_lib.sdv_engine_destroy(eng)
    assert rc == 0 and got.tobytes() == want.tobytes() and stats.tobytes() == wstats.tobytes(), _diff(got, want, stats, wstats)
    if name.startswith(("clean_normal", "clean_fast", "noisy", "dup_lines", "silence", "file_marks", "control_bits")):
        assert info.rounds == 1, "a tape that plays is decoded in one round (every frame's incoming state follows from the prescans)"


@pytest.mark.parametrize("mode,kw", [(2, dict(seed=503, jitter=1, noise_sigma=4.0)), (0, dict(seed=502, p_dropout=0.1, noise_sigma=5.0)),
                                     (0, dict(seed=505)), (1, dict(seed=504, p_dropout=0.15, noise_sigma=5.0))])
def test_emu_stream_in_two_calls(mode, kw, emu_lib, oracle_lib):
    """16 frames, then 16 more as the continuation of the stream: the chain state carries over between the calls, mispredicted
    frames (jitter, dropouts in DRAFT mode) are repaired, and the number of rounds stays far below the number of frames."""
    import ctypes as C
    from sdvpcmdecoder_amd import synth
    luma, _ = synth.pcm16x0_frames(16, height=24, **kw)
    w1, s1, h = pf.run_cpu(oracle_lib, "orc_", luma, mode, {}, keep=True)
    w2, s2 = pf.run_cpu(oracle_lib, "orc_", luma, mode, {}, first_frame_no=17, handle=h)
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    rc1, g1, t1 = pf.run_engine(emu_lib, eng, luma, mode, {})
    i1 = _info(emu_lib, eng)
    rc2, g2, t2 = pf.run_engine(emu_lib, eng, luma, mode, {}, first_frame_no=17, configure=False)
    emu_lib.sdv_engine_destroy(eng)
    assert rc1 == 0 and g1.tobytes() == w1.tobytes() and t1.tobytes() == s1.tobytes(), _diff(g1, w1, t1, s1)
    assert rc2 == 0 and g2.tobytes() == w2.tobytes() and t2.tobytes() == s2.tobytes(), _diff(g2, w2, t2, s2)
    assert i1.rounds <= 5


def test_emu_bad_arguments(emu_lib):
    import ctypes as C
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    luma = np.zeros((1, 8, 200), np.uint8)
    rc, _, _ = pf.run_engine(emu_lib, eng, luma[:, :, :180], 1, {})
    assert rc == 3                                              # SDV_ERR_SHORT_LINE: under 193 px
    f = emu_lib.sdv_pcm16x0_binarize_frames
    recs = np.zeros(27, dtype=pf.BIN16_DTYPE); st = np.zeros(1, dtype=pf.STATS_DTYPE)
    emu_lib.sdv_set_mode(eng, 1)
    assert f(eng, luma.ctypes.data, 200, 1600, 200, 8, 1, 1, 1, recs.ctypes.data, 27, st.ctypes.data, 1, None) == -1      # NEW_FILE needs 28
    assert f(eng, None, 200, 1600, 200, 8, 1, 1, 0, recs.ctypes.data, 27, st.ctypes.data, 1, None) == 1
    emu_lib.sdv_engine_destroy(eng)


# ---- the product on the GPU, through the C-ABI -----------------------------------------------------------------------
def _gpu_run(eng, luma, mode, st, torch, first_frame_no=1, configure=True):
    if configure:           # setFineSettings starts the statistics over (videotodigital.cpp:667-676): once per stream
        eng.setBinarizationMode(mode)
        p = eng.getDefaultFineSettings()
        if "force" in st:
            p.en_force_coords = 1
            p.horiz_start, p.horiz_stop = st["force"]
        if "first_line_dup" in st:
            p.en_first_line_dup = st["first_line_dup"]
        for k, v in st.get("preset", {}).items():
            setattr(p, k, v)
        eng.setFineSettings(p)
        eng.setCheckLineDup(bool(st.get("check_line_dup", 1)))
    d = torch.from_numpy(np.ascontiguousarray(luma)).to("cuda:0")
    lines, stats = eng.pcm16x0_binarize_frames(d, first_frame_no=first_frame_no, new_file=bool(st.get("new_file")), doubled=bool(st.get("doubled")),
                                            end_file=bool(st.get("end_file")))
    torch.cuda.synchronize()
    return lines.cpu().numpy().reshape(-1).view(pf.BIN16_DTYPE), stats.cpu().numpy().reshape(-1).view(pf.STATS_DTYPE)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(pf.CASES))
def test_gpu_matches_oracle(name, oracle_lib):
    import torch
    from sdvpcmdecoder_amd import Engine
    luma, mode, st = pf.make_input(name)
    want, wstats = pf.run_cpu(oracle_lib, "orc_", luma, mode, st)
    eng = Engine(0)
    got, stats = _gpu_run(eng, luma, mode, st, torch)
    eng.close()
    assert got.tobytes() == want.tobytes() and stats.tobytes() == wstats.tobytes(), _diff(got, want, stats, wstats)


@pytest.mark.gpu
@pytest.mark.parametrize("name", pf.GOLDEN)
def test_gpu_matches_golden_from_reference(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    luma, mode, st = pf.make_input(name)
    g = np.load(os.path.join(GOLD, "pcm16frames_" + name + ".npz"))
    assert hashlib.sha256(luma.tobytes()).hexdigest() == str(g["input_sha256"])
    want, wstats = g["recs"].reshape(-1).view(pf.BIN16_DTYPE), g["stats"].reshape(-1).view(pf.STATS_DTYPE)
    eng = Engine(0)
    got, stats = _gpu_run(eng, luma, mode, st, torch)
    eng.close()
    assert got.tobytes() == want.tobytes() and stats.tobytes() == wstats.tobytes(), _diff(got, want, stats, wstats)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gpu_stream_in_pieces_equals_oracle(mode, oracle_lib):
    """A 40-frame tape with jitter and lost lines handed over in three calls of uneven size: the chain state crosses the calls."""
    import torch
    from sdvpcmdecoder_amd import Engine, synth
    luma, _ = synth.pcm16x0_frames(40, seed=520 + mode, height=32, jitter=1, p_dropout=0.04, noise_sigma=5.0)
    want, wstats = pf.run_cpu(oracle_lib, "orc_", luma, mode, dict(new_file=True))
    eng = Engine(0)
    got, gst = [], []
    for a, b in ((0, 1), (1, 23), (23, 40)):
        r, s = _gpu_run(eng, luma[a:b], mode, dict(new_file=(a == 0)), torch, first_frame_no=1 + a, configure=(a == 0))
        got.append(r); gst.append(s)
    eng.close()
    got, gst = np.concatenate(got), np.concatenate(gst)
    assert got.tobytes() == want.tobytes() and gst.tobytes() == wstats.tobytes(), _diff(got, want, gst, wstats)


@pytest.mark.gpu
def test_gpu_full_size_batch_decodes_to_what_was_rendered():
    """BASELINE configs[3] scale: 300 NTSC PCM-16x0 frames in one call; every sub-line reads the words that were rendered, in one round."""
    import torch
    from sdvpcmdecoder_amd import Engine, synth
    n, h = 300, 486
    base, words = synth.pcm16x0_frames(6, seed=730, height=h, noise_sigma=4.0)
    luma = np.tile(base, (n // 6, 1, 1))
    eng = Engine(0)
    _gpu_run(eng, base, 2, {}, torch)                                       # the start of the tape: a cold chain takes two rounds
    got, stats = _gpu_run(eng, luma, 2, {}, torch, first_frame_no=7, configure=False)
    info = eng.run_info()
    eng.close()
    assert info.rounds == 1 and info.frames_launched == n, (info.rounds, info.frames_launched)     # a tape that plays: one round
    per = 3 * h + 3
    recs = got.reshape(n, per)
    w = words.reshape(6, h, 3, 4)
    for f in (0, 1, 149, 299):
        fr = recs[f]
        odd = fr[:3 * (h // 2)].reshape(h // 2, 3)
        even = fr[3 * (h // 2) + 1:3 * h + 1].reshape(h // 2, 3)
        assert (odd["words"] == w[f % 6, 0::2]).all() and (even["words"] == w[f % 6, 1::2]).all()
    assert (stats["lines_pcm_odd"] == h // 2).all() and (stats["lines_bad_odd"] <= 1).all()
