"""PCM-16x0 back half (PCM16X0Deinterleaver + PCM16X0DataStitcher -> PCMSamplePair).
  oracle (oracle/pcm16.c)  vs  golden fixtures of the real reference and - when the reference build is loadable - the real
                               PCM16X0Deinterleaver / PCM16X0DataStitcher run live on every scenario;
  HIP kernel source        vs  the oracle, on the SIMT emulator (CPU) and through the C-ABI on the GPU (-m gpu)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import libs
import pcm16_api as p16
from sdvpcmdecoder_amd import synth
from stitch_api import PAIR_DTYPE

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(pairs, frames, want_p, want_f):
    return len(pairs) == len(want_p) and pairs.tobytes() == want_p.tobytes() and len(frames) == len(want_f) and frames.tobytes() == want_f.tobytes()


def _diff(pairs, frames, want_p, want_f):
    out = [f"pairs {len(pairs)} vs {len(want_p)}, frames {len(frames)} vs {len(want_f)}"]
    for i in range(min(len(frames), len(want_f))):
        if frames[i].tobytes() != want_f[i].tobytes():
            out.append(f" frame {i}: " + str([(n, frames[i][n], want_f[i][n]) for n in p16.FRASM16_DTYPE.names if np.any(frames[i][n] != want_f[i][n])]))
    n = min(len(pairs), len(want_p))
    d = np.nonzero((pairs[:n].view(np.uint8).reshape(n, 12) != want_p[:n].view(np.uint8).reshape(n, 12)).any(axis=1))[0]
    out.append(f" {len(d)} pairs differ, first at {d[:5]}")
    return "\n".join(out)


def _oracle(name):
    recs, st = p16.make_input(name)
    pairs, frames = p16.run_cpu(libs.load_oracle(), "orc_", recs, st)
    return recs, st, pairs, frames


# ---- the oracle is pinned ------------------------------------------------------------------------------------------
def test_blocks_match_golden(oracle_lib):
    """PCM16X0Deinterleaver::processBlock: data blocks of damaged SI and EI queues under every combination of forced check,
    P-code correction and ignore-CRC, as the real reference assembled and corrected them (tests/golden/pcm16_blocks.npz)."""
    z = np.load(os.path.join(GOLD, "pcm16_blocks.npz"))
    inputs = p16.block_inputs()
    assert set(z.files) == set(inputs)
    states = np.zeros(3, dtype=np.int64)
    for key, (recs, kw) in inputs.items():
        got = p16.run_blocks(oracle_lib, "orc_", recs, **kw)
        want = np.ascontiguousarray(z[key]).view(p16.BLOCK16_DTYPE).reshape(-1)
        assert got.tobytes() == want.tobytes(), key
        states += np.bincount(got["audio_state"].ravel(), minlength=3)
    assert (states > 100).all()             # original, fixed by P and BROKEN blocks all occur


@pytest.mark.parametrize("name", p16.GOLDEN)
def test_oracle_matches_golden(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "pcm16_" + name + ".npz"))
    recs, st, pairs, frames = _oracle(name)
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"]), "regenerated input stream differs from the fixture's"
    assert bytes(st) == z["settings"].tobytes()
    want_p = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    want_f = np.ascontiguousarray(z["frames"]).view(p16.FRASM16_DTYPE).reshape(-1)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.ref
@pytest.mark.parametrize("name", list(p16.CASES))
def test_oracle_matches_live_reference(name, oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    recs, st, pairs, frames = _oracle(name)
    rp, rf = p16.run_cpu(libs.load_ref(), "ref_", recs, st)
    assert _same(pairs, frames, rp, rf), _diff(pairs, frames, rp, rf)


@pytest.mark.ref
def test_blocks_match_live_reference(oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    for key, (recs, kw) in p16.block_inputs().items():
        assert p16.run_blocks(oracle_lib, "orc_", recs, **kw).tobytes() == p16.run_blocks(libs.load_ref(), "ref_", recs, **kw).tobytes(), key


@pytest.mark.parametrize("ei", [False, True])
def test_clean_tape_decodes_to_its_audio(ei, oracle_lib):
    """Property: encode -> (lose rows at the top and bottom of the fields) -> decode returns the audio that went in, every pair
    valid, for both interleave formats; the padding the stitcher found is the number of rows that were lost."""
    cut, tail = (6, 9), (3, 4)
    recs, audio = synth.pcm16x0_sub_stream(6, seed=77, ei=ei, cut=cut, tail_cut=tail, rate_44100=True)
    st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
    pairs, frames = p16.run_cpu(oracle_lib, "orc_", recs, st)
    assert len(pairs) == 6 * 1470 and len(frames) == 6
    assert (frames["flags"] & p16.FA16_PADDING_OK).all()
    assert (frames["odd_top_padding"] == cut[0]).all() and (frames["even_top_padding"] == cut[1]).all()
    lost = np.zeros((6, 2, 735), dtype=bool)                       # sub-lines that never reached the stitcher
    for f in (0, 1):
        lost[:, f, :cut[f] * 3] = True
        lost[:, f, 735 - tail[f] * 3:] = True
    lost = lost.reshape(6, 1470)
    blk = np.arange(490)
    s1 = blk if ei else (blk // 35) * 105 + blk % 35
    step = 490 if ei else 35
    whole = ~(lost[:, s1] | lost[:, s1 + step] | lost[:, s1 + 2 * step])          # blocks with all three lines present
    ok = np.repeat(whole.reshape(-1), 3)
    assert (pairs["audio_word"][ok] == audio[ok]).all()
    assert ((pairs["sample_flags"][ok] & 3) == 3).all()
    assert (pairs["sample_rate"] == 44100).all()


# ---- the kernels on the emulator -----------------------------------------------------------------------------------
import engine_api as ea  # noqa: E402


@pytest.fixture(scope="module")
def emu(emu_lib):
    return ea.bind(emu_lib)


# (the 372-frame tape takes the emulator most of a minute: it is left to the GPU test and to the comparison of the oracle with the real reference)
@pytest.mark.parametrize("name", [n for n in p16.CASES if n != "ei_lost_every_field"])
def test_emu_matches_oracle(name, emu, oracle_lib):
    recs, st, want_p, want_f = _oracle(name)
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames = ea.emu_pcm16_stitch(emu, eng, recs, st)
    emu.sdv_engine_destroy(eng)
    assert rc == 0
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.parametrize("name", ["si_file_marks", "ei_wander"])
def test_emu_streaming_calls_equal_one_call(name, emu, oracle_lib):
    """The stream may arrive in arbitrary pieces: sub-lines wait in the engine for their END_FRAME, the padding and Control Bit
    histories carry over from call to call."""
    recs, st, want_p, want_f = _oracle(name)
    eng = emu.sdv_engine_create(0)
    cuts = [0, 1, 2, 700, 1473, 1474, 3000, 3001, 6000, len(recs) - 1, len(recs)]
    got_p, got_f = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f = ea.emu_pcm16_stitch(emu, eng, recs[a:b], st if a == 0 else None, pair_cap=9000, frame_cap=16)
        assert rc == 0
        got_p.append(p.copy())
        got_f.append(f.copy())
    emu.sdv_engine_destroy(eng)
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def test_emu_edge_inputs(emu, oracle_lib):
    eng = emu.sdv_engine_create(0)
    rc, p, f = ea.emu_pcm16_stitch(emu, eng, np.zeros(0, dtype=p16.SUB_DTYPE))            # empty
    assert rc == 0 and len(p) == 0 and len(f) == 0
    recs, st = p16.make_input("si_clean")
    end = int(np.nonzero(recs["service_type"] == p16.SRV_END_FRAME)[0][0])
    rc, p, f = ea.emu_pcm16_stitch(emu, eng, recs[:end], st)                             # a frame without its END_FRAME: nothing yet
    assert rc == 0 and len(p) == 0 and len(f) == 0
    rc, p, f = ea.emu_pcm16_stitch(emu, eng, recs[end:end + 1])                          # ... now it completes
    want_p, want_f = p16.run_cpu(oracle_lib, "orc_", recs[:end + 1], st)
    assert rc == 0 and _same(p, f, want_p, want_f)
    emu.sdv_engine_destroy(eng)
    eng = emu.sdv_engine_create(0)
    lone = recs[end:end + 1].copy()                                                      # a lone END_FRAME: an all-padding frame
    lone["frame_number"] = 9
    rc, p, f = ea.emu_pcm16_stitch(emu, eng, lone, st)
    want_p, want_f = p16.run_cpu(oracle_lib, "orc_", lone, st)
    assert rc == 0 and len(p) == 1470 and _same(p, f, want_p, want_f)
    rc, p, f = ea.emu_pcm16_stitch(emu, eng, recs, pair_cap=100, frame_cap=8)            # output buffer too small: reported, sized
    assert rc != 0 and b"too small" in emu.sdv_last_error(eng)
    emu.sdv_engine_destroy(eng)


def test_emu_failed_call_leaves_the_stream_untouched(emu, oracle_lib):
    """A call that is refused takes nothing: waiting sub-lines still wait, the histories are as before, and the same records can be
    handed over again."""
    recs, st, want_p, want_f = _oracle("si_wander")
    eng = emu.sdv_engine_create(0)
    cut = 4000                                                                           # inside the third frame
    rc, p0, f0 = ea.emu_pcm16_stitch(emu, eng, recs[:cut], st, pair_cap=20000, frame_cap=16)
    assert rc == 0 and len(f0) == 2
    rc, p, f = ea.emu_pcm16_stitch(emu, eng, recs[cut:], None, pair_cap=100, frame_cap=16)
    assert rc != 0 and b"too small" in emu.sdv_last_error(eng)
    rc, p1_, f1 = ea.emu_pcm16_stitch(emu, eng, recs[cut:], None, pair_cap=20000, frame_cap=16)   # the same records again
    assert rc == 0
    emu.sdv_engine_destroy(eng)
    pairs, frames = np.concatenate([p0, p1_]), np.concatenate([f0, f1])
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def test_emu_burst_counters_as_mask_arithmetic(emu):
    """bursts_word (the analysis' burst counters on 64 blocks at once, pcm16_stitch_device.h) = the block-by-block counters of the reference's
    sweep (pcm16x0datastitcher.cpp:1380-1462) on 100 000 random flag sequences, SI and EI limits, words cut at random places."""
    emu.sdv_emu_selftest_bursts.restype = C.c_int
    emu.sdv_emu_selftest_bursts.argtypes = [C.c_uint64, C.c_int]
    assert emu.sdv_emu_selftest_bursts(20261002, 100000) == 0


def _strangers():
    recs, st = p16.make_input("si_clean")
    recs = recs.copy()
    recs["frame_number"][100] = 2
    yield "later_number", recs, st
    for name in p16.LIVELOCK:
        yield (name,) + p16.make_input(name)


@pytest.mark.parametrize("name,recs,st", list(_strangers()), ids=lambda v: v if isinstance(v, str) else "")
def test_emu_refuses_what_the_reference_never_finishes(name, recs, st, emu):
    """A record numbered for another frame ahead of an END_FRAME: the reference pops its queue up to the stranger
    (pcm16x0datastitcher.cpp:5711-5732), finds the same END_FRAME again and assembles what is left of the frame without end (shown on
    the real reference: more frame reports than the stream has frames, until the driver's buffers are full).  The product refuses."""
    if libs.ref_available():
        pairs, n_frames = p16.run_cpu(libs.load_ref(), "ref_", recs, st, overflow_ok=True)
        assert pairs is None and n_frames > int((recs["service_type"] == p16.SRV_END_FRAME).sum())
    eng = emu.sdv_engine_create(0)
    rc, p, f = ea.emu_pcm16_stitch(emu, eng, recs, st)
    assert rc != 0 and b"another frame" in emu.sdv_last_error(eng)
    emu.sdv_engine_destroy(eng)


def test_emu_long_frame_takes_the_global_memory_path(emu, oracle_lib):
    """More sub-lines in a frame than the LDS staging holds (1536): the same result, read from global memory."""
    recs, _ = synth.pcm16x0_sub_stream(3, seed=88, cut=(4, 6), lead=(30, 30), trail=(20, 25), p_bad=0.03, p_picked=0.05, rate_44100=True)
    assert max(np.diff(np.nonzero(recs["service_type"] == p16.SRV_END_FRAME)[0])) > 1536
    st = p16.default_settings()
    want_p, want_f = p16.run_cpu(oracle_lib, "orc_", recs, st)
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames = ea.emu_pcm16_stitch(emu, eng, recs, st)
    emu.sdv_engine_destroy(eng)
    assert rc == 0 and _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


# ---- the product on the GPU ------------------------------------------------------------------------------------------
def _gpu_run(eng, recs, st, torch, configure=True, **kw):
    from sdvpcmdecoder_amd import Pcm16x0StitchSettings
    if configure:
        eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 36)).cuda()
    p, f = eng.pcm16x0_stitch_frames(d, **kw)
    return p.cpu().numpy().reshape(-1).view(PAIR_DTYPE), f.cpu().numpy().reshape(-1).view(p16.FRASM16_DTYPE)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(p16.CASES))
def test_gpu_matches_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    recs, st, want_p, want_f = _oracle(name)
    pairs, frames = _gpu_run(Engine(0), recs, st, torch)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
@pytest.mark.parametrize("name", p16.GOLDEN)
def test_gpu_matches_golden_from_reference(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    z = np.load(os.path.join(GOLD, "pcm16_" + name + ".npz"))
    recs, st = p16.make_input(name)
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"])
    pairs, frames = _gpu_run(Engine(0), recs, st, torch)
    want_p = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    want_f = np.ascontiguousarray(z["frames"]).view(p16.FRASM16_DTYPE).reshape(-1)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
@pytest.mark.parametrize("ei", [False, True])
def test_gpu_large_batch_in_pieces(ei):
    """600 frames with damage, a wandering picture and file marks in the middle, handed over in three calls cut mid-frame (more
    frames than one internal batch would take with SDV_P16_STITCH_BATCH lowered): equal to the oracle's run."""
    import torch
    from sdvpcmdecoder_amd import Engine
    a, _ = p16.make_stream(250, seed=41, ei=ei, cut=(5, 7), tail_cut=(3, 3), p_bad=0.03, wander=(7, 4), rate_44100=True, new_file=True, end_file=True)
    b, _ = p16.make_stream(350, seed=42, ei=ei, cut=(9, 4), tail_cut=(2, 5), p_bad=0.01, p_picked=0.02, silent=(40, 41, 42), new_file=True, first_frame=1000)
    recs = np.concatenate([a, b])
    st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
    want_p, want_f = p16.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = Engine(0)
    cuts = [0, 123457, 600001, len(recs)]
    got_p, got_f = [], []
    for i, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        p, f = _gpu_run(eng, recs[lo:hi], st, torch, configure=i == 0)
        got_p.append(p)
        got_f.append(f)
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_gpu_random_cuts(seed):
    """A damaged PCM-1630 tape with file tags handed over in pieces cut at random record positions equals the oracle's run."""
    import torch
    from sdvpcmdecoder_amd import Engine
    rng = np.random.default_rng(seed)
    ei = bool(seed & 1)
    recs, _ = p16.make_stream(int(rng.integers(30, 60)), seed=seed, ei=ei, cut=(int(rng.integers(0, 12)), int(rng.integers(0, 12))), tail_cut=(3, 4),
                              p_bad=0.04, p_picked=0.02, wander=(5, 3), rate_44100=bool(seed & 2), new_file=True, end_file=True)
    st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
    want_p, want_f = p16.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = Engine(0)
    cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(1, len(recs)), size=9, replace=False)) + [len(recs)]
    got_p, got_f = [], []
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        p, f = _gpu_run(eng, recs[a:b], st, torch, configure=i == 0)
        got_p.append(p.copy())
        got_f.append(f.copy())
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
@pytest.mark.parametrize("ei", [False, True])
def test_gpu_full_size_round_trip(ei):
    """BASELINE-size property: 4000 clean frames (5.9 M sub-lines) with rows lost at the top and bottom of both fields decode to the
    audio that was encoded, every block whose three lines survived valid; the call's device time is reported."""
    import time
    import torch
    from sdvpcmdecoder_amd import Engine
    n, period = 4000, 50
    cut, tail = (6, 9), (3, 4)
    base, audio = p16.make_stream(period, seed=70 + ei, ei=ei, cut=cut, tail_cut=tail, rate_44100=True)
    tiles = []
    for t in range(n // period):
        b = base.copy()
        b["frame_number"] += period * t
        tiles.append(b)
    recs = np.concatenate(tiles)
    st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
    eng = Engine(0)
    from sdvpcmdecoder_amd import Pcm16x0StitchSettings
    eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    d = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 36)).cuda()
    torch.cuda.synchronize()
    t0 = time.time()
    p, f = eng.pcm16x0_stitch_frames(d)
    torch.cuda.synchronize()
    dt = time.time() - t0
    pairs = p.cpu().numpy().reshape(-1).view(PAIR_DTYPE)
    frames = f.cpu().numpy().reshape(-1).view(p16.FRASM16_DTYPE)
    print(f"\n[pcm16x0 stitch, {'EI' if ei else 'SI'}] {n} frames in {dt * 1e3:.1f} ms: {n / dt:,.0f} frames/s")
    assert len(pairs) == n * 1470 and len(frames) == n
    assert (frames["flags"] & p16.FA16_PADDING_OK).all()
    lost = np.zeros((2, 735), dtype=bool)
    for fl in (0, 1):
        lost[fl, :cut[fl] * 3] = True
        lost[fl, 735 - tail[fl] * 3:] = True
    lost = lost.reshape(1470)
    blk = np.arange(490)
    s1 = blk if ei else (blk // 35) * 105 + blk % 35
    step = 490 if ei else 35
    whole = np.repeat(~(lost[s1] | lost[s1 + step] | lost[s1 + 2 * step]), 3)
    got = pairs["audio_word"].reshape(n // period, period, 1470, 2)
    want = audio.reshape(period, 1470, 2)
    assert (got[:, :, whole] == want[None, :, whole]).all()
    assert ((pairs["sample_flags"].reshape(n, 1470, 2)[:, whole] & 3) == 3).all()


@pytest.mark.gpu
@pytest.mark.parametrize("ei", [False, True])
def test_gpu_video_to_audio(ei):
    """The whole PCM-16x0 path on the device - video frames -> sdv_pcm16x0_binarize_frames -> sdv_pcm16x0_stitch_frames - against the
    oracle's two halves chained on the CPU, and against the audio the tape was made from: a 486-row capture shows 243 of the 245 PCM
    lines of a field, the blocks that lost one line come back through their P-code."""
    import torch
    import pcm16_frames_api as fa
    from sdvpcmdecoder_amd import Engine, Pcm16x0StitchSettings
    n = 6
    luma, audio, seen = synth.pcm16x0_tape_frames(n, seed=5 + ei, ei=ei, noise_sigma=4.0)
    orc = libs.load_oracle()
    want_recs, _ = fa.run_cpu(orc, "orc_", luma, 2, {})
    st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
    want_p, want_f = p16.run_cpu(orc, "orc_", want_recs, st)
    eng = Engine(0)
    eng.setPCMType(1)              # PCM_PCM16X0
    eng.setBinarizationMode(2)     # MODE_NORMAL
    eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    lines, _ = eng.pcm16x0_binarize_frames(torch.from_numpy(luma).cuda())
    p, f = eng.pcm16x0_stitch_frames(lines)
    pairs = p.cpu().numpy().reshape(-1).view(PAIR_DTYPE)
    frames = f.cpu().numpy().reshape(-1).view(p16.FRASM16_DTYPE)
    assert lines.cpu().numpy().tobytes() == want_recs.tobytes()
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)
    assert (frames["flags"] & p16.FA16_PADDING_OK).all() and (frames["odd_sample_rate"] == 44100).all()
    assert (pairs["audio_word"] == audio).all() and ((pairs["sample_flags"] & 3) == 3).all()


# ---- one file through both halves, against the real reference's two workers --------------------------------------------
def _e2e_fixture(ei):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_pcm16", os.path.join(GOLD, "make_golden_pcm16.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    z = np.load(os.path.join(GOLD, "e2e_pcm16x0_%s.npz" % ("ei" if ei else "si")))
    luma, audio = mg.make_e2e_luma(ei)
    assert hashlib.sha256(luma.tobytes()).hexdigest() == str(z["luma_sha256"]), "regenerated video differs from the fixture's"
    want_p = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    want_f = np.ascontiguousarray(z["frames"]).view(p16.FRASM16_DTYPE).reshape(-1)
    return luma, audio, z, want_p, want_f


@pytest.mark.parametrize("ei", [False, True])
def test_oracle_whole_file_matches_reference_golden(ei, oracle_lib):
    """video -> oracle VideoToDigital (TYPE_PCM16X0, NEW_FILE .. filler frame + END_FILE) -> oracle stitcher == both real reference workers,
    and the audio the tape was made from"""
    import pcm16_frames_api as fa
    luma, audio, z, want_p, want_f = _e2e_fixture(ei)
    recs, _ = fa.run_cpu(oracle_lib, "orc_", luma, 2, dict(new_file=True, end_file=True))
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["recs_sha256"])
    st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
    pairs, frames = p16.run_cpu(oracle_lib, "orc_", recs, st)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)
    assert (pairs["service_type"] == 1).sum() == 1 and pairs["service_type"][-1] == 2
    assert (pairs["audio_word"][pairs["service_type"] == 0] == audio).all()


@pytest.mark.gpu
@pytest.mark.parametrize("ei", [False, True])
def test_gpu_whole_file_matches_reference_golden(ei):
    """The drop-in path end to end on the GPU: sdv_pcm16x0_binarize_frames(NEW_FILE | END_FILE) -> sdv_pcm16x0_stitch_frames, device buffers
    handed from one stage to the next, against the output of the real reference's two workers."""
    import torch
    from sdvpcmdecoder_amd import Engine, Pcm16x0StitchSettings
    luma, audio, z, want_p, want_f = _e2e_fixture(ei)
    st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
    eng = Engine(0)
    eng.setPCMType(1)
    eng.setBinarizationMode(2)
    eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    lines, _ = eng.pcm16x0_binarize_frames(torch.from_numpy(luma).cuda(), new_file=True, end_file=True)
    assert hashlib.sha256(lines.cpu().numpy().tobytes()).hexdigest() == str(z["recs_sha256"])
    p, f = eng.pcm16x0_stitch_frames(lines)
    pairs = p.cpu().numpy().reshape(-1).view(PAIR_DTYPE)
    frames = f.cpu().numpy().reshape(-1).view(p16.FRASM16_DTYPE)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)

# ---- long tapes: the padding history saturates (65 entries), and the decision kernel's chunk-at-once path takes over ------------
def _long_tape(ei, hurt):
    from sdvpcmdecoder_amd import synth
    recs = synth.pcm16x0_tape(150, seed=21 if ei else 20, period=150, ei=ei, p_bad=0.01)
    recs = recs.copy()
    if hurt:
        # frames in the middle lose most of their CRCs (their padding checks fail: the in-order path and the way back into the fast one),
        # and a file boundary sits in the tape (the history starts over)
        ends = np.nonzero(recs["service_type"] == 5)[0]
        for f in (70, 71, 100, 131):
            a, b = ends[f - 1] + 1, ends[f]
            sel = np.arange(a, b)[::2]
            recs["flags"][sel] &= ~np.uint8(64) & 0xFF          # SDV_LF_CRC_VALID
    return recs


@pytest.mark.parametrize("ei,hurt", [(False, True), (True, True)])       # (the clean tapes are the first 70 frames of these)
def test_emu_long_tape_matches_oracle(ei, hurt, emu, oracle_lib):
    recs = _long_tape(ei, hurt)
    st = p16.default_settings(format=2 if ei else 1)
    want_p, want_f = p16.run_cpu(oracle_lib, "orc_", recs, st)
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames = ea.emu_pcm16_stitch(emu, eng, recs, st)
    emu.sdv_engine_destroy(eng)
    assert rc == 0 and len(want_f) == 150
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.parametrize("cuts", [(68, 69, 101, 140)])
def test_emu_long_ei_tape_in_calls_matches_oracle(cuts, emu, oracle_lib):
    """The EI tape above in several calls: from the second call on the padding history is full when a call begins, so the analysis makes its padding
    tables for that one padding (round 4) - all of them again for the hurt frames (70, 71, 100, 131), and the whole call once more where the history
    says something else in the middle of it (the second tape: its later frames are padded otherwise)."""
    recs = _long_tape(True, True)
    st = p16.default_settings(format=2)
    want_p, want_f = p16.run_cpu(oracle_lib, "orc_", recs, st)
    ends = np.nonzero(recs["service_type"] == 5)[0]
    eng = emu.sdv_engine_create(0)
    got_p, got_f = [], []
    lo = 0
    for k, c in enumerate(list(cuts) + [150]):
        hi = int(ends[c - 1]) + 1
        rc, pairs, frames = ea.emu_pcm16_stitch(emu, eng, recs[lo:hi], st if k == 0 else None)
        assert rc == 0
        got_p.append(pairs); got_f.append(frames)
        lo = hi
    emu.sdv_engine_destroy(eng)
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def test_emu_ei_call_runs_again_with_full_tables(emu, oracle_lib, monkeypatch, capfd):
    """The way back when a frame needed more than the one padding its table was made for: the emulator build's hook makes every such frame say so
    (SDV_P16_HINT_SKEW), the call then runs once more with full tables - the same answer, and the trace shows the second attempt."""
    recs = _long_tape(True, True)
    ends = np.nonzero(recs["service_type"] == 5)[0]
    recs = recs[:int(ends[94]) + 1]                # 95 frames are enough: 72 fill the history, the hurt frames 70 and 71 among them
    st = p16.default_settings(format=2)
    want_p, want_f = p16.run_cpu(oracle_lib, "orc_", recs, st)
    cut = int(ends[71]) + 1
    eng = emu.sdv_engine_create(0)
    rc, p0, f0 = ea.emu_pcm16_stitch(emu, eng, recs[:cut], st)
    assert rc == 0
    monkeypatch.setenv("SDV_P16_HINT_SKEW", "1"); monkeypatch.setenv("SDV_P16_HINT_TRACE", "1")
    rc, p1, f1 = ea.emu_pcm16_stitch(emu, eng, recs[cut:], None)
    assert rc == 0
    emu.sdv_engine_destroy(eng)
    err = capfd.readouterr().err
    assert "attempt 0: hint on" in err and "needs full tables: 1" in err and "attempt 1: hint off" in err, err
    pairs, frames = np.concatenate([p0, p1]), np.concatenate([f0, f1])
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
@pytest.mark.parametrize("ei", [False, True])
def test_gpu_long_damaged_tape_matches_oracle(ei, oracle_lib):
    """1 200 frames with a hurt frame every ~40 and a second file in the middle: the chunk-at-once decisions, the in-order loop and the way
    from one to the other, across batch boundaries and the three streams, against the oracle's sequential decode."""
    import torch
    from sdvpcmdecoder_amd import Engine, synth, Pcm16x0StitchSettings
    n = 1200
    recs = synth.pcm16x0_tape(n, seed=31 if ei else 30, period=300, ei=ei, p_bad=0.01).copy()
    ends = np.nonzero(recs["service_type"] == 5)[0]
    rng = np.random.default_rng(3)
    for f in sorted(rng.choice(np.arange(70, n - 1), 30, replace=False)):
        a, b = ends[f - 1] + 1, ends[f]
        recs["flags"][np.arange(a, b)[::2]] &= ~np.uint8(64) & 0xFF
    st = p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI)
    want_p, want_f = p16.run_cpu(oracle_lib, "orc_", recs, st)
    eng = Engine(0)
    eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    p, f = eng.pcm16x0_stitch_frames(torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 36)).cuda())
    pairs = p.cpu().numpy().reshape(-1).view(PAIR_DTYPE)
    frames = f.cpu().numpy().reshape(-1).view(p16.FRASM16_DTYPE)
    assert len(want_f) == n and _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)
