"""Stitch stage of the product (sdv_stitch_frames): the HIP kernels against the oracle restatement of STC007DataStitcher
and the golden PCMSamplePair fixtures of the real reference.  CPU legs run the same kernel source on the SIMT emulator."""
import ctypes as C
import os

import numpy as np
import pytest

import engine_api as ea
import libs
import stitch_api as sa
import stitch_cases as sc
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EMU_CASES = ["ntsc_clean", "ntsc_bad5", "ntsc_burst300", "pal_bad5", "f1_16bit_bad5", "ntsc_ctrlblk", "ntsc_bff", "ntsc_drift",
             "ntsc_bad10_no_q_cwd", "ntsc_bad30_noecc", "ntsc_res14_m2", "ntsc_toplinefix_sr44100"]


def _same(pairs, frames, want_p, want_f):
    return len(pairs) == len(want_p) and pairs.tobytes() == want_p.tobytes() and len(frames) == len(want_f) and frames.tobytes() == want_f.tobytes()


def _diff(pairs, frames, want_p, want_f):
    out = [f"pairs {len(pairs)} vs {len(want_p)}, frames {len(frames)} vs {len(want_f)}"]
    for i in range(min(len(frames), len(want_f))):
        if frames[i].tobytes() != want_f[i].tobytes():
            out.append(f" frame {i}: " + str([(n, frames[i][n], want_f[i][n]) for n in sa.FRASM_DTYPE.names if frames[i][n] != want_f[i][n]]))
    n = min(len(pairs), len(want_p))
    d = np.nonzero((pairs[:n].view(np.uint8).reshape(n, 12) != want_p[:n].view(np.uint8).reshape(n, 12)).any(axis=1))[0]
    out.append(f" {len(d)} pairs differ, first at {d[:5]}")
    return "\n".join(out)


@pytest.fixture(scope="module")
def emu(emu_lib):
    return ea.bind(emu_lib)


@pytest.mark.parametrize("name", EMU_CASES)
def test_emu_matches_oracle(name, emu, oracle_lib):
    recs, st = sc.make_input(name, lambda luma: oracle_binarize(luma, mode=2))
    want_p, want_f = sa.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames = ea.emu_stitch(emu, eng, recs, st)
    emu.sdv_engine_destroy(eng)
    assert rc == 0
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def test_emu_streaming_calls_equal_one_call(emu, oracle_lib):
    """The stream may arrive in arbitrary pieces (frame by frame, mid-frame): the engine keeps what cannot be stitched yet."""
    recs, st = sc.make_input("ntsc_bad5", lambda luma: oracle_binarize(luma, mode=2))
    want_p, want_f = sa.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = emu.sdv_engine_create(0)
    cuts = [0, 1, 300, 490, 1000, 1471, 1472, 2500, len(recs)]
    got_p, got_f = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f = ea.emu_stitch(emu, eng, recs[a:b], st if a == 0 else None, pair_cap=20000, frame_cap=64)
        assert rc == 0
        got_p.append(p.copy()); got_f.append(f.copy())
    emu.sdv_engine_destroy(eng)
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def _frame_cuts(recs, frames_per_call):
    """Cut positions behind the END_FRAME record of every frames_per_call[i]-th frame (the rest goes with the last call)."""
    ends = np.nonzero(recs["service_type"] == 5)[0]
    cuts, k = [0], 0
    for c in frames_per_call:
        k += c
        if k - 1 >= len(ends) - 1:
            break
        cuts.append(int(ends[k - 1]) + 1)
    cuts.append(len(recs))
    return cuts


def _pipelined_tape(n=20, seed=401, second=None, end_file=True, **first):
    """n clean frames (and, with `second`, frames of another kind behind them, numbered on), END_FILE at the end."""
    luma, _, _ = synth.stc007_frames(n, seed=seed, noise_sigma=2.0, **first)
    recs, _ = oracle_binarize(np.ascontiguousarray(luma), mode=2)
    if second is not None:
        kw = dict(second)
        m = kw.pop("n")
        if kw.pop("f1", False):
            rng = np.random.default_rng(seed + 1000)
            kw["words"] = synth.interleave_stream_f1(rng.integers(0, 1 << 16, size=(m * 2 * 245, 6), dtype=np.uint32))
        luma2, _, _ = synth.stc007_frames(m, seed=seed + 1, noise_sigma=2.0, **kw)
        recs2, _ = oracle_binarize(np.ascontiguousarray(luma2), mode=2, first_frame_no=1 + n, new_file=False)
        recs = np.concatenate([recs, recs2])
    return sa.with_end_file(recs) if end_file else recs


PIPE_CASES = {
    # name: (tape kwargs, frames per call, damage, calls expected to run pipelined at least)
    "plays": (dict(n=20), [3] * 7, None, 3),
    # 65 turns fill the field order history: from there on the host's check is the short one (pipelined == 2), until the errors come
    "plays_long": (dict(n=112), [12] * 10, (23, 0.03, 0, 97), 6),
    "ragged_calls": (dict(n=20), [3, 3, 3, 6, 1, 2, 2], None, 1),
    "burst_behind_a_steady_start": (dict(n=20), [4] * 5, (21, 0.0, 700, 9), 1),
    "single_errors_behind_a_steady_start": (dict(n=20), [4] * 5, (22, 0.05, 0, 9), 1),
    "resolution_changes": (dict(n=10, second=dict(n=10, f1=True)), [3] * 7, None, 1),
    "field_order_changes": (dict(n=10, second=dict(n=10, bff=True)), [3] * 7, None, 1),
    # shorter frames than the calls before had: more of them than the launch was made for, the call starts over
    "more_frames_than_estimated": (dict(n=9, height=576, lines_per_field=294, second=dict(n=21)), [3, 3, 3, 14, 3, 3], None, 1),
}


@pytest.mark.parametrize("name", list(PIPE_CASES))
def test_emu_pipelined_calls_equal_one_call(name, emu, oracle_lib):
    """A stream that plays is stitched without waiting for the host between the stages (stitch_engine.inc "1p"): whatever the later calls meet -
    more frames than the estimate, damage, another resolution or field order, the end of the file - the PCM stream is the sequential one's."""
    tape_kw, per_call, dmg, want_piped = PIPE_CASES[name]
    recs = _pipelined_tape(**tape_kw)
    if dmg is not None:
        seed, p_bad, burst, from_frame = dmg
        ends = np.nonzero(recs["service_type"] == 5)[0]
        a = int(ends[from_frame - 1]) + 1
        tail = sc.damage(recs[a:], seed, p_bad, burst=burst)
        recs = np.concatenate([recs[:a], tail])
    st = sa.default_settings()
    want_p, want_f = sa.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = emu.sdv_engine_create(0)
    cuts = _frame_cuts(recs, per_call)
    got_p, got_f, piped = [], [], []
    info = ea.StitchInfo()
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f = ea.emu_stitch(emu, eng, recs[a:b], st if a == 0 else None, pair_cap=30000, frame_cap=64)
        assert rc == 0, (a, b)
        got_p.append(p.copy()); got_f.append(f.copy())
        assert emu.sdv_get_stitch_info(eng, C.byref(info)) == 0
        piped.append(int(info.pipelined))
    emu.sdv_engine_destroy(eng)
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), (_diff(pairs, frames, want_p, want_f), piped)
    assert sum(1 for x in piped if x) >= want_piped, piped
    if name == "plays_long":
        assert piped[6] == 2 and piped[7] == 2, piped
    assert piped[0] == 0 and piped[-1] == 0, piped          # a cold start and the call with the end of the file never are


def test_emu_pipelined_calls_feed_the_visualiser(emu, oracle_lib):
    """The data blocks and the assembled lines of pipelined calls (their turns are run once more from the final hand-overs; the first
    round of such a call never went through the host's per-turn arrays) equal those of the sequential run."""
    recs = _pipelined_tape(n=14)
    st = sa.default_settings()
    _, _, want_blocks = sa.run_cpu_blocks(libs.load_oracle(), "orc_", recs, st)
    want_lines, _ = sa.last_asm_lines(libs.load_oracle(), "orc_")
    emu.sdv_set_stitch_block_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    emu.sdv_stitch_block_count.restype = C.c_size_t
    emu.sdv_stitch_block_count.argtypes = [C.c_void_p]
    emu.sdv_set_stitch_line_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    emu.sdv_stitch_line_count.restype = C.c_size_t
    emu.sdv_stitch_line_count.argtypes = [C.c_void_p]
    eng = emu.sdv_engine_create(0)
    bbuf = np.zeros(len(want_blocks) + 8, dtype=sa.BLOCK_DTYPE)
    lbuf = np.zeros(len(want_lines) + 8, dtype=sa.ASM_DTYPE)
    assert emu.sdv_set_stitch_block_output(eng, bbuf.ctypes.data, len(bbuf)) == 0
    assert emu.sdv_set_stitch_line_output(eng, lbuf.ctypes.data, len(lbuf)) == 0
    cuts = _frame_cuts(recs, [3] * 5)
    blocks, lines, piped = [], [], []
    info = ea.StitchInfo()
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, _, _ = ea.emu_stitch(emu, eng, recs[a:b], st if a == 0 else None, pair_cap=30000, frame_cap=64)
        assert rc == 0
        blocks.append(bbuf[:emu.sdv_stitch_block_count(eng)].copy()); lines.append(lbuf[:emu.sdv_stitch_line_count(eng)].copy())
        assert emu.sdv_get_stitch_info(eng, C.byref(info)) == 0
        piped.append(int(info.pipelined))
    emu.sdv_engine_destroy(eng)
    assert sum(1 for x in piped if x) >= 2, piped
    assert np.concatenate(blocks).tobytes() == want_blocks.tobytes()
    assert np.concatenate(lines).tobytes() == want_lines.tobytes()


def test_emu_pipelined_call_reports_what_the_host_would_have_refused(emu, oracle_lib):
    """Frame numbers that do not increase are refused before the turns run; a pipelined call has run them already - the call fails all the same."""
    recs = _pipelined_tape(n=14, end_file=False)
    st = sa.default_settings()
    eng = emu.sdv_engine_create(0)
    cuts = _frame_cuts(recs, [3, 3, 3, 3])
    info = ea.StitchInfo()
    for a, b in zip(cuts[:3], cuts[1:4]):
        rc, _, _ = ea.emu_stitch(emu, eng, recs[a:b], st if a == 0 else None, pair_cap=30000, frame_cap=64)
        assert rc == 0
    assert emu.sdv_get_stitch_info(eng, C.byref(info)) == 0 and info.pipelined == 1
    again = recs[cuts[2]:cuts[3]].copy()            # the frames of the last call once more: numbers go backwards
    rc, _, _ = ea.emu_stitch(emu, eng, again, None, pair_cap=30000, frame_cap=64)
    assert rc == -4 and b"frame numbers" in emu.sdv_last_error(eng)      # SDV_ERR_UNSUPPORTED
    emu.sdv_engine_destroy(eng)


def test_emu_empty_and_tiny_inputs(emu, oracle_lib):
    """Nothing in, nothing out; a single frame has no successor yet and produces nothing until the next one arrives; a stream that
    is only the end-of-file frame produces nothing at all."""
    recs, st = sc.make_input("ntsc_clean", lambda luma: oracle_binarize(luma, mode=2))
    want_p, want_f = sa.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = emu.sdv_engine_create(0)
    rc, p, f = ea.emu_stitch(emu, eng, recs[:0], st, pair_cap=16, frame_cap=4)
    assert rc == 0 and len(p) == 0 and len(f) == 0
    one = 1 + 489                                   # NEW_FILE + the first frame
    rc, p, f = ea.emu_stitch(emu, eng, recs[:one], None, pair_cap=16, frame_cap=4)
    assert rc == 0 and len(p) == 0 and len(f) == 0
    rc, p, f = ea.emu_stitch(emu, eng, recs[one:], None)
    assert rc == 0 and _same(p, f, want_p, want_f), _diff(p, f, want_p, want_f)
    emu.sdv_engine_destroy(eng)
    eng = emu.sdv_engine_create(0)
    tail = recs[-(486 + 4):]                        # filler frame + END_FILE only
    rc, p, f = ea.emu_stitch(emu, eng, tail, st, pair_cap=16, frame_cap=4)
    assert rc == 0 and len(p) == 0 and len(f) == 0
    emu.sdv_engine_destroy(eng)


def test_emu_rejects_what_it_cannot_reproduce(emu, oracle_lib):
    """A line numbered for a later frame in the middle of a frame: the real stitcher pops its queue up to that line and never gets past it
    (more frame reports than the stream has frames until the driver's buffers are full).  The product refuses the stream."""
    recs, st = sc.make_input("ntsc_clean", lambda luma: oracle_binarize(luma, mode=2))
    bad = recs.copy()
    bad["frame_number"][700] += 7                  # a line that claims another frame
    if libs.ref_available():
        import ctypes as C
        f = libs.load_ref().ref_stitch_run
        f.restype = C.c_long
        f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(sa.StitchSettings), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        pairs, frames, nf = np.zeros(40000, dtype=sa.PAIR_DTYPE), np.zeros(40, dtype=sa.FRASM_DTYPE), C.c_size_t(0)
        assert f(bad.ctypes.data, len(bad), C.byref(st), pairs.ctypes.data, len(pairs), frames.ctypes.data, len(frames), C.byref(nf)) == -1
        assert nf.value > int((bad["service_type"] == 5).sum()) + 2
    eng = emu.sdv_engine_create(0)
    rc, _, _ = ea.emu_stitch(emu, eng, bad, st)
    assert rc == -4 and b"frame" in emu.sdv_last_error(eng)      # SDV_ERR_UNSUPPORTED, loudly
    emu.sdv_engine_destroy(eng)
    eng = emu.sdv_engine_create(0)
    rc, _, _ = ea.emu_stitch(emu, eng, recs, st, pair_cap=100)
    assert rc == -1                                             # output buffer too small
    emu.sdv_engine_destroy(eng)


# ---------------------------------------------------------------------------------------------------------- GPU
def _gpu_stitch(eng, recs, st, torch):
    eng.set_stitch_settings(_settings(st))
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 48)).cuda()
    p, f = eng.stitch_frames(d)
    pairs = p.cpu().numpy().reshape(-1).view(sa.PAIR_DTYPE)
    frames = f.cpu().numpy().reshape(-1).view(sa.FRASM_DTYPE)
    return pairs, frames


def _settings(st):
    from sdvpcmdecoder_amd import StitchSettings
    out = StitchSettings()
    C.memmove(C.byref(out), C.byref(st), C.sizeof(out))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(sc.CASES))
def test_gpu_matches_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    recs, st = sc.make_input(name, lambda luma: oracle_binarize(luma, mode=2))
    want_p, want_f = sa.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = Engine(0)
    pairs, frames = _gpu_stitch(eng, recs, st, torch)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sc.GOLDEN)
def test_gpu_matches_golden_from_reference(name):
    """End to end on the GPU: synthetic video -> sdv_binarize_frames -> sdv_stitch_frames == the real reference's
    VideoToDigital + STC007DataStitcher output committed as a fixture."""
    import torch
    from sdvpcmdecoder_amd import Engine, LINE_DTYPE
    z = np.load(os.path.join(GOLD, "stitch_" + name + ".npz"))
    eng = Engine(0)

    def gpu_binarize(luma):
        lines, stats = eng.binarize_frames(torch.from_numpy(luma).cuda(), first_frame_no=1, new_file=True)
        return lines.cpu().numpy().reshape(-1).view(LINE_DTYPE), None
    recs, st = sc.make_input(name, gpu_binarize)
    assert sc.digest(recs) == str(z["input_sha256"])
    pairs, frames = _gpu_stitch(eng, recs, st, torch)
    want_p = np.ascontiguousarray(z["pairs"]).view(sa.PAIR_DTYPE).reshape(-1)
    want_f = np.ascontiguousarray(z["frames"]).view(sa.FRASM_DTYPE).reshape(-1)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def _e2e_fixture():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_stitch", os.path.join(GOLD, "make_golden_stitch.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    z = np.load(os.path.join(GOLD, "e2e_ntsc_file.npz"))
    want_p = np.ascontiguousarray(z["pairs"]).view(sa.PAIR_DTYPE).reshape(-1)
    want_f = np.ascontiguousarray(z["frames"]).view(sa.FRASM_DTYPE).reshape(-1)
    return mg.make_e2e_luma(), z, want_p, want_f


def test_oracle_whole_file_matches_reference_golden(oracle_lib):
    """video -> oracle VideoToDigital (NEW_FILE .. filler frame + END_FILE) -> oracle stitcher == both real reference workers"""
    luma, z, want_p, want_f = _e2e_fixture()
    recs, stats = oracle_binarize(luma, mode=2, new_file=True, end_file=True)
    assert sc.digest(recs) == str(z["recs_sha256"]) and stats.tobytes() == z["stats"].tobytes()
    pairs, frames = sa.run_cpu(libs.load_oracle(), "orc_", recs, sa.default_settings())
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)
    assert (pairs["service_type"] == 1).sum() == 1 and pairs["service_type"][-1] == 2


@pytest.mark.gpu
def test_gpu_whole_file_matches_reference_golden():
    """The drop-in path end to end on the GPU: sdv_binarize_frames(NEW_FILE | END_FILE) -> sdv_stitch_frames, device
    buffers handed from one stage to the next, against the output of the real reference's two workers."""
    import torch
    from sdvpcmdecoder_amd import Engine, LINE_DTYPE
    luma, z, want_p, want_f = _e2e_fixture()
    eng = Engine(0)
    lines, stats = eng.binarize_frames(torch.from_numpy(luma).cuda(), first_frame_no=1, new_file=True, end_file=True)
    assert sc.digest(lines.cpu().numpy().reshape(-1).view(LINE_DTYPE)) == str(z["recs_sha256"])
    assert stats.cpu().numpy().tobytes() == z["stats"].tobytes()
    eng.set_stitch_settings(_settings(sa.default_settings()))
    p, f = eng.stitch_frames(lines)
    pairs = p.cpu().numpy().reshape(-1).view(sa.PAIR_DTYPE)
    frames = f.cpu().numpy().reshape(-1).view(sa.FRASM_DTYPE)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
def test_gpu_long_stream_in_batches():
    """300 frames with dropouts, stitched in three calls: the PCM stream equals the oracle's sequential run and every
    turn ran from its predecessor's final state (rounds reported by the engine stay small)."""
    import torch
    from sdvpcmdecoder_amd import Engine, synth, LINE_DTYPE
    n = 300
    eng = Engine(0)
    luma, _ = synth.stc007_frames_torch(n, seed=5, device="cuda", noise_sigma=3.0)
    lines, _ = eng.binarize_frames(luma, first_frame_no=1, new_file=True)
    recs = sa.with_end_file(lines.cpu().numpy().reshape(-1).view(LINE_DTYPE))
    recs = sc.damage(recs, 77, 0.02, burst=700)
    st = sa.default_settings()
    want_p, want_f = sa.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng.set_stitch_settings(_settings(st))
    d = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 48)).cuda()
    cuts = [0, 40 * 489 + 7, 200 * 489, len(recs)]
    got_p, got_f, rounds = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        p, f = eng.stitch_frames(d[a:b].contiguous())
        got_p.append(p.cpu().numpy().reshape(-1).view(sa.PAIR_DTYPE).copy()); got_f.append(f.cpu().numpy().reshape(-1).view(sa.FRASM_DTYPE).copy())
        rounds.append(eng.stitch_info().rounds)
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)
    assert max(rounds) <= 12, rounds


@pytest.mark.gpu
def test_gpu_full_size_audio_recovery():
    """BASELINE's batch size through the whole path (10 000 NTSC frames -> 14.7 M sample pairs): too big for the oracle, so the check
    is the property the format was built for - although every field loses three lines (two cut off by the 486-row frame, the
    first visible one discarded by the duplicate-line rule), P/Q correction recovers every audio word of the generator, in order,
    and no block is dropped."""
    import torch
    from sdvpcmdecoder_amd import Engine, synth
    n = 10000
    eng = Engine(0)
    luma, w9 = synth.stc007_frames_torch(n, seed=2, device="cuda", noise_sigma=4.0, cyclic=True)
    lines, _ = eng.binarize_frames(luma, first_frame_no=1, new_file=True, end_file=True)
    del luma
    eng.set_stitch_settings(_settings(sa.default_settings()))
    pairs, frames = eng.stitch_frames(lines)
    info = eng.stitch_info()
    assert info.steps == n and info.rounds <= 3
    nb = w9.shape[0]                                            # one block starts at every line of the tape
    b = torch.arange(nb, device="cuda")
    exp = torch.stack([w9[(b + 16 * k) % nb, k] for k in range(6)], dim=1)           # L0 R0 L1 R1 L2 R2 of block b
    exp = ((exp << 2) & 0xFFFF).to(torch.int32)
    exp = torch.where(exp >= 32768, exp - 65536, exp).to(torch.int16).reshape(nb, 3, 2)
    p = pairs.contiguous()
    words = p[:, 0:4].contiguous().view(torch.int16).reshape(-1, 2)
    flags, service = p[:, 4:6], p[:, 9]
    assert int((service == 1).sum()) == 1 and int(service[0]) == 1 and int(service[-1]) == 2 and int((service != 0).sum()) == 2
    body, bflags = words[1:-1], flags[1:-1]                     # 3 pairs per block
    assert body.shape[0] % 3 == 0
    blocks = body.reshape(-1, 3, 2)
    # align: a block well inside the tape must be one block of the generator
    probe = blocks[3000]
    hit = torch.nonzero((exp == probe[None]).all(dim=2).all(dim=1)).reshape(-1)
    assert hit.numel() == 1
    b0 = int(hit[0]) - 3000                                      # generator block of decoded block 0
    k = torch.arange(blocks.shape[0], device="cuda")
    want = exp[(b0 + k) % nb]
    same = (blocks == want).all(dim=2).all(dim=1)
    # the first blocks reach into the filler lines in front of the tape, the last ones into the end-of-file flush: 112 lines each
    inner = same[200:-200]
    assert bool(inner.all()), f"{int((~inner).sum())} blocks differ from the generator"
    ok = ((bflags.reshape(-1, 3, 2)[200:-200] & 3) == 3)
    assert bool(ok.all())                                        # block ok + word valid everywhere
    fr = frames.cpu().numpy().reshape(-1).view(sa.FRASM_DTYPE)
    real = fr[fr["service_type"] == 0]
    assert len(real) == n and (real["blocks_drop"][1:-1] == 0).all() and (real["blocks_total"][1:-1] == 490).all()
    assert ((real["flags"][1:-1] & 0x18) == 0x18).all()         # inner and outer padding found (the last frame borders the filler frame)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_gpu_random_cuts_and_two_files(seed):
    """Two files back to back (NEW_FILE ... END_FILE, NEW_FILE ... END_FILE), a little damage, handed over in pieces cut at random
    record positions: calls on a playing tape write their pairs straight into the caller's buffer, calls that meet file tags fall
    back to the packed path - the concatenated output equals the oracle's sequential run whatever the cuts are.  (What shares a
    call with an END_FILE frame is flushed with it, as the reference flushes its input queue: the files meet at a call boundary.)"""
    import torch
    from sdvpcmdecoder_amd import Engine, synth, LINE_DTYPE
    rng = np.random.default_rng(seed)
    eng = Engine(0)
    st = sa.default_settings()
    eng.set_stitch_settings(_settings(st))
    eng.reset_stitcher()
    fno = 1
    got_p, got_f, want_p, want_f = [], [], [], []
    for part, nfr in enumerate((int(rng.integers(30, 60)), int(rng.integers(20, 50)))):
        luma, _ = synth.stc007_frames_torch(nfr, seed=seed * 10 + part, device="cuda", noise_sigma=3.0)
        eng.reset_stream()
        lines, _ = eng.binarize_frames(luma, first_frame_no=fno, new_file=True, end_file=True)
        recs = sc.damage(lines.cpu().numpy().reshape(-1).view(LINE_DTYPE).copy(), seed + part, 0.01, burst=0)
        fno += nfr + 1
        wp, wf = sa.run_cpu(libs.load_oracle(), "orc_", recs, st)
        want_p.append(wp)
        want_f.append(wf)
        d = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), 48)).cuda()
        cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(1, len(recs)), size=5, replace=False)) + [len(recs)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            p, f = eng.stitch_frames(d[a:b].contiguous())
            got_p.append(p.cpu().numpy().reshape(-1).view(sa.PAIR_DTYPE).copy())
            got_f.append(f.cpu().numpy().reshape(-1).view(sa.FRASM_DTYPE).copy())
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    want_p, want_f = np.concatenate(want_p), np.concatenate(want_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)
