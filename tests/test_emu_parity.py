"""CPU tests of the HIP kernel SOURCE: sdvpcmdecoder_amd/csrc/stc007_device.h + engine.inc compiled with g++
on the single-wavefront SIMT emulator (tests/emu/hip_emu.h) and compared bit-for-bit with the oracle.
This is not the product path (the product has no CPU path); it lets the kernel logic and the
chain-speculation scheduler be checked in the GPU-less container.  Sizes are small: the emulator is slow."""
import ctypes as C

import numpy as np
import pytest

import engine_api
import golden_cases
import libs
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import synth


def emu_run(emu, luma, mode, flags=1, first=1, eng=None):
    own = eng is None
    if own:
        eng = C.c_void_p(emu.sdv_engine_create(0))
        emu.sdv_set_mode(eng, mode)
    rc, recs, stats = engine_api.emu_binarize(emu, eng, luma, first_frame_no=first, flags=flags)
    info = engine_api.RunInfo()
    emu.sdv_get_run_info(eng, C.byref(info))
    if own:
        emu.sdv_engine_destroy(eng)
    assert rc == 0
    return recs, stats, info


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_emu_small_frames(emu_lib, oracle_lib, mode):
    cases = [dict(n_frames=3, seed=1, height=48),
             dict(n_frames=2, seed=2, height=40, noise_sigma=10.0, blur=2),
             dict(n_frames=3, seed=3, height=50, lines_per_field=25, ctrl_block=True),
             dict(n_frames=2, seed=4, height=24 if mode >= 2 else 40, noise_sigma=25.0, blur=3)]
    for kw in cases:
        luma, _, _ = synth.stc007_frames(**kw)
        want, want_stats = oracle_binarize(luma, mode=mode)
        got, got_stats, info = emu_run(emu_lib, luma, mode)
        assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
        assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()


def test_emu_end_of_file_frame_and_next_source(emu_lib, oracle_lib):
    """SDV_FLAG_END_FILE appends the filler frame + END_FILE the input plugin emits (vin_ffmpeg.cpp:367-523); the worker's
    statistics are reset by it, so a second source decoded by the same engine starts like the oracle's."""
    emu = emu_lib
    luma, _, _ = synth.stc007_frames(n_frames=3, seed=21, height=48, noise_sigma=3.0)
    luma2, _, _ = synth.stc007_frames(n_frames=2, seed=22, height=48, x0=20, x1=690)
    lib = libs.load_oracle()
    lib.orc_v2d_new.restype = C.c_void_p
    h = C.c_void_p(lib.orc_v2d_new())
    lib.orc_v2d_set_mode.argtypes = [C.c_void_p, C.c_int]
    lib.orc_v2d_set_mode(h, 2)
    want1, ws1 = oracle_binarize(luma, handle=h, new_file=True, end_file=True)
    want2, ws2 = oracle_binarize(luma2, handle=h, new_file=True, first_frame_no=1)
    eng = C.c_void_p(emu.sdv_engine_create(0))
    emu.sdv_set_mode(eng, 2)
    got1, gs1, _ = emu_run(emu, luma, 2, flags=1 | 4, eng=eng)
    got2, gs2, _ = emu_run(emu, luma2, 2, flags=1, eng=eng)
    emu.sdv_engine_destroy(eng)
    assert len(got1) == 3 * 51 + 1 + 52 and (got1["service_type"][-52:] != 0).all()
    assert got1.tobytes() == want1.tobytes(), golden_cases.diff_report(got1, want1)
    assert gs1.view(np.uint8).tobytes() == ws1.tobytes()
    assert got2.tobytes() == want2.tobytes(), golden_cases.diff_report(got2, want2)
    assert gs2.view(np.uint8).tobytes() == ws2.tobytes()


@pytest.mark.parametrize("width,height,row_stride,misalign", [(720, 47, None, 0), (721, 40, None, 0), (712, 42, 720, 0), (720, 40, 733, 0),
                                                             (720, 40, None, 3), (736, 38, 752, 16)])
def test_emu_ragged_geometry(emu_lib, oracle_lib, width, height, row_stride, misalign):
    """Odd heights, widths that are not a multiple of 16, padded rows, buffers that do not start 16-byte aligned: the vector
    staging of the frame loop has to fall back to its byte path (or not) without changing a record."""
    emu = emu_lib
    even = height + (height & 1)
    luma, _, _ = synth.stc007_frames(n_frames=3, seed=31, width=width, height=even, lines_per_field=even // 2 + 2, noise_sigma=3.0)
    luma = np.ascontiguousarray(luma[:, :height])               # an odd height: the second field is one row shorter
    want, want_stats = oracle_binarize(luma, mode=2)
    eng = C.c_void_p(emu.sdv_engine_create(0))
    emu.sdv_set_mode(eng, 2)
    rc, got, got_stats = engine_api.emu_binarize(emu, eng, luma, row_stride=row_stride, misalign=misalign)
    emu.sdv_engine_destroy(eng)
    assert rc == 0
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()


def test_emu_batch_path_corner_cases(emu_lib, oracle_lib):
    """Exercises the line-batch fast path: duplicated rows (dup-line rule), silent audio (almost-silent lines are
    exempt from it), a bad row in the middle of a batch, doubled-width sources, dup check switched off."""
    luma, _, _ = synth.stc007_frames(3, seed=5, height=80)
    luma = luma.copy()
    luma[1, 20] = luma[1, 18]            # same field: line copies the previous one
    luma[1, 41] = luma[1, 39]
    luma[2, 30] = 20                     # dropout inside a batch
    for kw in (dict(mode=2), dict(mode=1, check_line_dup=False)):
        want, want_stats = oracle_binarize(luma, **kw)
        eng = C.c_void_p(emu_lib.sdv_engine_create(0))
        emu_lib.sdv_set_mode(eng, kw["mode"])
        emu_lib.sdv_set_check_line_dup(eng, int(kw.get("check_line_dup", True)))
        got, got_stats, _ = emu_run(emu_lib, luma, kw["mode"], eng=eng)
        emu_lib.sdv_engine_destroy(eng)
        assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
        assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
    luma, _, _ = synth.stc007_frames(2, seed=6, height=60, silent=True)
    want, want_stats = oracle_binarize(luma, mode=2)
    got, got_stats, _ = emu_run(emu_lib, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    luma, _, _ = synth.stc007_frames(2, seed=7, height=40, width=1440, x0=24, x1=1416)
    want, want_stats = oracle_binarize(luma, mode=1, doubled=True, m2=True)
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    emu_lib.sdv_set_pcm_type(eng, 2, 1)
    emu_lib.sdv_set_mode(eng, 1)
    rc, got, got_stats = engine_api.emu_binarize(emu_lib, eng, luma, first_frame_no=1, flags=3)
    emu_lib.sdv_engine_destroy(eng)
    assert rc == 0 and got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()


def test_emu_golden_rough_draft(emu_lib):
    mode, luma, want, want_stats = golden_cases.load("ntsc_rough_draft")
    got, got_stats, _ = emu_run(emu_lib, luma, mode)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()


def test_emu_speculation_rounds(emu_lib, oracle_lib):
    """Clean stream: cold frame alone, then every remaining frame in ONE parallel round; the chain state
    is carried across calls."""
    luma, _, _ = synth.stc007_frames(6, seed=9, height=32)
    want, want_stats = oracle_binarize(luma, mode=1)
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    emu_lib.sdv_set_mode(eng, 1)
    a, sa, ia = emu_run(emu_lib, luma[:4], 1, flags=1, first=1, eng=eng)
    assert ia.rounds == 2 and ia.frames_launched == 4
    b, sb, ib = emu_run(emu_lib, luma[4:], 1, flags=0, first=5, eng=eng)
    assert ib.rounds == 1 and ib.frames_launched == 2
    emu_lib.sdv_engine_destroy(eng)
    got = np.concatenate([a, b])
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert np.concatenate([sa, sb]).view(np.uint8).tobytes() == want_stats.tobytes()


def test_emu_misprediction_is_repaired(emu_lib, oracle_lib):
    """A disturbance in the middle of the batch changes the tuning the later frames inherit: the first
    parallel round mispredicts, the engine re-decodes from the break and still matches the oracle."""
    luma, _, _ = synth.stc007_frames(6, seed=12, height=32)
    luma = luma.copy()
    luma[3:] = np.roll(luma[3:], 7, axis=2)                  # the data window moves by 7 px from frame 3 on
    want, want_stats = oracle_binarize(luma, mode=1)
    got, got_stats, info = emu_run(emu_lib, luma, 1)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
    assert info.rounds >= 3 and info.frames_launched > 6


@pytest.mark.parametrize("seed", [61, 62, 63])
def test_emu_random_dropouts_and_anchors(emu_lib, oracle_lib, seed):
    """Lost lines and a moving data window at random frames of a 14-frame tape: several links of the chain break in the first round,
    every one becomes an anchor, given-up frames go to the full kernel together - and the records equal the sequential decode."""
    rng = np.random.default_rng(seed)
    n = 14
    luma, _, _ = synth.stc007_frames(n, seed=seed, height=40, noise_sigma=3.0)
    luma = luma.copy()
    for f in rng.choice(np.arange(1, n), size=4, replace=False):
        luma[int(f), rng.integers(2, 38, size=int(rng.integers(1, 3)))] = 16
    shift_at = int(rng.integers(4, n - 2))
    luma[shift_at:] = np.roll(luma[shift_at:], int(rng.integers(3, 9)), axis=2)
    want, want_stats = oracle_binarize(luma, mode=2)
    got, got_stats, info = emu_run(emu_lib, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
    assert info.rounds <= 12, info.rounds


def test_emu_window_jump_settles_in_few_rounds(emu_lib, oracle_lib):
    """The data window moves in the middle of a 40-frame batch.  Every link behind the jump breaks (the 16-frame coordinate history
    of the predicted states is the old window's), but only over coordinates: the frames behind the jump are predicted again from the
    first real state behind it, not settled one per round."""
    n = 40
    luma, _, _ = synth.stc007_frames(n, seed=77, height=24, noise_sigma=2.0)
    luma = luma.copy()
    luma[9:] = np.roll(luma[9:], 6, axis=2)
    want, want_stats = oracle_binarize(luma, mode=1)
    got, got_stats, info = emu_run(emu_lib, luma, 1)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
    assert info.rounds <= 10, info.rounds


def test_emu_crowd_over_several_windows_is_led_by_the_first_frame_of_each(emu_lib, oracle_lib, monkeypatch):
    """The data window jumps five times in a call, every time out of reach of the state the frames were started from: behind the first jump every lean
    wave gives up - one crowd over five windows.  The waves leave where the line they gave up on began (FrameArgs::sig) and the scheduler sends the first frame
    of every window to the general kernel at once, not the crowd's first frame only (which finds the next window two rounds later, and so on): fewer rounds,
    fewer frames through the general kernel, the same records as the sequential oracle either way."""
    n = 160
    luma0, _, _ = synth.stc007_frames(n, seed=5, height=24, noise_sigma=3.0)
    luma = luma0.copy()
    for f, to in [(30, 6), (55, -7), (80, 4), (105, -5), (130, 8)]:
        luma[f:] = np.roll(luma0[f:], to, axis=2)
    want, want_stats = oracle_binarize(np.concatenate([luma0[:20], luma]), mode=2)
    seen = {}
    for switch in (None, "SDV_SCHED_NO_SIG"):
        if switch: monkeypatch.setenv(switch, "1")
        eng = C.c_void_p(emu_lib.sdv_engine_create(0))
        emu_lib.sdv_set_mode(eng, 2)
        a, sa, _ = emu_run(emu_lib, luma0[:20], 2, eng=eng)
        b, sb, info = emu_run(emu_lib, luma, 2, flags=0, first=21, eng=eng)
        emu_lib.sdv_engine_destroy(eng)
        if switch: monkeypatch.delenv(switch)
        got = np.concatenate([a, b])
        assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
        assert np.concatenate([sa, sb]).view(np.uint8).tobytes() == want_stats.tobytes()
        seen[switch] = (info.rounds, info.frames_general)
    assert seen[None][0] < seen["SDV_SCHED_NO_SIG"][0] and seen[None][1] < seen["SDV_SCHED_NO_SIG"][1], seen
    assert seen[None][0] <= 8, seen


@pytest.mark.parametrize("n,height,seed", [(30, 64, 3), (16, 200, 9)])
def test_emu_a_pass_that_meets_the_last_one_changes_nothing(emu_lib, oracle_lib, monkeypatch, n, height, seed):
    """A tape with lost lines and inverted cells in every frame: every frame goes through the general kernel several times (guessed state, sweeps settled,
    predecessor's real state).  A later pass that reaches a line with the state the frame's last complete pass had there ends on it (stc007_device.h, TcSnap):
    records, frame descriptors and the chain are those of the sequential oracle with and without that short cut, and the short cut is taken."""
    luma0, _, _ = synth.stc007_frames(n, seed=seed, height=height, noise_sigma=4.0)
    lum = luma0.copy()
    lum[:, 16::17, :] = 16
    flat = lum.reshape(-1, 720)
    rng = np.random.default_rng(53)
    rows = np.arange(0, flat.shape[0], 11)
    xs = 12 + (rng.integers(4, 132, size=rows.shape) * (720 - 24)) // 137
    for dx in range(5):
        flat[rows, xs + dx] = np.clip(230 - flat[rows, xs + dx].astype(np.int16), 0, 255).astype(np.uint8)
    want, want_stats = oracle_binarize(np.concatenate([luma0[:8], lum]), mode=2)
    met = {}
    monkeypatch.setenv("SDV_NO_FAT", "1")       # (rounds of a few frames take the build without snapshots that settles its sweeps itself: not what is looked at here)
    for switch in (None, "SDV_NO_TC"):
        if switch: monkeypatch.setenv(switch, "1")
        eng = C.c_void_p(emu_lib.sdv_engine_create(0))
        emu_lib.sdv_set_mode(eng, 2)
        a, sa, _ = emu_run(emu_lib, luma0[:8], 2, eng=eng)
        b, sb, info = emu_run(emu_lib, lum, 2, flags=0, first=9, eng=eng)
        emu_lib.sdv_engine_destroy(eng)
        if switch: monkeypatch.delenv(switch)
        got = np.concatenate([a, b])
        assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
        assert np.concatenate([sa, sb]).view(np.uint8).tobytes() == want_stats.tobytes()
        met[switch] = (info.frames_met, info.rounds, info.frames_general)
    assert met[None][0] > 0 and met["SDV_NO_TC"][0] == 0, met
    assert met[None][1:] == met["SDV_NO_TC"][1:], met           # (the short cut changes what a pass costs, not what the scheduler sees)


def test_emu_cold_chain_settles_its_first_sweep_in_one_pass(emu_lib, oracle_lib, monkeypatch):
    """The first frame of a cold chain (nothing tuned: its first line goes through the reference-level sweep) is decoded alone, by the kernel that settles
    the sweep while the frame waits: one round for it, one for the rest - and one more for the first frame without that kernel."""
    luma, _, _ = synth.stc007_frames(4, seed=14, height=64, noise_sigma=3.0)
    want, want_stats = oracle_binarize(luma, mode=2)
    seen = {}
    for switch in (None, "SDV_NO_FAT"):
        if switch: monkeypatch.setenv(switch, "1")
        got, got_stats, info = emu_run(emu_lib, luma, 2)
        if switch: monkeypatch.delenv(switch)
        assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
        assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
        seen[switch] = (info.rounds, info.sweeps)
    assert seen[None] == (2, seen["SDV_NO_FAT"][1]) and seen["SDV_NO_FAT"][0] == 3 and seen[None][1] >= 1, seen


def test_emu_small_rounds_settle_their_sweeps_themselves(emu_lib, oracle_lib, monkeypatch):
    """A round of a few frames goes through sdv_k_stc007_frames_fat: a frame that misses a reference-level sweep has it settled on the spot (by the four waves
    beside its own on the GPU, by its own wave here) and goes on with the outcome - its pass is complete, no round to decode it again.  Records and frame
    descriptors are the sequential oracle's either way, and the rounds are fewer."""
    from test_gpu_parity import _unreadable_cells
    luma, _, _ = synth.stc007_frames(16, seed=78, noise_sigma=4.0, height=120, lines_per_field=60)
    luma = _unreadable_cells(luma, every=23)
    luma[:, 50::31, :] = 16
    want, want_stats = oracle_binarize(np.ascontiguousarray(luma), mode=2)
    seen = {}
    for switch in (None, "SDV_NO_FAT"):
        if switch: monkeypatch.setenv(switch, "1")
        got, got_stats, info = emu_run(emu_lib, luma, 2)
        if switch: monkeypatch.delenv(switch)
        assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
        assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
        seen[switch] = (info.rounds, info.frames_general, info.sweeps)
    assert seen[None][0] < seen["SDV_NO_FAT"][0] and seen[None][1] < seen["SDV_NO_FAT"][1], seen
    assert seen[None][2] > 40, seen


@pytest.mark.parametrize("shift", [-2, 3])
def test_emu_tape_that_sits_on_a_later_shift_stage(emu_lib, oracle_lib, shift):
    """A tape that moves a few pixels to the side and stays there: the lines go on reading with the coordinates the binarizer holds (a line that reads hands them
    on), on a later shift stage.  The lean kernel's batches then park the masks of every stage up to that one per line and solve a lane per line
    (stc007_device.h: rung_hint); a line with a lost cell in between ends such a batch.  Against the sequential oracle."""
    n = 14
    luma, _, _ = synth.stc007_frames(n, seed=611, height=96, noise_sigma=4.0)
    luma = luma.copy()
    luma[3:] = np.roll(luma[3:], shift, axis=2)
    luma[7, 30, 200:206] = 255 - luma[7, 30, 200:206]          # a line that reads on no stage, in the middle of a batch
    luma[9, 51] = 16
    want, want_stats = oracle_binarize(luma, mode=2)
    data = want[(want["service_type"] == 0) & (want["frame_number"] > 4)]
    assert int((data["shift_stage"] != 0).sum()) > len(data) * 8 // 10
    got, got_stats, info = emu_run(emu_lib, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()


def test_emu_general_kernel_on_a_later_shift_stage(emu_lib, oracle_lib):
    """The same regime in the general kernel: frames with a lost line near their top go there, and the 230 lines behind it sit on the later stage - the
    general build takes its batches the same way once 48 lines in a row have read (stc007_device.h: calm_lines)."""
    n = 8
    luma, _, _ = synth.stc007_frames(n, seed=612, height=240, noise_sigma=4.0)
    luma = luma.copy()
    luma[2:] = np.roll(luma[2:], -2, axis=2)
    luma[4, 3] = 16; luma[6, 8] = 16; luma[6, 150] = 16
    want, want_stats = oracle_binarize(luma, mode=2)
    got, got_stats, info = emu_run(emu_lib, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
    assert info.frames_general >= 2


def test_emu_frames_the_model_gives_their_old_state_are_not_decoded_again(emu_lib, oracle_lib, monkeypatch):
    """Repair rounds predict every frame behind an anchor that changed again; beyond the reach of the histories most frames get the state they were last decoded
    from, and the frame kernels only look at what their successor was started from (v2d_relink).  The same records with and without that short cut."""
    n = 120
    luma0, _, _ = synth.stc007_frames(n, seed=9, height=24, noise_sigma=3.0)
    luma = luma0.copy()
    for f, to in [(25, 6), (50, -7), (90, 5)]:
        luma[f:] = np.roll(luma0[f:], to, axis=2)
    want, want_stats = oracle_binarize(np.concatenate([luma0[:20], luma]), mode=2)
    seen = {}
    for switch in (None, "SDV_SCHED_NO_SKIP"):
        if switch: monkeypatch.setenv(switch, "1")
        eng = C.c_void_p(emu_lib.sdv_engine_create(0))
        emu_lib.sdv_set_mode(eng, 2)
        a, sa, _ = emu_run(emu_lib, luma0[:20], 2, eng=eng)
        b, sb, info = emu_run(emu_lib, luma, 2, flags=0, first=21, eng=eng)
        emu_lib.sdv_engine_destroy(eng)
        if switch: monkeypatch.delenv(switch)
        got = np.concatenate([a, b])
        assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
        assert np.concatenate([sa, sb]).view(np.uint8).tobytes() == want_stats.tobytes()
        seen[switch] = (info.rounds, info.frames_launched)
    assert seen[None] == seen["SDV_SCHED_NO_SKIP"], seen


def test_emu_worn_tape_without_meetings_takes_the_plain_general_kernel(emu_lib, oracle_lib):
    """A tape with an unreadable cell in every fifth line: every frame goes through the general kernel, and hardly a decode meets the frame's last pass (each
    damaged line re-tunes the binarizer for good).  The engine sees that in the first call and gives the calls behind it to the build of the general kernel
    without snapshots (engine.inc, plain_general; stc007_device.h, kMeet) - the records stay the sequential oracle's."""
    n, h = 12, 96
    luma0, _, _ = synth.stc007_frames(4 * n, seed=31, height=h, noise_sigma=4.0)
    lum = luma0.copy()
    flat = lum.reshape(-1, 720)
    rng = np.random.default_rng(5)
    rows = np.arange(3, flat.shape[0], 5)
    xs = 12 + (rng.integers(4, 132, size=rows.shape) * (720 - 24)) // 137
    for dx in range(5):
        flat[rows, xs + dx] = np.clip(230 - flat[rows, xs + dx].astype(np.int16), 0, 255).astype(np.uint8)
    want, want_stats = oracle_binarize(lum, mode=2)
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    emu_lib.sdv_set_mode(eng, 2)
    got, gs, met, general = [], [], [], []
    for k in range(4):
        b, sb, info = emu_run(emu_lib, lum[k * n:(k + 1) * n], 2, flags=1 if k == 0 else 0, first=1 + k * n, eng=eng)
        got.append(b); gs.append(sb); met.append(info.frames_met); general.append(info.frames_general)
    emu_lib.sdv_engine_destroy(eng)
    got = np.concatenate(got)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert np.concatenate(gs).view(np.uint8).tobytes() == want_stats.tobytes()
    assert all(g >= n for g in general), general
    assert met[1:] == [0, 0, 0], met            # (the build without snapshots meets nothing)


def test_emu_bad_arguments(emu_lib):
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    buf = np.zeros((1, 8, 200), np.uint8)
    recs = np.zeros(11, dtype=libs.LINE_DTYPE)
    st = np.zeros(32, np.uint8)
    f = emu_lib.sdv_binarize_frames
    assert f(eng, None, 200, 1600, 200, 8, 1, 1, 0, recs.ctypes.data, 11, st.ctypes.data, 1, None) == 1      # SDV_ERR_NULL_VIDEO
    assert f(eng, buf.ctypes.data, 200, 1600, 200, 8, 1, 1, 0, None, 11, st.ctypes.data, 1, None) == 2         # SDV_ERR_NULL_PCM
    assert f(eng, buf.ctypes.data, 100, 800, 100, 8, 1, 1, 0, recs.ctypes.data, 11, st.ctypes.data, 1, None) == 3  # SHORT_LINE
    assert f(eng, buf.ctypes.data, 200, 1600, 200, 8, 0, 1, 0, recs.ctypes.data, 11, st.ctypes.data, 1, None) == -1  # BAD_ARG
    # the output capacities are checked against what the call will write (NEW_FILE: one more record)
    assert f(eng, buf.ctypes.data, 200, 1600, 200, 8, 1, 1, 1, recs.ctypes.data, 11, st.ctypes.data, 1, None) == -1
    assert b"12 line records" in emu_lib.sdv_last_error(eng)
    assert f(eng, buf.ctypes.data, 200, 1600, 200, 8, 1, 1, 0, recs.ctypes.data, 11, st.ctypes.data, 0, None) == -1
    # frames of a batch must not overlap
    buf2 = np.zeros((2, 8, 200), np.uint8); recs2 = np.zeros(22, dtype=libs.LINE_DTYPE); st2 = np.zeros(64, np.uint8)
    assert f(eng, buf2.ctypes.data, 200, 1000, 200, 8, 2, 1, 0, recs2.ctypes.data, 22, st2.ctypes.data, 2, None) == -1
    assert b"frame_stride" in emu_lib.sdv_last_error(eng)
    emu_lib.sdv_engine_destroy(eng)


@pytest.mark.parametrize("height,lpf", [(576, 294), (640, 320)])
def test_emu_tall_frames_keep_their_histories(emu_lib, oracle_lib, height, lpf):
    """PAL-size frames (more than 256 lines per field: five 64-line chunks of parked lines in the whole-frame capture) with a data window
    that moves half way: the 16-frame coordinate history decides what the frames behind the jump start from - records and frame
    descriptors equal the oracle's, and a tape that plays is decoded in one round per call."""
    n = 8
    luma, _, _ = synth.stc007_frames(n_frames=n, seed=41, height=height, lines_per_field=lpf, noise_sigma=3.0)
    moved, _, _ = synth.stc007_frames(n_frames=n, seed=41, height=height, lines_per_field=lpf, noise_sigma=3.0, x0=17, x1=713)
    tape = np.concatenate([luma[:5], moved[5:], luma[:4]])
    want, want_stats = oracle_binarize(tape, mode=2)
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    emu_lib.sdv_set_mode(eng, 2)
    got, got_stats, info = emu_run(emu_lib, tape, 2, eng=eng)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
    # a tape that plays, continued: every frame is predicted right
    steady, _, _ = synth.stc007_frames(n_frames=6, seed=42, height=height, lines_per_field=lpf, noise_sigma=3.0)
    emu_lib.sdv_reset_stream(eng)
    emu_run(emu_lib, steady, 2, eng=eng)
    _, _, info = emu_run(emu_lib, steady, 2, flags=0, first=7, eng=eng)
    emu_lib.sdv_engine_destroy(eng)
    assert info.rounds == 1 and info.frames_general == 0


def test_emu_worn_tape_mark_comes_and_goes(emu_lib, oracle_lib):
    """The scheduler starts a call on the full kernel when most frames of the last call took lines through the general path ("worn tape").
    A cold one-frame call must not set that mark (round 4: it did, and the mark never came off again - every later frame of a clean tape went
    through the full kernel), a call full of damaged frames sets it, a clean call takes it off again; the records equal the oracle's all the way."""
    from test_gpu_parity import _unreadable_cells
    clean, _, _ = synth.stc007_frames(30, seed=91, noise_sigma=3.0, height=96, lines_per_field=48)
    want, _ = oracle_binarize(clean, mode=2)
    per = 96 + 3
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    emu_lib.sdv_set_mode(eng, 2)
    got0, _, _ = emu_run(emu_lib, clean[:1], 2, eng=eng)                          # the cold frame alone
    got1, _, info = emu_run(emu_lib, clean[1:13], 2, flags=0, first=2, eng=eng)
    assert info.frames_general == 0, "a clean tape behind a one-frame cold call: nothing for the full kernel"
    assert np.concatenate([got0, got1]).tobytes() == want[:1 + 13 * per].tobytes()
    # damage in every frame: the lean kernel gives the frames up, the call ends marked
    worn = np.ascontiguousarray(_unreadable_cells(clean[13:25].copy(), every=7))
    want_w, _ = oracle_binarize(np.concatenate([clean[:13], worn, clean[25:]]), mode=2)
    got2, _, info2 = emu_run(emu_lib, worn, 2, flags=0, first=14, eng=eng)
    assert got2.tobytes() == want_w[1 + 13 * per:1 + 25 * per].tobytes()
    assert info2.frames_general > 0
    # ... so the next call starts on the full kernel; its frames are clean: the mark comes off, the call behind it is the lean kernel's again
    more, _, _ = synth.stc007_frames(40, seed=92, noise_sigma=3.0, height=96, lines_per_field=48)
    _, _, info3 = emu_run(emu_lib, more[:12], 2, flags=0, first=26, eng=eng)
    _, _, info4 = emu_run(emu_lib, more[12:24], 2, flags=0, first=38, eng=eng)
    emu_lib.sdv_engine_destroy(eng)
    assert info3.frames_general >= 12 and info4.frames_general == 0, (info3.frames_general, info4.frames_general)


@pytest.mark.parametrize("noise", [4.0, 0.0])
def test_emu_unreadable_cells_sweep_every_level(emu_lib, oracle_lib, noise):
    """A bit cell inverted on some lines: their reference level sweep runs over every level, the levels near white leave a zero source CRC word
    (two outcomes per level, chained through the lanes - stc007_device.h sweep_ref_level); the scheduler carries the level such a sweep settles on
    along the chain (engine.inc, "a level that passes through").  (Also on a tape without noise: two thirds of a sweep's levels then lie in the gap between the dark
    and the bright pixels and all come out alike.)"""
    from test_gpu_parity import _unreadable_cells
    luma, _, _ = synth.stc007_frames(6, seed=78, noise_sigma=noise, height=120, lines_per_field=60)
    luma = _unreadable_cells(luma, every=23)
    want, want_stats = oracle_binarize(np.ascontiguousarray(luma), mode=2)
    got, got_stats, info = emu_run(emu_lib, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
    assert int((((got["flags"] & 64) == 0) & (got["service_type"] == 0)).sum()) > 6      # data lines without SDV_LF_CRC_VALID: each of them was swept


def test_emu_crowd_waits_for_the_sweeps_of_its_first_frame(emu_lib, oracle_lib):
    """Every frame of a full-size tape has lines no level reads: the lean kernel gives all of them up at once (a crowd), the first one goes to the
    full kernel alone - and comes back as given up once more, because the reference-level sweeps it needs are only asked for by that pass
    (stc007_sweep_device.h).  The crowd behind it has to go on waiting until those are settled: taking the unsettled pass for the frame's outcome
    left its lines with the levels of a sweep that "found nothing"."""
    from test_gpu_parity import _unreadable_cells
    luma, _, _ = synth.stc007_frames(40, seed=77, noise_sigma=4.0)
    luma = np.ascontiguousarray(_unreadable_cells(luma)[:12])
    want, want_stats = oracle_binarize(luma, mode=2)
    got, got_stats, info = emu_run(emu_lib, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.view(np.uint8).tobytes() == want_stats.tobytes()
    assert info.sweeps > 50


def _jittered_tape(n, seed, jit, **kw):
    """A tape whose lines sit a few pixels beside the coordinates the worker keeps preset: from the third frame on every row is moved by its own
    -jit..jit pixels - the lines read, but on another rung of the hysteresis x shift ladder than the first."""
    luma, _, _ = synth.stc007_frames(n, seed=seed, **kw)
    luma = luma.copy()
    rng = np.random.default_rng(seed)
    for f in range(2, n):
        for r in range(luma.shape[1]):
            luma[f, r] = np.roll(luma[f, r], int(rng.integers(-jit, jit + 1)))
    return luma


@pytest.mark.parametrize("seed,jit,height", [(501, 2, 160), (502, 3, 160), (503, 3, 486)])
def test_emu_lines_that_read_on_other_rungs_of_the_ladder(emu_lib, oracle_lib, seed, jit, height):
    """The batches of the frame loop walk the ladder themselves once lines need it (sticky_rung, stc007_device.h): records as the sequential worker's."""
    luma = _jittered_tape(5, seed, jit, height=height, noise_sigma=3.0)
    want, want_stats = oracle_binarize(luma, mode=2)
    data = want[want["service_type"] == 0]
    assert int(((data["shift_stage"] != 0) | (data["hysteresis_depth"] != 0)).sum()) > len(data) // 12, "the tape is meant to need the ladder on many lines"
    got, stats, info = emu_run(emu_lib, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert stats.view(np.uint8).tobytes() == want_stats.tobytes()
