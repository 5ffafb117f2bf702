"""GPU parity tests: the HIP path through the C-ABI vs the committed reference goldens and vs the oracle."""
import numpy as np
import pytest

import golden_cases
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def gpu_run(torch, luma, mode, new_file=True, first=1, eng=None, **kw):
    from sdvpcmdecoder_amd import Engine, LINE_DTYPE
    own = eng is None
    if own:
        eng = Engine(0)
        eng.setBinarizationMode(mode)
    d = torch.from_numpy(np.ascontiguousarray(luma)).to("cuda:0")
    lines, stats = eng.binarize_frames(d, first_frame_no=first, new_file=new_file, **kw)
    torch.cuda.synchronize()
    info = eng.run_info()
    recs = lines.cpu().numpy().view(LINE_DTYPE).reshape(-1)
    st = stats.cpu().numpy()
    if own:
        eng.close()
    return recs, st, info


@pytest.mark.parametrize("name", list(golden_cases.CASES))
def test_hip_matches_reference_golden(torch_cuda, name):
    mode, luma, want, want_stats = golden_cases.load(name)
    got, got_stats, _ = gpu_run(torch_cuda, luma, mode)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_hip_vs_oracle_degraded(torch_cuda, mode):
    cases = [dict(n_frames=6, seed=21, noise_sigma=8.0, blur=1),
             dict(n_frames=4, seed=22, noise_sigma=18.0, blur=2, height=120 if mode >= 2 else 486),
             dict(n_frames=3, seed=23, black=70, white=130, noise_sigma=5.0, height=200),
             dict(n_frames=3, seed=24, width=1440, x0=24, x1=1416, height=100),
             dict(n_frames=3, seed=25, width=360, x0=6, x1=354, height=100)]
    for kw in cases:
        luma, _, _ = synth.stc007_frames(**kw)
        want, want_stats = oracle_binarize(luma, mode=mode)
        got, got_stats, _ = gpu_run(torch_cuda, luma, mode)
        assert got.tobytes() == want.tobytes(), (kw, golden_cases.diff_report(got, want))
        assert got_stats.tobytes() == want_stats.tobytes()


def test_hip_jitter_and_dropouts(torch_cuda):
    rng = np.random.default_rng(31)
    luma, _, _ = synth.stc007_frames(8, seed=31, noise_sigma=5.0, blur=1)
    luma = luma.copy()
    for f in range(8):
        rows = rng.integers(0, 486, 12)
        for r in rows:
            luma[f, r] = np.roll(luma[f, r], int(rng.integers(-3, 4)))
        luma[f, rng.integers(0, 486, 4)] = 16
    want, want_stats = oracle_binarize(luma, mode=2)
    got, got_stats, info = gpu_run(torch_cuda, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()


def test_hip_sparse_dropouts_settle_in_few_rounds(torch_cuda):
    """A long tape with a lost line now and then: every dropout frame is given up by the lean kernel and re-tunes the chain behind it.
    The output equals the sequential decode and the number of rounds stays small (anchors at every broken link, engine.inc)."""
    n = 400
    luma, _, _ = synth.stc007_frames(n, seed=77, noise_sigma=4.0)
    luma = luma.copy()
    rng = np.random.default_rng(78)
    frames = np.sort(rng.choice(np.arange(20, n - 5), size=24, replace=False))
    for f in frames:
        luma[f, int(rng.integers(40, 440))] = 16
    want, want_stats = oracle_binarize(luma, mode=2)
    got, got_stats, info = gpu_run(torch_cuda, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()
    assert info.rounds <= 12, info.rounds                     # cold first frame + first pass + the given-up frames (twice: their sweeps) + a handful
    assert 24 <= info.frames_general <= 6 * 24 + 2, info.frames_general


def test_hip_stream_continuation_and_rounds(torch_cuda):
    from sdvpcmdecoder_amd import Engine
    luma, _, _ = synth.stc007_frames(64, seed=41)
    want, want_stats = oracle_binarize(luma, mode=2)
    eng = Engine(0)
    eng.setBinarizationMode(2)
    a, sa, ia = gpu_run(torch_cuda, luma[:40], 2, new_file=True, first=1, eng=eng)
    b, sb, ib = gpu_run(torch_cuda, luma[40:], 2, new_file=False, first=41, eng=eng)
    eng.close()
    assert ia.rounds == 2 and ib.rounds == 1, (ia.rounds, ib.rounds)      # the cold frame alone (the sweep its first line needs is settled while it waits: sdv_k_stc007_frames_fat), then the rest at once
    got = np.concatenate([a, b])
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert np.concatenate([sa, sb]).tobytes() == want_stats.tobytes()


def test_hip_doubled_and_m2_and_presets(torch_cuda):
    from sdvpcmdecoder_amd import Engine
    import libs
    luma, _, _ = synth.stc007_frames(3, seed=51, width=1440, x0=24, x1=1416, height=80, noise_sigma=6.0)
    want, want_stats = oracle_binarize(luma, mode=1, doubled=True, m2=True, check_line_dup=False)
    eng = Engine(0)
    eng.setPCMType(3)           # TYPE_M2
    eng.setBinarizationMode(1)
    eng.setCheckLineDup(False)
    got, got_stats, _ = gpu_run(torch_cuda, luma, 1, eng=eng, doubled=True)
    eng.close()
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()
    # fine settings: markers required, tighter contrast
    p = libs.default_preset()
    p.en_good_no_marker = 0
    p.min_contrast = 40
    luma, _, _ = synth.stc007_frames(3, seed=52, height=80, noise_sigma=12.0, blur=2)
    want, want_stats = oracle_binarize(luma, mode=2, preset=p)
    eng = Engine(0)
    eng.setBinarizationMode(2)
    ep = eng.getDefaultFineSettings()
    ep.en_good_no_marker = 0
    ep.min_contrast = 40
    eng.setFineSettings(ep)
    got, got_stats, _ = gpu_run(torch_cuda, luma, 2, eng=eng)
    eng.close()
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)


def test_hip_full_size_properties(torch_cuda):
    """Size-independent properties on a big batch (oracle too slow to run here in full): every clean line must
    decode to the generator's words with a valid CRC, except the first line of each field (FIELD_UNSAFE rule,
    videotodigital.cpp:1159-1211), and the schedule must be 2 rounds (the cold first frame - the reference-level sweep its first line needs is settled by the
    waves beside its own while it waits, sdv_k_stc007_frames_fat - then all other frames at once)."""
    n = 512
    luma, w9, _ = synth.stc007_frames(n, seed=61)
    got, st, info = gpu_run(torch_cuda, luma, 2)
    assert info.rounds == 2 and info.frames_launched == n, (info.rounds, info.frames_launched)
    body = got[1:].reshape(n, 489)
    lines = np.concatenate([body[:, :243], body[:, 244:487]], axis=1)      # odd rows, even rows
    f = np.arange(n)[:, None]
    r = np.arange(243)[None, :]
    idx = np.concatenate([f * 490 + 2 + r, f * 490 + 245 + 2 + r], axis=1)
    assert (lines["words"] == w9[idx]).all()
    valid = (lines["flags"] & 64) != 0
    assert valid[:, 1:243].all() and valid[:, 244:].all() and not valid[:, 0].any() and not valid[:, 243].any()
    assert (body[:, 243]["service_type"] == 4).all() and (body[:, 487]["service_type"] == 4).all() and (body[:, 488]["service_type"] == 5).all()


@pytest.mark.parametrize("width,height,pad,shift", [(720, 487, 0, 0), (721, 120, 0, 0), (712, 120, 8, 0), (720, 120, 13, 0), (720, 120, 0, 5),
                                                     (736, 122, 16, 16)])
def test_hip_ragged_geometry(torch_cuda, width, height, pad, shift):
    """Odd heights, widths that are not a multiple of 16, padded rows (row_stride > width), buffers that do not start 16-byte
    aligned: same records as the oracle (the frame loop picks its byte path or its vector path per frame)."""
    torch = torch_cuda
    from sdvpcmdecoder_amd import Engine, LINE_DTYPE
    even = height + (height & 1)
    luma, _, _ = synth.stc007_frames(n_frames=3, seed=33, width=width, height=even, lines_per_field=even // 2 + 2, noise_sigma=3.0)
    luma = np.ascontiguousarray(luma[:, :height])
    want, want_stats = oracle_binarize(luma, mode=2)
    n = luma.shape[0]
    rs = width + pad
    buf = torch.full((n * height * rs + shift + 64,), 0x5A, dtype=torch.uint8, device="cuda:0")
    view = buf[shift:shift + n * height * rs].view(n, height, rs)[:, :, :width]
    view.copy_(torch.from_numpy(luma).to("cuda:0"))
    eng = Engine(0)
    eng.setBinarizationMode(2)
    lines, stats = eng.binarize_frames(view, first_frame_no=1, new_file=True)
    got = lines.cpu().numpy().view(LINE_DTYPE).reshape(-1)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert stats.cpu().numpy().tobytes() == want_stats.tobytes()


@pytest.mark.parametrize("seed", [51, 52])
def test_hip_dropouts_across_calls(torch_cuda, seed):
    """Lost lines at random places of a 160-frame tape that arrives in three calls: anchors, given-up frames and the hand-over between
    calls together reproduce the sequential decode."""
    from sdvpcmdecoder_amd import Engine
    rng = np.random.default_rng(seed)
    n = 160
    luma, _, _ = synth.stc007_frames(n, seed=seed, noise_sigma=4.0)
    luma = luma.copy()
    for f in rng.choice(np.arange(3, n), size=14, replace=False):
        luma[int(f), rng.integers(20, 460, size=int(rng.integers(1, 4)))] = 16
    want, want_stats = oracle_binarize(luma, mode=2)
    eng = Engine(0)
    eng.setBinarizationMode(2)
    cuts = [0, int(rng.integers(20, 70)), int(rng.integers(80, 140)), n]
    got, got_stats = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r, s, _ = gpu_run(torch_cuda, luma[a:b], 2, new_file=(a == 0), first=1 + a, eng=eng)
        got.append(r)
        got_stats.append(s)
    eng.close()
    got = np.concatenate(got)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert np.concatenate(got_stats).tobytes() == want_stats.tobytes()


def _unreadable_cells(luma, every=53, seed=53):
    """A bit cell inverted on one line in `every`: no reference level makes the CRC of such a line come out (the reference level sweep of
    MODE_NORMAL runs over every level for nothing), and the levels near white read sixteen zero bits as the source CRC word."""
    rng = np.random.default_rng(seed)
    out = luma.copy()
    flat = out.reshape(-1, out.shape[-1])
    w = flat.shape[1]
    for r in range(0, flat.shape[0], every):
        x = 12 + int(rng.integers(4, 132)) * (w - 24) // 137
        flat[r, x:x + 5] = np.clip(230 - flat[r, x:x + 5].astype(np.int16), 0, 255).astype(np.uint8)
    return out


@pytest.mark.parametrize("pal", [False, True])
def test_hip_unreadable_cells_sweep_every_level(torch_cuda, pal):
    """SURVEY 8d C3's kind of damage: lines whose sweep finds nothing, in every frame; the chain of the sweep's levels (a level that leaves a
    zero source CRC word sends the next one another way) is resolved from two outcomes per level - bit-exact with the sequential oracle, and the
    scheduler carries a reference level that passes through along the chain (rounds stay far below the frame count)."""
    kw = dict(height=576, lines_per_field=294) if pal else {}
    n = 40
    luma, _, _ = synth.stc007_frames(n, seed=77, noise_sigma=4.0, **kw)
    luma = _unreadable_cells(luma)
    want, want_stats = oracle_binarize(np.ascontiguousarray(luma), mode=2)
    got, got_stats, info = gpu_run(torch_cuda, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()
    unread = int((((got["flags"] & 64) == 0) & (got["service_type"] == 0)).sum())     # data lines without SDV_LF_CRC_VALID: each of them was swept
    assert unread > 5 * n, unread
    assert info.rounds <= 24, info.rounds


def test_hip_small_rounds_settle_their_sweeps_themselves(torch_cuda):
    """Rounds of a few frames run in sdv_k_stc007_frames_fat: the four waves beside a frame's own settle the reference-level sweep it misses while it waits at
    the workgroup's barrier, and the pass goes on with the outcome (stc007_sweep_device.h, fat_sweep).  Same tape as the emulator's test of that name: bit-exact
    with the sequential oracle, in the rounds the emulator needs with that kernel (13 without it)."""
    luma, _, _ = synth.stc007_frames(16, seed=78, noise_sigma=4.0, height=120, lines_per_field=60)
    luma = _unreadable_cells(luma, every=23)
    luma[:, 50::31, :] = 16
    want, want_stats = oracle_binarize(np.ascontiguousarray(luma), mode=2)
    got, got_stats, info = gpu_run(torch_cuda, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()
    assert info.sweeps > 40 and info.rounds <= 10, (info.sweeps, info.rounds)


@pytest.mark.parametrize("seed,jit", [(511, 2), (512, 3)])
def test_hip_lines_that_read_on_other_rungs_of_the_ladder(torch_cuda, seed, jit):
    """A tape whose rows sit a few pixels beside the preset coordinates (every row moved by its own -jit..jit pixels): the lines read on other rungs
    of the hysteresis x shift ladder; the batches of the frame loop then walk the ladder themselves (sticky_rung).  Against the sequential oracle."""
    n = 40
    luma, _, _ = synth.stc007_frames(n, seed=seed, noise_sigma=3.0)
    luma = luma.copy()
    rng = np.random.default_rng(seed)
    shifts = rng.integers(-jit, jit + 1, size=(n, luma.shape[1]))
    shifts[:2] = 0
    for f in range(2, n):
        for r in range(luma.shape[1]):
            if shifts[f, r]:
                luma[f, r] = np.roll(luma[f, r], int(shifts[f, r]))
    want, want_stats = oracle_binarize(luma, mode=2)
    data = want[want["service_type"] == 0]
    assert int(((data["shift_stage"] != 0) | (data["hysteresis_depth"] != 0)).sum()) > len(data) // 12
    got, got_stats, info = gpu_run(torch_cuda, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()


def _tape_beside_its_coordinates(n, shift, seed=611, jump_at=None, height=486, noise=4.0):
    """Frames that play, then the same tape `shift` pixels to the side: every line still reads with the coordinates the binarizer holds, on a later shift stage."""
    luma, _, _ = synth.stc007_frames(n, seed=seed, height=height, noise_sigma=noise)
    luma = luma.copy()
    at = 3 if jump_at is None else jump_at
    luma[at:] = np.roll(luma[at:], shift, axis=2)
    return luma


@pytest.mark.parametrize("shift", [-2, 3])
def test_hip_tape_that_sits_on_a_later_shift_stage(torch_cuda, shift):
    """The whole tape a few pixels beside its coordinates (which stay: a line that reads hands them on): the batches of the lean kernel take the lines with
    the masks of every shift stage up to the one in use parked per line and a lane per line solving them stage by stage.  Against the sequential oracle."""
    n = 24
    luma = _tape_beside_its_coordinates(n, shift)
    want, want_stats = oracle_binarize(luma, mode=2)
    data = want[(want["service_type"] == 0) & (want["frame_number"] > 4)]
    assert int((data["shift_stage"] != 0).sum()) > len(data) * 9 // 10
    got, got_stats, info = gpu_run(torch_cuda, luma, 2)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert got_stats.tobytes() == want_stats.tobytes()


def test_hip_window_jumps_far_and_near(torch_cuda):
    """Jumps of the data window out of reach of the coordinates (crowds of frames given up: a leader per window), jumps within reach (the lines read on another
    shift stage from then on) and lost lines in between, in calls of uneven length: records, descriptors and the chain equal the sequential oracle's."""
    from sdvpcmdecoder_amd import Engine
    n = 150
    luma0, _, _ = synth.stc007_frames(n, seed=615, height=120, noise_sigma=3.0)
    luma = luma0.copy()
    for f, to in [(20, 7), (41, 5), (60, -6), (75, -8), (99, 3), (120, 5)]:
        luma[f:] = np.roll(luma0[f:], to, axis=2)
    luma[33, 40] = 16; luma[88, 17::23] = 16
    want, want_stats = oracle_binarize(luma, mode=2)
    eng = Engine(0); eng.setBinarizationMode(2)
    got, st, met, skipped = [], [], 0, 0
    for lo, hi in [(0, 10), (10, 97), (97, 150)]:
        r, s_, info = gpu_run(torch_cuda, luma[lo:hi], 2, new_file=lo == 0, first=1 + lo, eng=eng)
        got.append(r); st.append(s_); met += info.frames_met
    eng.close()
    got = np.concatenate(got)
    assert got.tobytes() == want.tobytes(), golden_cases.diff_report(got, want)
    assert np.concatenate(st).tobytes() == want_stats.tobytes()
