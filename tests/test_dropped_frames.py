"""Dropped frames: VideoInFFMPEG::insertDummyFrame(false, true) sends the lines of a frame the video decoder
lost as empty VideoLines (vin_ffmpeg.cpp:367-522); Binarizer::processLine answers each with a silent line of invalid CRC
(binarizer.cpp:569-570, :1689-1700) and the worker books them as lines that did not read.  The C-ABI takes the marks per frame
(sdv_set_frame_flags, SDV_FRAME_EMPTY).
  oracle  vs  golden fixtures of the real VideoToDigital worker fed such frames (tests/golden/dropped_*.npz) and - when the reference
              build is loadable - the worker live;
  HIP kernel source  vs  the oracle, on the emulator and (-m gpu) through the C-ABI on the GPU, for the three formats."""
import ctypes as C
import os

import numpy as np
import pytest

import engine_api as ea
import libs
import pcm16_frames_api as f16
import pcm1_frames_api as f1
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name: (format, frames, generator keywords, mode, dropped frames, flags)
CASES = {
    "stc007_normal": ("stc007", 9, dict(seed=5, height=120, lines_per_field=62, noise_sigma=4.0), 2, (3, 4, 7), dict(new_file=True, end_file=True)),
    "stc007_fast_first_frames": ("stc007", 8, dict(seed=6, height=96, lines_per_field=50, noise_sigma=3.0), 1, (0, 1, 5), dict(new_file=True)),
    "stc007_long_gap": ("stc007", 16, dict(seed=7, height=64, lines_per_field=34, noise_sigma=3.0), 2, tuple(range(3, 13)), dict(new_file=True, end_file=True)),
    "pcm1_normal": ("pcm1", 7, dict(seed=9, height=60, noise_sigma=3.0), 2, (2, 3, 5), dict(new_file=True, end_file=True)),
    "pcm1_draft": ("pcm1", 7, dict(seed=10, height=60, noise_sigma=3.0), 0, (1, 6), dict(new_file=True)),
    "pcm16_normal": ("pcm16", 7, dict(seed=11, height=60, noise_sigma=3.0), 2, (2, 3, 5), dict(new_file=True, end_file=True)),
    "pcm16_fast": ("pcm16", 6, dict(seed=12, height=60, noise_sigma=3.0), 1, (0, 4), dict(new_file=True)),
}


def make_case(name):
    fmt, n, kw, mode, dropped, fl = CASES[name]
    if fmt == "stc007":
        luma = synth.stc007_frames(n_frames=n, **kw)[0]
    elif fmt == "pcm1":
        luma = synth.pcm1_frames(n, **kw)[0]
    else:
        luma = synth.pcm16x0_frames(n, **kw)[0]
    luma = luma.copy()
    mask = np.zeros(n, dtype=np.uint8)
    mask[list(dropped)] = 1
    luma[mask != 0] = 0xA5              # whatever the buffer holds for a dropped frame is not looked at
    return fmt, np.ascontiguousarray(luma), mode, mask, fl


def run_cpu(lib, prefix, name):
    fmt, luma, mode, mask, fl = make_case(name)
    setter = getattr(lib, prefix + "set_empty_frames")
    setter.argtypes = [C.c_void_p, C.c_size_t]
    setter(mask.ctypes.data, len(mask))
    if fmt == "stc007":
        if prefix == "orc_":
            return oracle_binarize(luma, mode=mode, new_file=fl.get("new_file", False), end_file=fl.get("end_file", False))
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
        mg = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mg)
        return mg.run_ref(luma, mode, new_file=int(fl.get("new_file", False)), end_file=int(fl.get("end_file", False)))
    api = f1 if fmt == "pcm1" else f16
    return api.run_cpu(lib, prefix, luma, mode, dict(fl))


def _bytes(recs, stats):
    return np.ascontiguousarray(recs).tobytes(), np.ascontiguousarray(stats).view(np.uint8).tobytes()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "dropped_" + name + ".npz"))
    recs, stats = run_cpu(oracle_lib, "orc_", name)
    r, s = _bytes(recs, stats)
    assert r == z["recs"].tobytes() and s == z["stats"].tobytes()
    # every line of a dropped frame is a silent line that did not read
    fmt, luma, mode, mask, fl = make_case(name)
    first = 1
    for f in np.nonzero(mask)[0]:
        sel = (recs["frame_number"] == first + f) & (recs["service_type"] == 0)
        assert sel.sum() == luma.shape[1] * (3 if fmt == "pcm16" else 1)
        assert ((recs["flags"][sel] & 64) == 0).all()


@pytest.mark.ref
@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_live_reference(name, oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    assert _bytes(*run_cpu(oracle_lib, "orc_", name)) == _bytes(*run_cpu(libs.load_ref(), "ref_", name))


def _engine_run(lib, eng, name, device=False):
    fmt, luma, mode, mask, fl = make_case(name)
    lib.sdv_set_frame_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_set_mode.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_set_mode(eng, mode)
    assert lib.sdv_set_frame_flags(eng, mask.ctypes.data, len(mask)) == 0
    if fmt == "stc007":
        rc, recs, stats = ea.emu_binarize(lib, eng, luma, first_frame_no=1, flags=(1 if fl.get("new_file") else 0) | (4 if fl.get("end_file") else 0))
    else:
        api = f1 if fmt == "pcm1" else f16
        rc, recs, stats = api.run_engine(lib, eng, luma, mode, dict(fl), configure=True)
        assert lib.sdv_set_frame_flags(eng, None, 0) == 0
    assert rc == 0, lib.sdv_last_error(eng)
    return recs, stats


@pytest.mark.parametrize("name", sorted(CASES))
def test_emu_matches_oracle(name, emu_lib, oracle_lib):
    lib = ea.bind(emu_lib)
    eng = C.c_void_p(lib.sdv_engine_create(0))
    fmt, luma, mode, mask, fl = make_case(name)
    if fmt != "stc007":
        # configure first: run_engine's configure would come after the marks were set otherwise (the marks belong to the next frame call only)
        pass
    got = _engine_run(lib, eng, name)
    lib.sdv_engine_destroy(eng)
    assert _bytes(*got) == _bytes(*run_cpu(oracle_lib, "orc_", name))


def test_emu_marks_are_consumed_by_one_call(emu_lib, oracle_lib):
    """The marks belong to the next frame call only: the call after it decodes its pixels."""
    lib = ea.bind(emu_lib)
    eng = C.c_void_p(lib.sdv_engine_create(0))
    lib.sdv_set_frame_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_set_mode(eng, 2)
    luma = synth.stc007_frames(n_frames=4, seed=31, height=64, lines_per_field=34, noise_sigma=3.0)[0]
    mask = np.array([0, 1, 0, 0], dtype=np.uint8)
    assert lib.sdv_set_frame_flags(eng, mask.ctypes.data, 4) == 0
    rc, recs1, _ = ea.emu_binarize(lib, eng, luma, first_frame_no=1, flags=1)
    rc2, recs2, _ = ea.emu_binarize(lib, eng, luma, first_frame_no=5, flags=0)
    lib.sdv_engine_destroy(eng)
    assert rc == 0 and rc2 == 0
    ok1 = [int(((recs1["frame_number"] == f) & ((recs1["flags"] & 64) != 0)).sum()) for f in (1, 2, 3, 4)]
    ok2 = [int(((recs2["frame_number"] == f) & ((recs2["flags"] & 64) != 0)).sum()) for f in (5, 6, 7, 8)]
    assert ok1[1] == 0 and min(ok1[0], ok1[2], ok1[3]) > 50 and min(ok2) > 50


# ---- the product on the GPU ------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_golden_from_reference(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    z = np.load(os.path.join(GOLD, "dropped_" + name + ".npz"))
    fmt, luma, mode, mask, fl = make_case(name)
    eng = Engine(0)
    eng.setBinarizationMode(mode)
    eng.set_frame_flags(mask)
    d = torch.from_numpy(luma).cuda()
    call = {"stc007": eng.binarize_frames, "pcm1": eng.pcm1_binarize_frames, "pcm16": eng.pcm16x0_binarize_frames}[fmt]
    recs, stats = call(d, first_frame_no=1, new_file=fl.get("new_file", False), end_file=fl.get("end_file", False))
    torch.cuda.synchronize()
    assert recs.cpu().numpy().tobytes() == z["recs"].tobytes() and stats.cpu().numpy().tobytes() == z["stats"].tobytes()


@pytest.mark.gpu
def test_gpu_dropped_frames_in_a_long_tape():
    """2 000 full-size NTSC frames with a dropped frame here and there and a run of dropped frames: the records and frame descriptors
    equal the oracle's on the frames around every gap, and the chain verifies (what the frames behind a gap start from)."""
    import torch
    from sdvpcmdecoder_amd import Engine, LINE_DTYPE
    n = 400
    luma, _, _ = synth.stc007_frames(n, seed=77, noise_sigma=4.0)
    mask = np.zeros(n, dtype=np.uint8)
    mask[[17, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 250, 399]] = 1
    want, want_stats = oracle_binarize(_with_mask(luma, mask), mode=2)
    libs.load_oracle()
    eng = Engine(0)
    eng.setBinarizationMode(2)
    eng.set_frame_flags(mask)
    recs, stats = eng.binarize_frames(torch.from_numpy(luma).cuda(), first_frame_no=1, new_file=True)
    torch.cuda.synchronize()
    assert recs.cpu().numpy().tobytes() == want.tobytes() and stats.cpu().numpy().tobytes() == want_stats.tobytes()


def _with_mask(luma, mask):
    lib = libs.load_oracle()
    lib.orc_set_empty_frames.argtypes = [C.c_void_p, C.c_size_t]
    lib.orc_set_empty_frames(mask.ctypes.data, len(mask))
    return luma


# ---- the 2x width doubler ---------------------------------------------------------------------------------------------
def test_emu_double_width_feeds_the_doubled_path(emu_lib, oracle_lib):
    """sdv_double_width: integer pixel replication (NOT libswscale's Gauss filter, which the reference uses and which is outside the
    rebuilt path); what it makes decodes, with SDV_FLAG_DOUBLED, to what the oracle makes of the same doubled pixels - and to the
    generator's words."""
    lib = ea.bind(emu_lib)
    lib.sdv_double_width.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_needs_double_width.argtypes = [C.c_int]
    assert lib.sdv_needs_double_width(720) == 1 and lib.sdv_needs_double_width(1440) == 0 and lib.sdv_needs_double_width(8) == 0
    luma, _, _ = synth.stc007_frames(n_frames=3, seed=51, width=717, height=48, noise_sigma=3.0)
    n, h, w = luma.shape
    eng = C.c_void_p(lib.sdv_engine_create(0))
    dst = np.zeros((n, h, 2 * w + 6), dtype=np.uint8)            # padded rows
    assert lib.sdv_double_width(eng, luma.ctypes.data, w, w, n * h, dst.ctypes.data, 2 * w + 6, None) == 0
    assert (dst[:, :, :2 * w] == np.repeat(luma, 2, axis=2)).all() and (dst[:, :, 2 * w:] == 0).all()
    doubled = np.ascontiguousarray(dst[:, :, :2 * w])
    lib.sdv_set_mode(eng, 2)
    rc, recs, stats = ea.emu_binarize(lib, eng, doubled, first_frame_no=1, flags=1 | 2)
    lib.sdv_engine_destroy(eng)
    want, want_stats = oracle_binarize(doubled, mode=2, doubled=True)
    assert rc == 0 and recs.tobytes() == want.tobytes() and stats.view(np.uint8).tobytes() == want_stats.tobytes()
    assert ((recs["flags"] & 64) != 0).sum() > 3 * 40


@pytest.mark.gpu
def test_gpu_double_width():
    import torch
    from sdvpcmdecoder_amd import Engine
    luma, _, _ = synth.stc007_frames(n_frames=5, seed=52, width=720, height=486, noise_sigma=3.0)
    eng = Engine(0)
    assert eng.needs_double_width(720) and not eng.needs_double_width(1440)
    d = eng.double_width(torch.from_numpy(luma).cuda())
    torch.cuda.synchronize()
    assert (d.cpu().numpy() == np.repeat(luma, 2, axis=2)).all()
    eng.setBinarizationMode(2)
    recs, stats = eng.binarize_frames(d, first_frame_no=1, new_file=True, doubled=True)
    want, want_stats = oracle_binarize(np.repeat(luma, 2, axis=2), mode=2, doubled=True)
    assert recs.cpu().numpy().tobytes() == want.tobytes() and stats.cpu().numpy().tobytes() == want_stats.tobytes()
