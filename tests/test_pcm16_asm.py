"""The re-assembled window of PCM-16x0: PCM16X0DataStitcher::newLineProcessed(PCM16X0SubLine) (pcm16x0datastitcher.cpp:5196-5213) as
records of the binarizer's own type with an END_FRAME record behind every frame's (sdv_set_pcm16x0_stitch_line_output), drawn by the renderer of the
lines window (RenderPCM::renderNewLine(PCM16X0SubLine), renderpcm.cpp:743-937: sdv_vis_render_lines, SDV_VIS_PCM16X0_LINES), as MainWindow wires it
(mainwindow.cpp:2040-2044).
  oracle (oracle/pcm16.c, oracle/render.c)  vs  the real stitcher's sub-lines and the real RenderPCM on them (live, when oracle/_ref is built) and the
                                                committed fixtures;
  HIP kernels                               vs  the oracle: on the emulator, and through the C-ABI on the GPU (-m gpu)."""
import functools
import hashlib
import os

import numpy as np
import pytest

import engine_api as ea
import libs
import pcm16_api as p16
import render_api as ra

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [n for n in p16.CASES if n not in p16.LIVELOCK and n not in ("si_lost_long", "ei_lost_every_field")]      # (the two long tapes: minutes of reference time)
ASM_GOLDEN = ("si_bad10", "si_picked_forced", "ei_bad10", "si_file_marks", "si_lost_sublines")


@functools.lru_cache(maxsize=None)        # (the tests of a case share one run of the oracle; nobody writes into what it returns)
def _oracle(name):
    recs, st = p16.make_input(name)
    pairs, frames, blocks, lines = p16.run_cpu_feeds(libs.load_oracle(), "orc_", recs, st)
    assert int((lines["service_type"] == p16.SRV_END_FRAME).sum()) == int((frames["service_type"] == 0).sum())
    return recs, st, pairs, frames, np.ascontiguousarray(lines)


@pytest.mark.ref
@pytest.mark.parametrize("name", CASES)
def test_oracle_lines_match_live_reference(name, oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    recs, st, pairs, frames, lines = _oracle(name)
    rp, rf, rb, rl = p16.run_cpu_feeds(libs.load_ref(), "ref_", recs, st)
    assert pairs.tobytes() == rp.tobytes() and frames.tobytes() == rf.tobytes()
    assert lines.tobytes() == rl.tobytes(), [(f, np.argwhere(lines[f] != rl[f])[:4].tolist()) for f in lines.dtype.names if lines[f].tobytes() != rl[f].tobytes()]
    # the window: the real renderer on the real stitcher's sub-lines against the oracle's renderer on the oracle's
    want = ra.run_ref(ra.PCM16X0, np.ascontiguousarray(rl))
    got, _ = ra.run_oracle(ra.PCM16X0, lines)
    mask = ra.written(ra.PCM16X0, lines)
    assert (np.where(mask, got, 0) == np.where(mask, want, 0)).all()


@pytest.mark.parametrize("name", ASM_GOLDEN)
def test_oracle_matches_golden_from_reference(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "pcm16asm_" + name + ".npz"))
    recs, st, pairs, frames, lines = _oracle(name)
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"])
    assert hashlib.sha256(lines.tobytes()).hexdigest() == str(z["lines_sha256"]), "the assembled sub-lines differ from the real stitcher's"
    out, _ = ra.run_oracle(ra.PCM16X0, lines)
    mask = ra.written(ra.PCM16X0, lines)
    assert ra.digest(out, mask) == str(z["canvases_sha256"])
    assert (np.where(mask[-1], out[-1], 0) == z["last_canvas"]).all()


@pytest.fixture(scope="module")
def emu(emu_lib):
    return ea.bind(emu_lib)


@pytest.mark.parametrize("name", CASES)
def test_emu_lines_match_oracle(name, emu, oracle_lib):
    recs, st, want_p, want_f, want_l = _oracle(name)
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames, lines = ea.emu_pcm16_stitch_lines(emu, eng, recs, st)
    assert rc == 0 and pairs.tobytes() == want_p.tobytes() and frames.tobytes() == want_f.tobytes()
    assert lines.tobytes() == want_l.tobytes(), [(f, np.argwhere(lines[f] != want_l[f])[:4].tolist()) for f in lines.dtype.names if lines[f].tobytes() != want_l[f].tobytes()]
    emu.sdv_engine_destroy(eng)


def test_emu_lines_in_calls_and_too_small(emu, oracle_lib):
    recs, st, want_p, want_f, want_l = _oracle("si_file_marks")
    eng = emu.sdv_engine_create(0)
    cuts = [0, len(recs) // 3, len(recs) // 3 + 11, len(recs)]
    got = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f, ln = ea.emu_pcm16_stitch_lines(emu, eng, recs[a:b], st if a == 0 else None)
        assert rc == 0
        got.append(ln)
    assert np.concatenate(got).tobytes() == want_l.tobytes()
    emu.sdv_engine_destroy(eng)
    eng = emu.sdv_engine_create(0)
    rc, p, f, ln = ea.emu_pcm16_stitch_lines(emu, eng, recs, st, line_cap=100)
    assert rc != 0 and b"assembled sub-line records" in emu.sdv_last_error(eng) and ea.emu_pcm16_stitch_lines.last_count == len(want_l)
    rc, p, f, ln = ea.emu_pcm16_stitch_lines(emu, eng, recs, st)                 # the refused call took nothing: once more with room
    assert rc == 0 and ln.tobytes() == want_l.tobytes()
    emu.sdv_engine_destroy(eng)


def test_emu_window_is_the_lines_renderer_on_the_feed(emu, oracle_lib):
    """sdv_vis_render_lines(SDV_VIS_PCM16X0_LINES) on the feed = the oracle's renderer on the oracle's feed."""
    import ctypes as C
    recs, st, want_p, want_f, want_l = _oracle("si_picked_forced")
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames, lines = ea.emu_pcm16_stitch_lines(emu, eng, recs, st)
    assert rc == 0
    w, h = ra.SIZE[ra.PCM16X0]
    n = ra.n_frames(lines)
    out = np.zeros((n, h, w), dtype=np.uint32)
    nf = C.c_size_t(0)
    emu.sdv_vis_render_lines.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    assert emu.sdv_vis_render_lines(eng, ra.PCM16X0, lines.ctypes.data, len(lines), out.ctypes.data, n, C.byref(nf), None) == 0 and nf.value == n
    want, _ = ra.run_oracle(ra.PCM16X0, want_l)
    assert (out == want).all()
    emu.sdv_engine_destroy(eng)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_lines_and_window_match_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine, Pcm16x0StitchSettings
    recs, st, want_p, want_f, want_l = _oracle(name)
    eng = Engine(0)
    eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    ln = torch.zeros((len(want_l) + 64, 36), dtype=torch.uint8, device="cuda")
    eng.set_pcm16x0_stitch_line_output(ln)
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 36)).cuda()
    p, f = eng.pcm16x0_stitch_frames(d)
    nl = eng.pcm16x0_stitch_line_count()
    got = ln[:nl].cpu().numpy().reshape(-1).view(recs.dtype)
    assert p.cpu().numpy().tobytes() == want_p.tobytes() and got.tobytes() == want_l.tobytes()
    canvases = eng.vis_render_lines(ra.PCM16X0, ln[:nl].contiguous(), ra.n_frames(want_l)).cpu().numpy().view(np.uint32)
    want, _ = ra.run_oracle(ra.PCM16X0, want_l)
    assert (canvases.reshape(want.shape) == want).all()
    if name in ASM_GOLDEN:
        z = np.load(os.path.join(GOLD, "pcm16asm_" + name + ".npz"))
        assert hashlib.sha256(got.tobytes()).hexdigest() == str(z["lines_sha256"])
        assert ra.digest(canvases.reshape(want.shape), ra.written(ra.PCM16X0, got)) == str(z["canvases_sha256"])
