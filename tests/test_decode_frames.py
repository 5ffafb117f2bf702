"""sdv_decode_frames (the workers back to back, chain selected by the PCM type): the fused call gives what the separate
entry points give - line records and the raw pair stream never leave the engine - for the three formats, with and without the audio stage;
on the SIMT emulator (CPU) and through the C-ABI on the GPU (-m gpu).  The separate entry points are themselves pinned against the oracle
and the real reference by their own test files."""
import ctypes as C

import numpy as np
import pytest

import audio_api as A
import dist_worker
import engine_api as ea
import pcm1_api as p1
import pcm16_api as p16
import stitch_api as sa
from emu_engine_adapter import EmuEngine
from stitch_api import PAIR_DTYPE

PCM1, PCM16X0, STC007 = 0, 1, 2
FRASM = {STC007: sa.FRASM_DTYPE, PCM1: p1.FRASM1_DTYPE, PCM16X0: p16.FRASM16_DTYPE}


def _bind(lib):
    lib.sdv_decode_frames.restype = C.c_int
    lib.sdv_decode_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint,
                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                      C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_void_p]
    lib.sdv_set_pcm_type.argtypes = [C.c_void_p, C.c_int, C.c_int]
    return lib


def _tape(fmt, n):
    from sdvpcmdecoder_amd import synth
    if fmt == STC007:
        luma = synth.stc007_frames(n, seed=12, noise_sigma=4.0)[0].copy()
        luma[:, 77::61] = 16        # lost lines: the error correction and, where it gives up, the dropout masking have something to do
        return luma
    return dist_worker.pcm_tape("pcm1" if fmt == PCM1 else "pcm16x0", n)


def _separate(eng, fmt, luma):
    """The chain stage by stage through the separate entry points -> (pairs, frame descriptors, frame stats)."""
    if fmt == STC007:
        recs, stats = eng.binarize_frames(luma, first_frame_no=1, new_file=True, end_file=True)
        pairs, frames = eng.stitch_frames(recs)
    elif fmt == PCM1:
        recs, stats = eng.pcm1_binarize_frames(luma, first_frame_no=1, new_file=True, end_file=True)
        pairs, frames = eng.pcm1_stitch_frames(eng.pcm1_bin_to_line_recs(recs))
    else:
        recs, stats = eng.pcm16x0_binarize_frames(luma, first_frame_no=1, new_file=True, end_file=True)
        pairs, frames = eng.pcm16x0_stitch_frames(recs)
    return pairs, frames, stats


def _fused_host(lib, h, fmt, luma, with_audio, stop=1, give_stats=True, first_frame_no=1, flags=1 | 4):
    luma = np.ascontiguousarray(luma)
    n, hgt, w = luma.shape
    cap = (n + 2) * 1800 + 8192
    pairs = np.zeros(cap, dtype=PAIR_DTYPE)
    frames = np.zeros(n + 16, dtype=FRASM[fmt])
    stats = np.zeros(n + 1, dtype=ea.STATS_DTYPE)
    pur = np.zeros(8, dtype=A.PURGE_DTYPE)
    npairs, nfr, npur, nm = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
    rc = lib.sdv_decode_frames(h, fmt, luma.ctypes.data, w, w * hgt, w, hgt, n, first_frame_no, flags, pairs.ctypes.data, cap, C.byref(npairs), frames.ctypes.data, len(frames),
                               C.byref(nfr), stats.ctypes.data if give_stats else None, len(stats) if give_stats else 0, 1 if with_audio else 0, stop,
                               pur.ctypes.data, len(pur), C.byref(npur), C.byref(nm), None)
    assert rc == 0, lib.sdv_last_error(h)
    return pairs[:npairs.value], frames[:nfr.value], stats, pur[:npur.value], nm.value


@pytest.mark.parametrize("fmt", [STC007, PCM1, PCM16X0])
def test_emu_fused_equals_separate_calls(fmt, emu_lib, oracle_lib):
    lib = A.bind_product(_bind(ea.bind(emu_lib)))
    luma = _tape(fmt, 3)
    a = EmuEngine(lib)
    lib.sdv_set_pcm_type(a.h, fmt, 0)
    want_p, want_f, want_s = _separate(a, fmt, luma)
    assert len(want_p) > 3 * 1400 and want_p["service_type"][0] == 1 and want_p["service_type"][-1] == 2
    a.close()
    b = EmuEngine(lib)
    lib.sdv_set_pcm_type(b.h, fmt, 0)
    got_p, got_f, got_s, _, _ = _fused_host(lib, b.h, fmt, luma, with_audio=False)
    assert got_p.tobytes() == want_p.tobytes() and got_f.tobytes() == want_f.tobytes() and got_s.tobytes() == want_s.tobytes()
    b.close()
    # ... and with the audio stage behind it: the oracle's AudioProcessor on the separate calls' pair stream
    c = EmuEngine(lib)
    lib.sdv_set_pcm_type(c.h, fmt, 0)
    lib.sdv_set_audio_masking(c.h, A.DROP_INTER_LIN_WORD)
    got_a, got_f2, _, pur, masked = _fused_host(lib, c.h, fmt, luma, with_audio=True, give_stats=False)
    w_out, _, w_pur, w_masked, hit = A.run_cpu(oracle_lib, "orc_", want_p, A.DROP_INTER_LIN_WORD, np.array([len(want_p)], dtype=np.uint64), 1)
    assert hit == 0 and got_a.tobytes() == w_out.tobytes() and pur.tobytes() == w_pur.tobytes() and masked == w_masked and got_f2.tobytes() == want_f.tobytes()
    if fmt == STC007:
        assert masked > 0       # the lost lines left samples the error correction could not restore
    c.close()


@pytest.mark.parametrize("clean", [False, True])
def test_emu_fused_stream_in_calls_equals_one_call(clean, emu_lib, oracle_lib):
    """A source decoded a few frames per call (NEW_FILE with the first, END_FILE with the last): the calls in between hand records of known
    layout to the stitch stage, which then neither looks for the frame ends nor waits for its analysis (sdv_stitch_info.pipelined)."""
    lib = A.bind_product(_bind(ea.bind(emu_lib)))
    luma = _tape(STC007, 12)
    if clean:       # every call settles in one round: the next call's state and the waiting frame are copied ahead of the read-back
        from sdvpcmdecoder_amd import synth
        luma = synth.stc007_frames(12, seed=12, noise_sigma=4.0)[0].copy()
    a = EmuEngine(lib)
    lib.sdv_set_pcm_type(a.h, STC007, 0)
    want_p, want_f, want_s, _, _ = _fused_host(lib, a.h, STC007, luma, with_audio=False)
    a.close()
    b = EmuEngine(lib)
    lib.sdv_set_pcm_type(b.h, STC007, 0)
    got_p, got_f, got_s, piped, direct = [], [], [], [], []
    info = ea.StitchInfo()
    for k in range(0, 12, 2):
        flags = (1 if k == 0 else 0) | (4 if k == 10 else 0)
        p, f, st, _, _ = _fused_host(lib, b.h, STC007, luma[k:k + 2], with_audio=False, first_frame_no=1 + k, flags=flags)
        got_p.append(p.copy()); got_f.append(f.copy()); got_s.append(st[:2 + (1 if k == 10 else 0)].copy())
        assert lib.sdv_get_stitch_info(b.h, C.byref(info)) == 0
        piped.append(int(info.pipelined)); direct.append(int(info.direct_frames))
    b.close()
    assert np.concatenate(got_p).tobytes() == want_p.tobytes() and np.concatenate(got_f).tobytes() == want_f.tobytes()
    assert np.concatenate(got_s).tobytes() == want_s[:13].tobytes()
    assert piped[0] == 0 and piped[-1] == 0 and sum(1 for x in piped if x) >= 2, piped
    # frames the whole-frame capture takes go into the stitch stage's field buffers without line records (the calls of a stream that plays)
    assert direct[0] == 0 and direct[-1] == 0 and all(d <= 2 for d in direct), direct
    if clean:
        assert sum(direct) >= 4, (direct, piped)
        assert any(x & 4 for x in piped), piped            # ... and the stitch kernels of a call behind one that settled at once are queued ahead of the host's look at the frame kernel's round


def test_emu_fused_direct_frames_decoded_again_with_records(emu_lib, oracle_lib, monkeypatch):
    """The way back of the frames that went straight into the field buffers: should the stitch call not take the path they were written for, the
    binarize stage runs again from the stream state it started with and leaves records (forced here by a switch of the developer builds)."""
    from sdvpcmdecoder_amd import synth
    lib = A.bind_product(_bind(ea.bind(emu_lib)))
    luma = synth.stc007_frames(12, seed=12, noise_sigma=4.0)[0].copy()
    a = EmuEngine(lib)
    lib.sdv_set_pcm_type(a.h, STC007, 0)
    want_p, want_f, want_s, _, _ = _fused_host(lib, a.h, STC007, luma, with_audio=False)
    a.close()
    for switch in ("SDV_DIRECT_FORCE_RETRY", "SDV_NO_DIRECT_FIELDS"):
        monkeypatch.setenv(switch, "1")
        b = EmuEngine(lib)
        lib.sdv_set_pcm_type(b.h, STC007, 0)
        got_p, got_f, got_s = [], [], []
        info = ea.StitchInfo()
        for k in range(0, 12, 3):
            p, f, st, _, _ = _fused_host(lib, b.h, STC007, luma[k:k + 3], with_audio=False, first_frame_no=1 + k, flags=(1 if k == 0 else 0) | (4 if k == 9 else 0))
            got_p.append(p.copy()); got_f.append(f.copy()); got_s.append(st[:3 + (1 if k == 9 else 0)].copy())
            assert lib.sdv_get_stitch_info(b.h, C.byref(info)) == 0 and info.direct_frames == 0
        b.close()
        monkeypatch.delenv(switch)
        assert np.concatenate(got_p).tobytes() == want_p.tobytes() and np.concatenate(got_f).tobytes() == want_f.tobytes()
        assert np.concatenate(got_s).tobytes() == want_s[:13].tobytes()


def test_emu_fused_refuses_bad_arguments(emu_lib):
    lib = A.bind_product(_bind(ea.bind(emu_lib)))
    e = EmuEngine(lib)
    luma = np.zeros((1, 16, 720), dtype=np.uint8)
    n1, n2 = C.c_size_t(0), C.c_size_t(0)
    buf = np.zeros(4096, dtype=PAIR_DTYPE)
    args = (luma.ctypes.data, 720, 720 * 16, 720, 16, 1, 1, 0, buf.ctypes.data, len(buf), C.byref(n1), buf.ctypes.data, 8, C.byref(n2), None, 0)
    assert lib.sdv_decode_frames(e.h, 5, *args, 0, 0, None, 0, None, None, None) == -1 and b"unknown PCM type" in lib.sdv_last_error(e.h)
    assert lib.sdv_decode_frames(e.h, STC007, *args, 1, 0, None, 0, None, None, None) == -1 and b"with_audio" in lib.sdv_last_error(e.h)
    assert lib.sdv_decode_frames(e.h, STC007, None, *args[1:], 0, 0, None, 0, None, None, None) == 1      # LB_RET_NULL_VIDEO
    e.close()


# ---- the product on the GPU ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [STC007, PCM1, PCM16X0])
def test_gpu_fused_equals_separate_calls(fmt, oracle_lib):
    import torch
    from sdvpcmdecoder_amd import Engine, Pcm16x0StitchSettings
    luma = _tape(fmt, 6)
    d = torch.from_numpy(np.ascontiguousarray(luma)).cuda()
    eng = Engine(0)
    eng.setPCMType(fmt)
    if fmt == STC007:
        lines, stats = eng.binarize_frames(d, first_frame_no=1, new_file=True, end_file=True)
        p, f = eng.stitch_frames(lines)
    elif fmt == PCM1:
        lines, stats = eng.pcm1_binarize_frames(d, first_frame_no=1, new_file=True, end_file=True)
        p, f = eng.pcm1_stitch_frames(eng.pcm1_bin_to_line_recs(lines))
    else:
        lines, stats = eng.pcm16x0_binarize_frames(d, first_frame_no=1, new_file=True, end_file=True)
        p, f = eng.pcm16x0_stitch_frames(lines)
    want_p, want_f, want_s = p.cpu().numpy().copy(), f.cpu().numpy().copy(), stats.cpu().numpy().copy()
    eng2 = Engine(0)
    eng2.setPCMType(fmt)
    gp, gf, gs = eng2.decode_frames(fmt, d, first_frame_no=1, new_file=True, end_file=True)
    assert gp.cpu().numpy().tobytes() == want_p.tobytes() and gf.cpu().numpy().tobytes() == want_f.tobytes() and gs.cpu().numpy().tobytes() == want_s.tobytes()
    eng3 = Engine(0)
    eng3.setPCMType(fmt)
    eng3.set_audio_masking(A.DROP_INTER_LIN_WORD)
    ga, gf3, _, pur, masked = eng3.decode_frames(fmt, d, first_frame_no=1, new_file=True, end_file=True, with_audio=True, audio_stop=True)
    wp = want_p.view(PAIR_DTYPE).reshape(-1)
    w_out, _, w_pur, w_masked, hit = A.run_cpu(oracle_lib, "orc_", wp, A.DROP_INTER_LIN_WORD, np.array([len(wp)], dtype=np.uint64), 1)
    assert hit == 0 and ga.cpu().numpy().tobytes() == w_out.tobytes() and pur.cpu().numpy().tobytes() == w_pur.tobytes() and masked == w_masked
    assert gf3.cpu().numpy().tobytes() == want_f.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("damaged", [False, True])
def test_gpu_fused_stream_in_calls_equals_one_call(damaged):
    """1 200 frames through sdv_decode_frames in eight calls (NEW_FILE with the first, END_FILE with the last) against one call over all of them:
    the calls in between take the pipelined way through the stitch stage (records of known layout, no wait for the analysis, the next call's
    state copied ahead of the read-back) - with lost lines in some frames the assumptions fail now and then and the classic way takes over."""
    import torch
    from sdvpcmdecoder_amd import Engine, synth
    n, per = 1200, 150
    luma, _ = synth.stc007_frames_torch(n, seed=31, device="cuda", noise_sigma=3.0)
    if damaged:
        rng = np.random.default_rng(5)
        for f in rng.choice(np.arange(10, n - 10), size=40, replace=False):
            luma[int(f), int(rng.integers(30, 450))] = 16
    one = Engine(0); one.setPCMType(STC007)
    wp, wf, ws = one.decode_frames(STC007, luma, first_frame_no=1, new_file=True, end_file=True)
    want_p, want_f, want_s = wp.cpu().numpy().copy(), wf.cpu().numpy().copy(), ws.cpu().numpy().copy()
    eng = Engine(0); eng.setPCMType(STC007)
    got_p, got_f, got_s, piped = [], [], [], []
    for k in range(0, n, per):
        p, f, st = eng.decode_frames(STC007, luma[k:k + per], first_frame_no=1 + k, new_file=k == 0, end_file=k + per == n)
        got_p.append(p.cpu().numpy().copy()); got_f.append(f.cpu().numpy().copy()); got_s.append(st.cpu().numpy().copy())
        piped.append(int(eng.stitch_info().pipelined))
    assert np.concatenate(got_p).tobytes() == want_p.tobytes()
    assert np.concatenate(got_f).tobytes() == want_f.tobytes()
    assert np.concatenate(got_s)[:n].tobytes() == want_s[:n].tobytes()
    assert piped[0] == 0 and piped[-1] == 0, piped
    if not damaged:
        assert sum(1 for x in piped if x) >= 4, piped


# ---- the fused entry against the reference's own output and the sequential oracle (not against the separate calls) ------------------------
def _load_gen(name):
    import importlib.util
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location(name, os.path.join(gold, name + ".py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg, gold


def _damaged_tape(n, seed=91):
    """Full-size NTSC frames with the kinds of damage the stitch stage and the speculation have to get through together: lost lines, lines no
    reference level reads (their sweeps are settled by the sweep kernels between the passes), and a data window that moves half way."""
    from sdvpcmdecoder_amd import synth
    from test_gpu_parity import _unreadable_cells
    luma = synth.stc007_frames(n, seed=seed, noise_sigma=4.0)[0].copy()
    luma[:, 77::61] = 16
    luma = _unreadable_cells(luma, every=211, seed=seed)
    luma[n // 2:] = np.roll(luma[n // 2:], 5, axis=2)
    return np.ascontiguousarray(luma)


def _oracle_chain(oracle_lib, luma):
    from oracle_run import oracle_binarize
    recs, stats = oracle_binarize(luma, mode=2, first_frame_no=1, new_file=True, end_file=True)
    pairs, frames = sa.run_cpu(oracle_lib, "orc_", recs, sa.default_settings())
    return pairs, frames, stats


def test_emu_fused_uneven_calls_equal_the_sequential_oracle(emu_lib, oracle_lib):
    """A damaged tape through sdv_decode_frames in calls of 1, 3, 2 and 1 frames (emulator): pair stream, frame descriptors and frame statistics equal
    what the oracle's two workers make of the whole tape one line after the other."""
    lib = A.bind_product(_bind(ea.bind(emu_lib)))
    from sdvpcmdecoder_amd import synth
    from test_gpu_parity import _unreadable_cells
    luma = synth.stc007_frames(7, seed=93, noise_sigma=4.0, height=120, lines_per_field=60)[0].copy()
    luma[:, 33::29] = 16
    luma = np.ascontiguousarray(_unreadable_cells(luma, every=47, seed=5))
    want_p, want_f, want_s = _oracle_chain(oracle_lib, luma)
    e = EmuEngine(lib)
    lib.sdv_set_pcm_type(e.h, STC007, 0)
    got_p, got_f, got_s = [], [], []
    k = 0
    for cnt in (1, 3, 2, 1):
        p, f, st, _, _ = _fused_host(lib, e.h, STC007, luma[k:k + cnt], with_audio=False, first_frame_no=1 + k, flags=(1 if k == 0 else 0) | (4 if k + cnt == 7 else 0))
        got_p.append(p.copy()); got_f.append(f.copy()); got_s.append(st[:cnt + (1 if k + cnt == 7 else 0)].copy())
        k += cnt
    e.close()
    assert np.concatenate(got_p).tobytes() == want_p.tobytes() and np.concatenate(got_f).tobytes() == want_f.tobytes()
    assert np.concatenate(got_s).view(np.uint8).tobytes() == want_s.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("calls", [(4,), (1, 2, 1), (3, 1)])
def test_gpu_fused_entry_matches_reference_golden_ntsc(calls):
    """sdv_decode_frames over the file of tests/golden/e2e_ntsc_file.npz, in one call and in uneven ones: PCMSamplePair stream, FrameAsmSTC007 rows and
    frame statistics equal what the REAL reference's VideoToDigital and STC007DataStitcher made of that file."""
    import torch
    from sdvpcmdecoder_amd import Engine
    mg, gold = _load_gen("make_golden_stitch")
    import os
    z = np.load(os.path.join(gold, "e2e_ntsc_file.npz"))
    luma = mg.make_e2e_luma()
    d = torch.from_numpy(luma).cuda()
    eng = Engine(0); eng.setPCMType(STC007)
    got_p, got_f, got_s, k = [], [], [], 0
    for cnt in calls:
        p, f, st = eng.decode_frames(STC007, d[k:k + cnt], first_frame_no=1 + k, new_file=k == 0, end_file=k + cnt == len(luma))
        got_p.append(p.cpu().numpy().copy()); got_f.append(f.cpu().numpy().copy()); got_s.append(st.cpu().numpy().copy())
        k += cnt
    assert np.concatenate(got_p).tobytes() == np.ascontiguousarray(z["pairs"]).tobytes()
    assert np.concatenate(got_f).tobytes() == np.ascontiguousarray(z["frames"]).tobytes()
    assert np.concatenate(got_s).tobytes() == np.ascontiguousarray(z["stats"]).tobytes()


@pytest.mark.gpu
def test_gpu_fused_entry_with_audio_matches_reference_golden():
    """... and with the AudioProcessor behind it: the masked stream the real AudioProcessor made of the real workers' output for that file
    (tests/golden/e2e_ntsc_file_audio.npz)."""
    import os
    import torch
    from sdvpcmdecoder_amd import Engine
    mg, gold = _load_gen("make_golden_stitch")
    z = np.load(os.path.join(gold, "e2e_ntsc_file_audio.npz"))
    d = torch.from_numpy(mg.make_e2e_luma()).cuda()
    eng = Engine(0); eng.setPCMType(STC007)
    eng.set_audio_masking(A.DROP_INTER_LIN_WORD)
    out, _f, _s, pur, masked = eng.decode_frames(STC007, d, first_frame_no=1, new_file=True, end_file=True, with_audio=True, audio_stop=True)
    assert out.cpu().numpy().tobytes() == np.ascontiguousarray(z["pairs"]).tobytes()
    assert masked == int(z["masked"])
    assert eng.wav_files(out, pur)[0] == np.ascontiguousarray(z["wav0"]).tobytes()      # the file the real SamplesToWAV wrote


@pytest.mark.gpu
@pytest.mark.parametrize("ei", [False, True])
def test_gpu_fused_entry_matches_reference_golden_pcm16x0(ei):
    """sdv_decode_frames(SDV_PCM_PCM16X0) over the file of tests/golden/e2e_pcm16x0_<si|ei>.npz, in one call and in calls of 2 + 3 frames: what the REAL
    reference's VideoToDigital (TYPE_PCM16X0) and PCM16X0DataStitcher made of it."""
    import os
    import torch
    from sdvpcmdecoder_amd import Engine, Pcm16x0StitchSettings
    mg, gold = _load_gen("make_golden_pcm16")
    z = np.load(os.path.join(gold, "e2e_pcm16x0_%s.npz" % ("ei" if ei else "si")))
    luma, _audio = mg.make_e2e_luma(ei)
    d = torch.from_numpy(np.ascontiguousarray(luma)).cuda()
    for calls in ((len(luma),), (2, len(luma) - 2)):
        eng = Engine(0); eng.setPCMType(PCM16X0)
        eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(p16.default_settings(format=p16.FORMAT_EI if ei else p16.FORMAT_SI))))
        got_p, got_f, k = [], [], 0
        for cnt in calls:
            p, f, _st = eng.decode_frames(PCM16X0, d[k:k + cnt], first_frame_no=1 + k, new_file=k == 0, end_file=k + cnt == len(luma))
            got_p.append(p.cpu().numpy().copy()); got_f.append(f.cpu().numpy().copy())
            k += cnt
        assert np.concatenate(got_p).tobytes() == np.ascontiguousarray(z["pairs"]).tobytes(), calls
        assert np.concatenate(got_f).tobytes() == np.ascontiguousarray(z["frames"]).tobytes(), calls


@pytest.mark.gpu
def test_gpu_fused_uneven_calls_equal_the_sequential_oracle(oracle_lib):
    """A damaged 40-frame tape (lost lines, unreadable lines, a window that moves) through sdv_decode_frames in uneven calls, so that the pipelined
    way through the stitch stage, the classic one and the sweeps between the passes of the frame kernel all take turns: compared with the
    sequential truth - the oracle's two workers over the whole tape - not with the engine's other entry points."""
    import torch
    from sdvpcmdecoder_amd import Engine
    n = 40
    luma = _damaged_tape(n)
    want_p, want_f, want_s = _oracle_chain(oracle_lib, luma)
    d = torch.from_numpy(luma).cuda()
    eng = Engine(0); eng.setPCMType(STC007)
    got_p, got_f, got_s, k = [], [], [], 0
    for cnt in (1, 7, 2, 13, 5, 11, 1):
        p, f, st = eng.decode_frames(STC007, d[k:k + cnt], first_frame_no=1 + k, new_file=k == 0, end_file=k + cnt == n)
        got_p.append(p.cpu().numpy().copy()); got_f.append(f.cpu().numpy().copy()); got_s.append(st.cpu().numpy().copy())
        k += cnt
    assert k == n
    assert np.concatenate(got_p).tobytes() == want_p.tobytes()
    assert np.concatenate(got_f).tobytes() == want_f.tobytes()
    assert np.concatenate(got_s).tobytes() == want_s.tobytes()


def _geometry_tape(geometry, n):
    from sdvpcmdecoder_amd import synth
    if geometry == "odd_height":
        return np.ascontiguousarray(synth.stc007_frames(n, seed=14, noise_sigma=4.0)[0][:, :485, :])
    if geometry == "pal":
        return synth.stc007_frames(n, seed=15, height=576, lines_per_field=294, noise_sigma=4.0)[0].copy()
    if geometry == "field_buffer_full":         # 588 rows: both fields exactly the 294 lines a field buffer of the stitcher holds (stc007datastitcher.h: BUF_SIZE_FIELD)
        return synth.stc007_frames(n, seed=16, height=588, lines_per_field=294, noise_sigma=4.0)[0].copy()
    assert geometry == "field_too_long"         # 590 rows: 295 lines per field, one more than a field buffer holds - such a frame must go through its records
    return synth.stc007_frames(n, seed=17, height=590, lines_per_field=295, noise_sigma=4.0)[0].copy()


@pytest.mark.parametrize("geometry", ["odd_height", "pal", "field_buffer_full", "field_too_long"])
def test_emu_fused_direct_frames_other_geometries(geometry, emu_lib, oracle_lib):
    """Frames without records (the frame kernel writes the stitch stage's field buffers itself) with fields of unequal length - a frame of 485 rows: 243 + 242
    lines - with the 288 lines per field of a PAL frame (five chunks of 64 lines), with fields that fill a field buffer to its last line and with fields one line
    longer than that (no frame of those may go without records): calls of two frames against one call over the whole tape."""
    lib = A.bind_product(_bind(ea.bind(emu_lib)))
    luma = _geometry_tape(geometry, 10)
    a = EmuEngine(lib)
    lib.sdv_set_pcm_type(a.h, STC007, 0)
    want_p, want_f, want_s, _, _ = _fused_host(lib, a.h, STC007, luma, with_audio=False)
    a.close()
    b = EmuEngine(lib)
    lib.sdv_set_pcm_type(b.h, STC007, 0)
    got_p, got_f, direct = [], [], []
    info = ea.StitchInfo()
    for k in range(0, 10, 2):
        p, f, st, _, _ = _fused_host(lib, b.h, STC007, luma[k:k + 2], with_audio=False, first_frame_no=1 + k, flags=(1 if k == 0 else 0) | (4 if k == 8 else 0))
        got_p.append(p.copy()); got_f.append(f.copy())
        assert lib.sdv_get_stitch_info(b.h, C.byref(info)) == 0
        direct.append(int(info.direct_frames))
    b.close()
    assert np.concatenate(got_p).tobytes() == want_p.tobytes() and np.concatenate(got_f).tobytes() == want_f.tobytes()
    if geometry == "field_too_long": assert sum(direct) == 0, direct
    else: assert sum(direct) >= 2, direct


@pytest.mark.gpu
@pytest.mark.parametrize("geometry", ["odd_height", "pal", "field_buffer_full", "field_too_long"])
def test_gpu_fused_direct_frames_other_geometries(geometry, oracle_lib):
    """The same on the GPU, against the sequential oracle."""
    import torch
    from sdvpcmdecoder_amd import Engine
    luma = _geometry_tape(geometry, 12)
    want_p, want_f, want_s = _oracle_chain(oracle_lib, luma)
    d = torch.from_numpy(luma).cuda()
    eng = Engine(0); eng.setPCMType(STC007)
    got_p, got_f, got_s, direct = [], [], [], []
    for k in range(0, 12, 3):
        p, f, st = eng.decode_frames(STC007, d[k:k + 3], first_frame_no=1 + k, new_file=k == 0, end_file=k + 3 == 12)
        got_p.append(p.cpu().numpy().copy()); got_f.append(f.cpu().numpy().copy()); got_s.append(st.cpu().numpy().copy())
        direct.append(int(eng.stitch_info().direct_frames))
    assert np.concatenate(got_p).tobytes() == want_p.tobytes()
    assert np.concatenate(got_f).tobytes() == want_f.tobytes()
    assert np.concatenate(got_s).tobytes() == want_s.tobytes()
    if geometry == "field_too_long": assert sum(direct) == 0, direct
    else: assert sum(direct) >= 2, direct


_RETRY_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from sdvpcmdecoder_amd import Engine, synth
STC007 = 2
luma = synth.stc007_frames(12, seed=12, noise_sigma=4.0)[0].copy()
d = torch.from_numpy(luma).cuda()
def run(switch):
    eng = Engine(0); eng.setPCMType(STC007)
    out = []; direct = []; piped = []
    for k in range(0, 12, 3):
        if switch and k >= 6: os.environ[switch] = "1"          # the first calls arm the direct fields and queue the stitch stage ahead; then the switch
        p, f, st = eng.decode_frames(STC007, d[k:k + 3], first_frame_no=1 + k, new_file=k == 0, end_file=k + 3 == 12)
        out.append((p.cpu().numpy().tobytes(), f.cpu().numpy().tobytes(), st.cpu().numpy().tobytes()))
        i = eng.stitch_info(); direct.append(int(i.direct_frames)); piped.append(int(i.pipelined))
    if switch: del os.environ[switch]
    return out, direct, piped
want, d0, p0 = run(None)
assert sum(d0) >= 2, d0
for switch in ("SDV_DIRECT_FORCE_RETRY", "SDV_NO_DIRECT_FIELDS", "SDV_NO_QUEUE_AHEAD", "SDV_NO_PREDICT_IN_KERNEL"):
    got, d1, p1 = run(switch)
    assert got == want, switch
    if switch in ("SDV_DIRECT_FORCE_RETRY", "SDV_NO_DIRECT_FIELDS"): assert d1[2] == 0 and d1[3] == 0, (switch, d1)
    if switch == "SDV_NO_QUEUE_AHEAD": assert not any(x & 4 for x in p1[2:]), p1
print("RETRY_OK", d0, p0)
"""


@pytest.mark.gpu
def test_gpu_fused_way_back_with_records_on_a_developer_build():
    """The fused entry's defensive ways on real hardware (stream order, the asynchronous read-back and its event are trivial under the emulator): the frames
    that went straight into the field buffers decoded again with records from the restored stream state (SDV_STITCH_RETRY_WITH_RECORDS), a call without direct
    fields, without the stitch stage queued ahead, without the in-kernel prediction.  The product library cannot be steered there from outside (its two checks agree
    by construction, it does not read the environment): this runs the developer build of the same sources (build.py: libsdvpcm_hip_dev.so, -DSDV_DEV_AIDS) in a
    process of its own and compares every way with the default one, call by call."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = os.path.join(root, "sdvpcmdecoder_amd", "libsdvpcm_hip_dev.so")
    if not os.path.exists(dev): pytest.skip("no developer build of the HIP library (sdvpcmdecoder_amd/build.py: build_hip_dev)")
    env = dict(os.environ); env["SDVPCM_LIB"] = dev
    r = subprocess.run([sys.executable, "-c", _RETRY_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RETRY_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _tape_clean_then_damaged(n=15, bad_from=9):
    """Frames that play, then frames with lost lines: the call that meets them has its frame stage run more than one round."""
    from sdvpcmdecoder_amd import synth
    luma = synth.stc007_frames(n, seed=12, noise_sigma=4.0)[0].copy()
    luma[bad_from:, 77::61] = 16
    return luma


def test_emu_fused_stitch_queued_ahead_is_made_over_when_the_frame_stage_needs_more_rounds(emu_lib, oracle_lib):
    """Calls of three frames: while the tape plays the stitch kernels of a call are queued behind the frame kernel's first round (pipelined & 4); the call
    that meets the damage had them queued too - its frame stage then took more rounds and the stitch stage was run again on the final records (& 8)."""
    lib = A.bind_product(_bind(ea.bind(emu_lib)))
    luma = _tape_clean_then_damaged()
    a = EmuEngine(lib)
    lib.sdv_set_pcm_type(a.h, STC007, 0)
    want_p, want_f, want_s, _, _ = _fused_host(lib, a.h, STC007, luma, with_audio=False)
    a.close()
    b = EmuEngine(lib)
    lib.sdv_set_pcm_type(b.h, STC007, 0)
    got_p, got_f, got_s, piped = [], [], [], []
    info = ea.StitchInfo()
    for k in range(0, 15, 3):
        p, f, st, _, _ = _fused_host(lib, b.h, STC007, luma[k:k + 3], with_audio=False, first_frame_no=1 + k, flags=(1 if k == 0 else 0) | (4 if k == 12 else 0))
        got_p.append(p.copy()); got_f.append(f.copy()); got_s.append(st[:3 + (1 if k == 12 else 0)].copy())
        assert lib.sdv_get_stitch_info(b.h, C.byref(info)) == 0
        piped.append(int(info.pipelined))
    b.close()
    assert np.concatenate(got_p).tobytes() == want_p.tobytes() and np.concatenate(got_f).tobytes() == want_f.tobytes()
    assert np.concatenate(got_s).tobytes() == want_s[:16].tobytes()
    assert any(x & 4 for x in piped) and any(x & 8 for x in piped), piped


@pytest.mark.gpu
def test_gpu_fused_stitch_queued_ahead_is_made_over_when_the_frame_stage_needs_more_rounds(oracle_lib):
    """The same on the GPU (where the queued kernels really run beside the host's look at the flags), against the sequential oracle."""
    import torch
    from sdvpcmdecoder_amd import Engine
    luma = _tape_clean_then_damaged(n=30, bad_from=21)
    want_p, want_f, want_s = _oracle_chain(oracle_lib, luma)
    d = torch.from_numpy(luma).cuda()
    eng = Engine(0); eng.setPCMType(STC007)
    got_p, got_f, got_s, piped = [], [], [], []
    for k in range(0, 30, 3):
        p, f, st = eng.decode_frames(STC007, d[k:k + 3], first_frame_no=1 + k, new_file=k == 0, end_file=k + 3 == 30)
        got_p.append(p.cpu().numpy().copy()); got_f.append(f.cpu().numpy().copy()); got_s.append(st.cpu().numpy().copy())
        piped.append(int(eng.stitch_info().pipelined))
    assert np.concatenate(got_p).tobytes() == want_p.tobytes()
    assert np.concatenate(got_f).tobytes() == want_f.tobytes()
    assert np.concatenate(got_s).tobytes() == want_s.tobytes()
    assert any(x & 4 for x in piped) and any(x & 8 for x in piped), piped


def test_emu_stitch_and_fused_calls_interleaved_with_the_visualiser_feeds_toggled(emu_lib, oracle_lib):
    """One stream fed alternately through sdv_decode_frames (records of known layout: the pipelined way through the stitch stage, the next call's state
    copied ahead of the read-back) and through sdv_binarize_frames + sdv_stitch_frames, with the block and assembled-line outputs of the stitch stage
    switched on and off in between (switched on they make the stage run its turns once more for the feed): the hand-over between the calls must not
    depend on which way a call took.  The whole equals the sequential oracle; the emulator build also checks the state that was copied ahead against
    the last turn's (SDV_DEV_AIDS, stitch_engine.inc)."""
    from sdvpcmdecoder_amd import synth
    lib = A.bind_product(_bind(ea.bind(emu_lib)))
    lib.sdv_set_stitch_block_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_set_stitch_line_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    n = 14
    luma = np.ascontiguousarray(synth.stc007_frames(n, seed=97, noise_sigma=3.0, height=120, lines_per_field=60)[0])
    want_p, want_f, _ = _oracle_chain(oracle_lib, luma)
    e = EmuEngine(lib)
    lib.sdv_set_pcm_type(e.h, STC007, 0)
    blocks = np.zeros(4096, dtype=sa.BLOCK_DTYPE)
    asm = np.zeros(64 * 1024, dtype=np.uint8)
    got_p, got_f = [], []
    piped = 0
    info = ea.StitchInfo()
    plan = [("fused", 2, None), ("fused", 2, None), ("fused", 2, "blocks"), ("split", 2, None), ("fused", 2, "lines"), ("fused", 2, None), ("fused", 2, "off")]
    k = 0
    for how, cnt, feed in plan:
        if feed == "blocks":
            assert lib.sdv_set_stitch_block_output(e.h, blocks.ctypes.data, len(blocks)) == 0
        elif feed == "lines":
            assert lib.sdv_set_stitch_block_output(e.h, None, 0) == 0
            assert lib.sdv_set_stitch_line_output(e.h, asm.ctypes.data, len(asm) // 64) == 0
        elif feed == "off":
            assert lib.sdv_set_stitch_line_output(e.h, None, 0) == 0
        flags = (1 if k == 0 else 0) | (4 if k + cnt == n else 0)
        if how == "fused":
            p, f, _st, _, _ = _fused_host(lib, e.h, STC007, luma[k:k + cnt], with_audio=False, first_frame_no=1 + k, flags=flags)
        else:
            recs, _stats = e.binarize_frames(luma[k:k + cnt], first_frame_no=1 + k, new_file=k == 0, end_file=k + cnt == n)
            p, f = e.stitch_frames(recs)
        got_p.append(p.copy()); got_f.append(f.copy())
        assert lib.sdv_get_stitch_info(e.h, C.byref(info)) == 0
        piped += int(info.pipelined)
        k += cnt
    e.close()
    assert k == n
    assert np.concatenate(got_p).tobytes() == want_p.tobytes() and np.concatenate(got_f).tobytes() == want_f.tobytes()
    assert piped >= 2, piped
