"""The canvases of RenderPCM's binarized-lines visualiser (sdv_vis_render_lines).
  -m "not gpu": the oracle (oracle/render.c) against the real RenderPCM (when the reference build is loadable) and against the fixtures
                the real RenderPCM made; the product's kernels in the SIMT emulator against the oracle.
  -m gpu:       the product on the GPU against the oracle and the fixtures."""
import ctypes as C
import os

import numpy as np
import pytest

import libs
import render_api as ra

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _masked(canvases, mask):
    return np.where(mask, canvases, 0)


def _diff(got, want, mask):
    d = np.argwhere(mask & (got != want))
    return f"{len(d)} pixels differ, first (frame, row, x) {d[:4].tolist()}: got {[hex(got[tuple(x)]) for x in d[:4]]} want {[hex(want[tuple(x)]) for x in d[:4]]}"


@pytest.mark.ref
@pytest.mark.parametrize("name", list(ra.CASES))
def test_oracle_matches_live_reference(name, oracle_lib):
    """Every pixel some frame has drawn so far equals the real RenderPCM's (the rest of its canvas is uninitialised memory)."""
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    kind, recs = ra.make_input(name)
    out, _ = ra.run_oracle(kind, recs)
    ref = ra.run_ref(kind, recs)
    mask = ra.written(kind, recs)
    assert (_masked(out, mask) == _masked(ref, mask)).all(), _diff(out, ref, mask)
    assert (out[~mask] == ra.BLANK).all()


@pytest.mark.parametrize("name", ra.GOLDEN)
def test_oracle_matches_golden(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "render_" + name + ".npz"))
    kind, recs = ra.make_input(name)
    assert ra.hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"]), "regenerated record stream differs from the fixture's"
    out, _ = ra.run_oracle(kind, recs)
    mask = ra.written(kind, recs)
    assert ra.digest(out, mask) == str(z["canvases_sha256"])
    last = z["last_canvas"]                      # the real RenderPCM's last canvas, drawn pixels only
    assert (_masked(out[-1], mask[-1]) == last).all(), _diff(out[-1:], last[None], mask[-1:])


def test_canvas_is_kept_between_calls(oracle_lib):
    """Frame by frame on one canvas = all frames in one go."""
    kind, recs = ra.make_input("pcm16_shrinking")
    whole, _ = ra.run_oracle(kind, recs)
    ends = np.nonzero(recs["service_type"] == ra.SRV_END_FRAME)[0]
    canvas, lo, parts = None, 0, []
    for e in ends:
        out, canvas = ra.run_oracle(kind, recs[lo:e + 1], canvas)
        parts.append(out)
        lo = e + 1
    assert (np.concatenate(parts) == whole).all()


# ---- the product's kernels in the SIMT emulator ------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emu(emu_lib):
    lib = emu_lib
    lib.sdv_engine_create.restype = C.c_void_p
    lib.sdv_engine_destroy.argtypes = [C.c_void_p]
    lib.sdv_last_error.restype = C.c_char_p
    lib.sdv_last_error.argtypes = [C.c_void_p]
    lib.sdv_vis_render_lines.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    lib.sdv_vis_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.sdv_vis_canvas_size.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return lib


def _emu_run(emu, eng, kind, recs, cap=None):
    w, h = ra.SIZE[kind]
    n = ra.n_frames(recs) if cap is None else cap
    out = np.zeros((max(n, 1), h, w), dtype=np.uint32)
    got = C.c_size_t(0)
    recs = np.ascontiguousarray(recs)
    rc = emu.sdv_vis_render_lines(eng, kind, recs.ctypes.data, len(recs), out.ctypes.data, n, C.byref(got), None)
    return rc, out[:min(got.value, n)], got.value


@pytest.mark.parametrize("name", list(ra.CASES))
def test_emu_matches_oracle(name, emu):
    kind, recs = ra.make_input(name)
    want, _ = ra.run_oracle(kind, recs)
    eng = emu.sdv_engine_create(0)
    rc, out, n = _emu_run(emu, eng, kind, recs)
    emu.sdv_engine_destroy(eng)
    assert rc == 0 and n == len(want)
    assert (out == want).all(), _diff(out, want, np.ones_like(want, dtype=bool))


def test_emu_calls_continue_on_the_kept_canvas(emu):
    """Two calls = one call; a reset in between starts from a blank canvas; a buffer that is too small is refused with the count."""
    kind, recs = ra.make_input("stc_shrinking")
    want, _ = ra.run_oracle(kind, recs)
    ends = np.nonzero(recs["service_type"] == ra.SRV_END_FRAME)[0]
    cut = ends[0] + 1
    eng = emu.sdv_engine_create(0)
    rc, a, _ = _emu_run(emu, eng, kind, recs[:cut])
    assert rc == 0
    rc, b, n = _emu_run(emu, eng, kind, recs[cut:], cap=1)
    assert rc != 0 and n == 3 and b"3 canvases" in emu.sdv_last_error(eng)
    rc, b, _ = _emu_run(emu, eng, kind, recs[cut:])
    assert rc == 0 and (np.concatenate([a, b]) == want).all()
    assert emu.sdv_vis_reset(eng, kind, None) == 0
    rc, c, _ = _emu_run(emu, eng, kind, recs[cut:])
    fresh, _ = ra.run_oracle(kind, recs[cut:])
    assert rc == 0 and (c == fresh).all()
    rc, d, n = _emu_run(emu, eng, kind, recs[cut:ends[1]])         # no frame ends in these records: nothing is handed out
    assert rc == 0 and n == 0
    w, h = C.c_uint32(0), C.c_uint32(0)
    assert emu.sdv_vis_canvas_size(kind, C.byref(w), C.byref(h)) == 0 and (w.value, h.value) == ra.SIZE[kind]
    assert emu.sdv_vis_canvas_size(10, C.byref(w), C.byref(h)) != 0          # no such canvas
    emu.sdv_engine_destroy(eng)


# ---- the product on the GPU --------------------------------------------------------------------------------------------------------
def _gpu_run(eng, kind, recs, torch):
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), recs.dtype.itemsize)).cuda()
    out = eng.vis_render_lines(kind, d, ra.n_frames(recs))
    return out.cpu().numpy().view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ra.CASES))
def test_gpu_matches_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    kind, recs = ra.make_input(name)
    want, _ = ra.run_oracle(kind, recs)
    out = _gpu_run(Engine(0), kind, recs, torch)
    assert out.shape == want.shape and (out == want).all(), _diff(out, want, np.ones_like(want, dtype=bool))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ra.GOLDEN)
def test_gpu_matches_golden_from_reference(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    z = np.load(os.path.join(GOLD, "render_" + name + ".npz"))
    kind, recs = ra.make_input(name)
    assert ra.hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"])
    out = _gpu_run(Engine(0), kind, recs, torch)
    mask = ra.written(kind, recs)
    assert ra.digest(out, mask) == str(z["canvases_sha256"])
    assert (_masked(out[-1], mask[-1]) == z["last_canvas"]).all()


@pytest.mark.gpu
def test_gpu_frames_entry_to_canvas_in_batches():
    """The records of the frame entry drawn in two calls = in one call = the oracle on the oracle's records (200 frames: the pass that
    carries undrawn rows over walks several groups of 64 frames)."""
    import torch
    from sdvpcmdecoder_amd import Engine, synth
    import oracle_run
    luma = synth.stc007_frames(200, seed=77, height=24, noise_sigma=5.0)[0]
    luma[::7, ::5] = 16                                   # some rows lost
    want_recs = oracle_run.oracle_binarize(luma, mode=2)[0]
    want, _ = ra.run_oracle(ra.STC007, want_recs)
    eng = Engine(0)
    recs, _ = eng.binarize_frames(torch.from_numpy(luma).cuda(), new_file=True)
    n_rec = recs.shape[0]
    assert bytes(recs.cpu().numpy().tobytes()) == want_recs.tobytes()
    one = eng.vis_render_lines(ra.STC007, recs, 200).cpu().numpy().view(np.uint32)
    assert (one == want).all()
    eng.vis_reset(ra.STC007)
    ends = np.nonzero(want_recs["service_type"] == ra.SRV_END_FRAME)[0]
    cut = int(ends[130]) + 1
    a = eng.vis_render_lines(ra.STC007, recs[:cut].contiguous(), 131).cpu().numpy().view(np.uint32)
    b = eng.vis_render_lines(ra.STC007, recs[cut:n_rec].contiguous(), 69).cpu().numpy().view(np.uint32)
    assert (np.concatenate([a, b]) == want).all()


# ---- the data blocks window: sdv_set_stitch_block_output + sdv_vis_render_blocks (renderNewBlock(STC007DataBlock)) -------------------------------------
@pytest.mark.ref
@pytest.mark.parametrize("name", list(ra.BLOCK_CASES))
def test_oracle_block_canvases_match_live_reference(name, oracle_lib):
    """The oracle's stitcher blocks equal the real stitcher's newBlockProcessed blocks, and their canvases the real RenderPCM's."""
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    import stitch_api as sa
    import stitch_cases as sc
    import oracle_run
    kind, blocks, per = ra.make_block_input(name)
    recs, st = sc.make_input(ra.BLOCK_CASES[name][1], lambda luma: oracle_run.oracle_binarize(luma, mode=2))
    ref_blocks = sa.run_cpu_blocks(libs.load_ref(), "ref_", recs, st)[2]
    assert blocks.tobytes() == ref_blocks.tobytes()
    out, _ = ra.run_oracle_blocks(kind, blocks, per)
    ref = ra.run_ref_blocks(kind, blocks, per)
    mask = ra.written_blocks(kind, per)
    assert (_masked(out, mask) == _masked(ref, mask)).all(), _diff(out, ref, mask)


@pytest.mark.parametrize("name", ra.BLOCK_GOLDEN)
def test_oracle_block_canvases_match_golden(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "render_" + name + ".npz"))
    kind, blocks, per = ra.make_block_input(name)
    assert ra.hashlib.sha256(blocks.tobytes()).hexdigest() == str(z["blocks_sha256"]), "the stitcher's blocks differ from the real stitcher's"
    out, _ = ra.run_oracle_blocks(kind, blocks, per)
    mask = ra.written_blocks(kind, per)
    assert ra.digest(out, mask) == str(z["canvases_sha256"])
    assert (_masked(out[-1], mask[-1]) == z["last_canvas"]).all()


def _emu_blocks(emu, name, two_calls=False):
    """The product's path in the emulator: records -> sdv_stitch_frames with a block buffer set -> sdv_vis_render_blocks."""
    import engine_api as ea
    import stitch_cases as sc
    import stitch_api as sa
    import oracle_run
    kind, want_blocks, per = ra.make_block_input(name)
    recs, st = sc.make_input(ra.BLOCK_CASES[name][1], lambda luma: oracle_run.oracle_binarize(luma, mode=2))
    lib = ea.bind(emu)
    lib.sdv_set_stitch_block_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_stitch_block_count.restype = C.c_size_t
    lib.sdv_stitch_block_count.argtypes = [C.c_void_p]
    lib.sdv_vis_render_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    eng = lib.sdv_engine_create(0)
    buf = np.zeros(len(want_blocks) + 8, dtype=sa.BLOCK_DTYPE)
    assert lib.sdv_set_stitch_block_output(eng, buf.ctypes.data, len(buf)) == 0
    ends = np.nonzero(recs["service_type"] == 5)[0]
    cuts = [0, int(ends[len(ends) // 2]) + 1, len(recs)] if two_calls else [0, len(recs)]
    w, h = ra.SIZE[kind]
    canv, got_blocks, frames = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f = ea.emu_stitch(lib, eng, recs[a:b], st if a == 0 else None)
        assert rc == 0
        nb = lib.sdv_stitch_block_count(eng)
        got_blocks.append(buf[:nb].copy())
        per_call = np.ascontiguousarray(f["blocks_total"][f["service_type"] == 0].astype(np.uint32))
        assert int(per_call.sum()) == nb
        out = np.zeros((max(len(per_call), 1), h, w), dtype=np.uint32)
        assert lib.sdv_vis_render_blocks(eng, kind, buf.ctypes.data, nb, per_call.ctypes.data, len(per_call), out.ctypes.data, len(per_call), None) == 0
        canv.append(out[:len(per_call)])
    lib.sdv_engine_destroy(eng)
    return kind, want_blocks, per, np.concatenate(got_blocks), np.concatenate(canv)


@pytest.mark.parametrize("name", list(ra.BLOCK_CASES))
def test_emu_blocks_and_their_canvases_match_oracle(name, emu):
    kind, want_blocks, per, blocks, canvases = _emu_blocks(emu, name)
    assert blocks.tobytes() == want_blocks.tobytes()
    want, _ = ra.run_oracle_blocks(kind, want_blocks, per)
    assert (canvases == want).all(), _diff(canvases, want, np.ones_like(want, dtype=bool))


def test_emu_blocks_in_two_calls(emu):
    kind, want_blocks, per, blocks, canvases = _emu_blocks(emu, "blk_burst", two_calls=True)
    assert blocks.tobytes() == want_blocks.tobytes()
    want, _ = ra.run_oracle_blocks(kind, want_blocks, per)
    assert (canvases == want).all()


def test_emu_block_buffer_too_small_is_refused_with_the_count(emu):
    import engine_api as ea
    import stitch_cases as sc
    import stitch_api as sa
    import oracle_run
    recs, st = sc.make_input("ntsc_clean", lambda luma: oracle_run.oracle_binarize(luma, mode=2))
    lib = ea.bind(emu)
    lib.sdv_set_stitch_block_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_stitch_block_count.restype = C.c_size_t
    lib.sdv_stitch_block_count.argtypes = [C.c_void_p]
    eng = lib.sdv_engine_create(0)
    buf = np.zeros(100, dtype=sa.BLOCK_DTYPE)
    assert lib.sdv_set_stitch_block_output(eng, buf.ctypes.data, len(buf)) == 0
    rc, p, f = ea.emu_stitch(lib, eng, recs, st)
    assert rc != 0 and b"data blocks are needed" in lib.sdv_last_error(eng) and lib.sdv_stitch_block_count(eng) == 2530
    lib.sdv_engine_destroy(eng)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ra.BLOCK_CASES))
def test_gpu_blocks_and_their_canvases_match_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine, StitchSettings
    import stitch_cases as sc
    import stitch_api as sa
    import oracle_run
    kind, want_blocks, per = ra.make_block_input(name)
    recs, st = sc.make_input(ra.BLOCK_CASES[name][1], lambda luma: oracle_run.oracle_binarize(luma, mode=2))
    eng = Engine(0)
    eng.set_stitch_settings(StitchSettings.from_buffer_copy(bytes(st)))
    buf = torch.zeros((len(want_blocks) + 8, 72), dtype=torch.uint8, device="cuda")
    eng.set_stitch_block_output(buf)
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 48)).cuda()
    pairs, frames = eng.stitch_frames(d)
    nb = eng.stitch_block_count()
    got = buf[:nb].cpu().numpy().reshape(-1).view(sa.BLOCK_DTYPE)
    assert got.tobytes() == want_blocks.tobytes()
    fr = frames.cpu().numpy().reshape(-1).view(sa.FRASM_DTYPE)
    per_gpu = fr["blocks_total"][fr["service_type"] == 0].astype(np.uint32)
    assert (per_gpu == per).all()
    canvases = eng.vis_render_blocks(kind, buf[:nb].contiguous(), per_gpu).cpu().numpy().view(np.uint32)
    want, _ = ra.run_oracle_blocks(kind, want_blocks, per)
    assert (canvases == want).all(), _diff(canvases, want, np.ones_like(want, dtype=bool))
    if name in ra.BLOCK_GOLDEN:
        z = np.load(os.path.join(GOLD, "render_" + name + ".npz"))
        assert ra.digest(canvases, ra.written_blocks(kind, per)) == str(z["canvases_sha256"])


# ---- the assembled-lines window: sdv_set_stitch_line_output + sdv_vis_render_asm_lines (renderNewLine(STC007Line) on the stitcher's lines) ---------------
@pytest.mark.ref
@pytest.mark.parametrize("name", list(ra.ASM_CASES))
def test_oracle_asm_canvases_match_live_reference(name, oracle_lib):
    """The oracle's assembled lines equal the real stitcher's newLineProcessed lines, and their canvases the real RenderPCM's."""
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    import stitch_api as sa
    import stitch_cases as sc
    import oracle_run
    kind, lines, per = ra.make_asm_input(name)
    recs, st = sc.make_input(ra.ASM_CASES[name][1], lambda luma: oracle_run.oracle_binarize(luma, mode=2))
    sa.run_cpu_blocks(libs.load_ref(), "ref_", recs, st)
    ref_lines, ref_per = sa.last_asm_lines(libs.load_ref(), "ref_")
    assert lines.tobytes() == ref_lines.tobytes() and per.tolist() == ref_per.tolist()
    out, _ = ra.run_oracle_asm(kind, lines, per)
    ref = ra.run_ref_asm(kind, lines, per)
    mask = ra.written_blocks(kind, per)
    assert (_masked(out, mask) == _masked(ref, mask)).all(), _diff(out, ref, mask)


@pytest.mark.parametrize("name", ra.ASM_GOLDEN)
def test_oracle_asm_canvases_match_golden(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "render_" + name + ".npz"))
    kind, lines, per = ra.make_asm_input(name)
    assert ra.hashlib.sha256(lines.tobytes()).hexdigest() == str(z["lines_sha256"]), "the stitcher's assembled lines differ from the real stitcher's"
    out, _ = ra.run_oracle_asm(kind, lines, per)
    mask = ra.written_blocks(kind, per)
    assert ra.digest(out, mask) == str(z["canvases_sha256"])
    assert (_masked(out[-1], mask[-1]) == z["last_canvas"]).all()


@pytest.mark.parametrize("name", list(ra.ASM_CASES))
def test_emu_asm_lines_and_their_canvases_match_oracle(name, emu):
    """Records -> sdv_stitch_frames with a line buffer set (two calls) -> sdv_vis_render_asm_lines, in the emulator."""
    import engine_api as ea
    import stitch_cases as sc
    import stitch_api as sa
    import oracle_run
    kind, want_lines, want_per = ra.make_asm_input(name)
    recs, st = sc.make_input(ra.ASM_CASES[name][1], lambda luma: oracle_run.oracle_binarize(luma, mode=2))
    lib = ea.bind(emu)
    lib.sdv_set_stitch_line_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_stitch_line_count.restype = C.c_size_t
    lib.sdv_stitch_line_count.argtypes = [C.c_void_p]
    lib.sdv_stitch_line_counts.restype = C.c_size_t
    lib.sdv_stitch_line_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_vis_render_asm_lines.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    eng = lib.sdv_engine_create(0)
    buf = np.zeros(len(want_lines) + 8, dtype=sa.ASM_DTYPE)
    assert lib.sdv_set_stitch_line_output(eng, buf.ctypes.data, 10) == 0
    rc, p, f = ea.emu_stitch(lib, eng, recs, st)
    assert rc != 0 and b"assembled lines are needed" in lib.sdv_last_error(eng) and lib.sdv_stitch_line_count(eng) == len(want_lines)
    assert lib.sdv_set_stitch_line_output(eng, buf.ctypes.data, len(buf)) == 0
    ends = np.nonzero(recs["service_type"] == 5)[0]
    cuts = [0, int(ends[len(ends) // 2]) + 1, len(recs)]
    w, h = ra.SIZE[kind]
    got_lines, got_per, canv = [], [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f = ea.emu_stitch(lib, eng, recs[a:b], st if a == 0 else None)
        assert rc == 0
        n = lib.sdv_stitch_line_count(eng)
        per = np.zeros(64, dtype=np.uint32)
        k = lib.sdv_stitch_line_counts(eng, per.ctypes.data, len(per))
        per = np.ascontiguousarray(per[:k])
        assert int(per.sum()) == n and k == int((f["service_type"] == 0).sum())
        got_lines.append(buf[:n].copy()); got_per += per.tolist()
        out = np.zeros((max(k, 1), h, w), dtype=np.uint32)
        assert lib.sdv_vis_render_asm_lines(eng, kind, buf.ctypes.data, n, per.ctypes.data, k, out.ctypes.data, k, None) == 0
        canv.append(out[:k])
    lib.sdv_engine_destroy(eng)
    assert np.concatenate(got_lines).tobytes() == want_lines.tobytes() and got_per == want_per.tolist()
    want, _ = ra.run_oracle_asm(kind, want_lines, want_per)
    assert (np.concatenate(canv) == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ra.ASM_CASES))
def test_gpu_asm_lines_and_their_canvases_match_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine, StitchSettings
    import stitch_cases as sc
    import stitch_api as sa
    import oracle_run
    kind, want_lines, want_per = ra.make_asm_input(name)
    recs, st = sc.make_input(ra.ASM_CASES[name][1], lambda luma: oracle_run.oracle_binarize(luma, mode=2))
    eng = Engine(0)
    eng.set_stitch_settings(StitchSettings.from_buffer_copy(bytes(st)))
    buf = torch.zeros((len(want_lines) + 8, 32), dtype=torch.uint8, device="cuda")
    blk = torch.zeros((len(want_lines) + 8, 72), dtype=torch.uint8, device="cuda")
    eng.set_stitch_line_output(buf)
    eng.set_stitch_block_output(blk)                        # both feeds at once
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 48)).cuda()
    pairs, frames = eng.stitch_frames(d)
    n = eng.stitch_line_count()
    got = buf[:n].cpu().numpy().reshape(-1).view(sa.ASM_DTYPE)
    per = eng.stitch_line_counts()
    assert got.tobytes() == want_lines.tobytes() and per.tolist() == want_per.tolist()
    assert eng.stitch_block_count() == pairs.shape[0] // 3
    canvases = eng.vis_render_asm_lines(kind, buf[:n].contiguous(), per).cpu().numpy().view(np.uint32)
    want, _ = ra.run_oracle_asm(kind, want_lines, want_per)
    assert (canvases == want).all(), _diff(canvases, want, np.ones_like(want, dtype=bool))
    if name in ra.ASM_GOLDEN:
        z = np.load(os.path.join(GOLD, "render_" + name + ".npz"))
        assert ra.digest(canvases, ra.written_blocks(kind, want_per)) == str(z["canvases_sha256"])


# ---- the two windows of the PCM-1 stitcher: sdv_set_pcm1_stitch_block_output -> sdv_vis_render_blocks(SDV_VIS_PCM1_BLOCKS), sdv_set_pcm1_stitch_line_output
# -> sdv_vis_render_lines(SDV_VIS_PCM1_ASM) -------------------------------------------------------------------------------------------------------------------
@pytest.mark.ref
@pytest.mark.parametrize("name", ra.P1VIS_CASES)
def test_oracle_pcm1_stitcher_canvases_match_live_reference(name, oracle_lib):
    """The real RenderPCM on the oracle's blocks and sub-lines (tests/test_pcm1_vis.py: those equal the real stitcher's)."""
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    blocks, per, lines = ra.make_p1vis_input(name)
    out, _ = ra.run_oracle_blocks(ra.PCM1_BLOCKS, blocks, per)
    ref = ra.run_ref_blocks(ra.PCM1_BLOCKS, blocks, per)
    mask = ra.written_p1_blocks(per)
    assert (_masked(out, mask) == _masked(ref, mask)).all(), _diff(out, ref, mask)
    out, _ = ra.run_oracle_lines_into(ra.PCM1_ASM, lines, len(per))
    ref = ra.run_ref_lines_into(ra.PCM1_ASM, lines, len(per))
    mask = ra.written_p1_asm(lines)
    assert (_masked(out, mask) == _masked(ref, mask)).all(), _diff(out, ref, mask)


@pytest.mark.parametrize("name", ra.P1VIS_GOLDEN)
def test_oracle_pcm1_stitcher_canvases_match_golden(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "render_p1vis_" + name + ".npz"))
    blocks, per, lines = ra.make_p1vis_input(name)
    out, _ = ra.run_oracle_blocks(ra.PCM1_BLOCKS, blocks, per)
    assert ra.digest(out, ra.written_p1_blocks(per)) == str(z["block_canvases_sha256"])
    out, _ = ra.run_oracle_lines_into(ra.PCM1_ASM, lines, len(per))
    mask = ra.written_p1_asm(lines)
    assert ra.digest(out, mask) == str(z["line_canvases_sha256"])
    assert (_masked(out[-1], mask[-1]) == z["last_line_canvas"]).all()


def _emu_p1vis(emu, blocks, per, lines, calls=1):
    import engine_api as ea
    lib = ea.bind(emu)
    lib.sdv_vis_render_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    eng = lib.sdv_engine_create(0)
    bc, lc = [], []
    n = len(per)
    cuts = [0, n] if calls == 1 else [0, n // 2, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        w, h = ra.SIZE[ra.PCM1_BLOCKS]
        out = np.zeros((b - a, h, w), dtype=np.uint32)
        pb, pp = np.ascontiguousarray(blocks[16 * a:16 * b]), np.ascontiguousarray(per[a:b])
        assert lib.sdv_vis_render_blocks(eng, ra.PCM1_BLOCKS, pb.ctypes.data, len(pb), pp.ctypes.data, len(pp), out.ctypes.data, len(pp), None) == 0
        bc.append(out)
        rc, out, got = _emu_run(emu, eng, ra.PCM1_ASM, np.ascontiguousarray(lines[1470 * a:1470 * b]), cap=b - a)
        assert rc == 0 and got == b - a
        lc.append(out)
    lib.sdv_engine_destroy(eng)
    return np.concatenate(bc), np.concatenate(lc)


@pytest.mark.parametrize("name", ra.P1VIS_CASES)
def test_emu_pcm1_stitcher_canvases_match_oracle(name, emu):
    blocks, per, lines = ra.make_p1vis_input(name)
    bc, lc = _emu_p1vis(emu, blocks, per, lines)
    want, _ = ra.run_oracle_blocks(ra.PCM1_BLOCKS, blocks, per)
    assert (bc == want).all(), _diff(bc, want, np.ones_like(want, dtype=bool))
    want, _ = ra.run_oracle_lines_into(ra.PCM1_ASM, lines, len(per))
    assert (lc == want).all(), _diff(lc, want, np.ones_like(want, dtype=bool))


def test_emu_pcm1_stitcher_canvases_in_two_calls(emu):
    blocks, per, lines = ra.make_p1vis_input("manual_lost_lines")
    bc, lc = _emu_p1vis(emu, blocks, per, lines, calls=2)
    want, _ = ra.run_oracle_blocks(ra.PCM1_BLOCKS, blocks, per)
    assert (bc == want).all()
    want, _ = ra.run_oracle_lines_into(ra.PCM1_ASM, lines, len(per))
    assert (lc == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ra.P1VIS_CASES)
def test_gpu_pcm1_stitcher_feeds_to_canvases_match_oracle(name):
    """Records -> sdv_pcm1_stitch_frames with both feeds set -> the two canvases, all on the device."""
    import torch
    import pcm1_api as p1
    from sdvpcmdecoder_amd import Engine, Pcm1StitchSettings
    blocks, per, lines = ra.make_p1vis_input(name)
    recs, st = p1.make_input(name)
    eng = Engine(0)
    eng.set_pcm1_stitch_settings(Pcm1StitchSettings.from_buffer_copy(bytes(st)))
    bl = torch.zeros((len(blocks) + 16, 576), dtype=torch.uint8, device="cuda")
    ln = torch.zeros((len(lines) + 1470, 16), dtype=torch.uint8, device="cuda")
    eng.set_pcm1_stitch_block_output(bl); eng.set_pcm1_stitch_line_output(ln)
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 32)).cuda()
    eng.pcm1_stitch_frames(d)
    nb, nl = eng.pcm1_stitch_block_count(), eng.pcm1_stitch_line_count()
    assert nb == len(blocks) and nl == len(lines)
    bc = eng.vis_render_blocks(ra.PCM1_BLOCKS, bl[:nb].contiguous(), per).cpu().numpy().view(np.uint32)
    lc = eng.vis_render_lines(ra.PCM1_ASM, ln[:nl].contiguous(), len(per)).cpu().numpy().view(np.uint32)
    want, _ = ra.run_oracle_blocks(ra.PCM1_BLOCKS, blocks, per)
    assert (bc == want).all(), _diff(bc, want, np.ones_like(want, dtype=bool))
    want, _ = ra.run_oracle_lines_into(ra.PCM1_ASM, lines, len(per))
    assert (lc == want).all(), _diff(lc, want, np.ones_like(want, dtype=bool))
    if name in ra.P1VIS_GOLDEN:
        z = np.load(os.path.join(GOLD, "render_p1vis_" + name + ".npz"))
        assert ra.digest(bc, ra.written_p1_blocks(per)) == str(z["block_canvases_sha256"])
        assert ra.digest(lc, ra.written_p1_asm(lines)) == str(z["line_canvases_sha256"])
