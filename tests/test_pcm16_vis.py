"""The data blocks window of PCM-16x0: PCM16X0DataStitcher::newBlockProcessed(PCM16X0DataBlock) (pcm16x0datastitcher.cpp:5116) as
records next to the sample pairs (sdv_set_pcm16x0_stitch_block_output) and RenderPCM::renderNewBlock(PCM16X0DataBlock) (renderpcm.cpp:1403-1768) on them
(sdv_vis_render_blocks, SDV_VIS_PCM16X0_BLOCKS).
  oracle (oracle/pcm16.c, oracle/render.c)  vs  the real stitcher's blocks and the real RenderPCM fed by the real stitcher (live, when oracle/_ref is
                                                built) and the committed fixtures;
  HIP kernels                               vs  the oracle: on the emulator, and through the C-ABI on the GPU (-m gpu)."""
import ctypes as C
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
RENDER_CASES = ("si_bad10", "si_picked_forced", "si_burst", "si_file_marks", "si_rate_emph", "ei_bad10", "ei_picked", "si_lost_sublines", "si_no_p")


@functools.lru_cache(maxsize=None)        # (the tests of a case share one run of the oracle; nobody writes into what it returns)
def _oracle(name):
    recs, st = p16.make_input(name)
    pairs, frames, blocks = p16.run_cpu_vis(libs.load_oracle(), "orc_", recs, st)
    per = np.ascontiguousarray((frames["blocks_total"][frames["service_type"] == 0] // 3).astype(np.uint32))
    assert int(per.sum()) == len(blocks)
    return recs, st, pairs, frames, np.ascontiguousarray(blocks), per


def _ref_canvases(recs, st, n):
    lib = libs.load_ref()
    w, h = ra.SIZE[ra.PCM16X0_BLOCKS]
    out = np.zeros((n, h, w), dtype=np.uint32)
    f = lib.ref_vis_pcm16x0_stitch_block_canvases
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(p16.Pcm16Settings), C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rw, rh = C.c_uint32(0), C.c_uint32(0)
    recs = np.ascontiguousarray(recs)
    got = f(recs.ctypes.data, len(recs), C.byref(st), out.ctypes.data, n, C.byref(rw), C.byref(rh))
    assert got == n and (rw.value, rh.value) == (w, h), (got, n, rw.value, rh.value)
    return out


def _masked(c, m):
    return np.where(m, c, 0)


@pytest.mark.ref
@pytest.mark.parametrize("name", CASES)
def test_oracle_blocks_match_live_reference(name, oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    recs, st, pairs, frames, blocks, per = _oracle(name)
    rp, rf, rb = p16.run_cpu_vis(libs.load_ref(), "ref_", recs, st)
    assert pairs.tobytes() == rp.tobytes() and frames.tobytes() == rf.tobytes()
    assert blocks.tobytes() == rb.tobytes()


@pytest.mark.ref
@pytest.mark.parametrize("name", RENDER_CASES)
def test_oracle_canvases_match_the_real_stitcher_on_the_real_renderer(name, oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    recs, st, pairs, frames, blocks, per = _oracle(name)
    out, _ = ra.run_oracle_blocks(ra.PCM16X0_BLOCKS, blocks, per)
    ref = _ref_canvases(recs, st, len(per))
    mask = ra.written_blocks(ra.PCM16X0_BLOCKS, per)
    d = np.argwhere(_masked(out, mask) != _masked(ref, mask))
    assert len(d) == 0, (len(d), d[:4].tolist(), [hex(int(out[tuple(i)])) for i in d[:4]], [hex(int(ref[tuple(i)])) for i in d[:4]])


@pytest.mark.parametrize("name", p16.VIS_GOLDEN)
def test_oracle_matches_golden_from_reference(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "pcm16vis_" + name + ".npz"))
    recs, st, pairs, frames, blocks, per = _oracle(name)
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"])
    assert hashlib.sha256(blocks.tobytes()).hexdigest() == str(z["blocks_sha256"]), "the stitcher's blocks differ from the real stitcher's"
    out, _ = ra.run_oracle_blocks(ra.PCM16X0_BLOCKS, blocks, per)
    mask = ra.written_blocks(ra.PCM16X0_BLOCKS, per)
    assert ra.digest(out, mask) == str(z["canvases_sha256"])
    assert (_masked(out[-1], mask[-1]) == z["last_canvas"]).all()


@pytest.fixture(scope="module")
def emu(emu_lib):
    return ea.bind(emu_lib)


def _emu_canvases(lib, eng, blocks, per):
    lib.sdv_vis_render_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    w, h = ra.SIZE[ra.PCM16X0_BLOCKS]
    out = np.zeros((max(len(per), 1), h, w), dtype=np.uint32)
    assert lib.sdv_vis_render_blocks(eng, ra.PCM16X0_BLOCKS, blocks.ctypes.data, len(blocks), per.ctypes.data, len(per), out.ctypes.data, len(per), None) == 0
    return out[:len(per)]


@pytest.mark.parametrize("name", CASES)
def test_emu_blocks_match_oracle(name, emu, oracle_lib):
    recs, st, want_p, want_f, want_b, per = _oracle(name)
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames, blocks = ea.emu_pcm16_stitch_vis(emu, eng, recs, st)
    assert rc == 0 and pairs.tobytes() == want_p.tobytes() and frames.tobytes() == want_f.tobytes()
    assert blocks.tobytes() == want_b.tobytes(), [(f, np.argwhere(blocks[f] != want_b[f])[:4].tolist()) for f in blocks.dtype.names if blocks[f].tobytes() != want_b[f].tobytes()]
    if name in RENDER_CASES:
        got = _emu_canvases(emu, eng, np.ascontiguousarray(blocks), per)
        want, _ = ra.run_oracle_blocks(ra.PCM16X0_BLOCKS, want_b, per)
        assert (got == want).all(), np.argwhere(got != want)[:4].tolist()
    emu.sdv_engine_destroy(eng)


def test_emu_blocks_in_calls_and_too_small(emu, oracle_lib):
    recs, st, want_p, want_f, want_b, per = _oracle("si_file_marks")
    eng = emu.sdv_engine_create(0)
    cuts = [0, len(recs) // 3, len(recs) // 3 + 11, len(recs)]
    got = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f, bl = ea.emu_pcm16_stitch_vis(emu, eng, recs[a:b], st if a == 0 else None)
        assert rc == 0
        got.append(bl)
    assert np.concatenate(got).tobytes() == want_b.tobytes()
    emu.sdv_engine_destroy(eng)
    eng = emu.sdv_engine_create(0)
    rc, p, f, bl = ea.emu_pcm16_stitch_vis(emu, eng, recs, st, block_cap=7)
    assert rc != 0 and b"blocks for the visualiser" in emu.sdv_last_error(eng) and ea.emu_pcm16_stitch_vis.last_count == len(want_b)
    rc, p, f, bl = ea.emu_pcm16_stitch_vis(emu, eng, recs, st)                 # the refused call took nothing: once more with room
    assert rc == 0 and bl.tobytes() == want_b.tobytes()
    emu.sdv_engine_destroy(eng)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_blocks_and_canvases_match_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine, Pcm16x0StitchSettings
    recs, st, want_p, want_f, want_b, per = _oracle(name)
    eng = Engine(0)
    eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    bl = torch.zeros((len(want_b) + 64, 32), dtype=torch.uint8, device="cuda")
    eng.set_pcm16x0_stitch_block_output(bl)
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 36)).cuda()
    p, f = eng.pcm16x0_stitch_frames(d)
    nb = eng.pcm16x0_stitch_block_count()
    got = bl[:nb].cpu().numpy().reshape(-1).view(p16.VBLOCK16_DTYPE)
    assert p.cpu().numpy().tobytes() == want_p.tobytes() and got.tobytes() == want_b.tobytes()
    canvases = eng.vis_render_blocks(ra.PCM16X0_BLOCKS, bl[:nb].contiguous(), per).cpu().numpy().view(np.uint32)
    want, _ = ra.run_oracle_blocks(ra.PCM16X0_BLOCKS, want_b, per)
    assert (canvases == want).all(), np.argwhere(canvases != want)[:4].tolist()
    if name in p16.VIS_GOLDEN:
        z = np.load(os.path.join(GOLD, "pcm16vis_" + name + ".npz"))
        assert hashlib.sha256(got.tobytes()).hexdigest() == str(z["blocks_sha256"])
        assert ra.digest(canvases, ra.written_blocks(ra.PCM16X0_BLOCKS, per)) == str(z["canvases_sha256"])
