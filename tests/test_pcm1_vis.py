"""What PCM1DataStitcher hands to the visualiser: newBlockProcessed(PCM1DataBlock) and newLineProcessed(PCM1SubLine)
(pcm1datastitcher.cpp:1333, :1392-1407) as records next to the sample pairs - sdv_set_pcm1_stitch_block_output / _line_output.
  oracle (oracle/pcm1.c)  vs  the real PCM1DataStitcher's signals (live, when oracle/_ref is built) and the committed fixtures;
  HIP kernel              vs  the oracle: on the emulator, and through the C-ABI on the GPU (-m gpu)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import engine_api as ea
import libs
import pcm1_api as p1

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = list(p1.CASES)
STALE = [n for n in CASES if n.startswith("manual_")]        # fields that read what earlier frames left in the field buffers


def _oracle(name):
    recs, st = p1.make_input(name)
    return (recs, st) + p1.run_cpu_vis(libs.load_oracle(), "orc_", recs, st)


@pytest.mark.ref
@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_live_reference(name, oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    recs, st, pairs, frames, blocks, lines = _oracle(name)
    rp, rf, rb, rl = p1.run_cpu_vis(libs.load_ref(), "ref_", recs, st)
    assert pairs.tobytes() == rp.tobytes() and frames.tobytes() == rf.tobytes()
    assert len(blocks) == 16 * int((frames["service_type"] == 0).sum()) and len(lines) == 1470 * int((frames["service_type"] == 0).sum())
    assert p1.comparable_blocks(blocks).tobytes() == p1.comparable_blocks(rb).tobytes()
    # the reference hands over the sub-lines that carry the frame's number, in queue order: the places that are not marked
    assert lines[lines["flags"] != p1.P1S_SKIP].tobytes() == rl.tobytes()


@pytest.mark.parametrize("name", p1.VIS_GOLDEN)
def test_oracle_matches_golden_from_reference(name):
    z = np.load(os.path.join(GOLD, "pcm1vis_" + name + ".npz"))
    recs, st, pairs, frames, blocks, lines = _oracle(name)
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"])
    assert p1.comparable_blocks(blocks).tobytes() == np.ascontiguousarray(z["blocks"]).tobytes()
    assert lines[lines["flags"] != p1.P1S_SKIP].tobytes() == np.ascontiguousarray(z["lines"]).tobytes()


def _check(name, blocks, lines, want_b, want_l):
    # a block that starts on a line an earlier frame left behind carries that frame's number in the reference; the engine keeps no frame numbers
    # for those lines (include/sdvpcm.h) - nothing but the number differs, and only with manual offsets over short fields
    got_b, wb = p1.comparable_blocks(blocks, stale_frames=name in STALE), p1.comparable_blocks(want_b, stale_frames=name in STALE)
    assert got_b.tobytes() == wb.tobytes(), [(f, np.argwhere(got_b[f] != wb[f])[:4].tolist()) for f in got_b.dtype.names if got_b[f].tobytes() != wb[f].tobytes()]
    assert lines.tobytes() == want_l.tobytes(), [(f, np.argwhere(lines[f] != want_l[f])[:4].tolist()) for f in lines.dtype.names if lines[f].tobytes() != want_l[f].tobytes()]


@pytest.fixture(scope="module")
def emu(emu_lib):
    return ea.bind(emu_lib)


@pytest.mark.parametrize("name", CASES)
def test_emu_matches_oracle(name, emu, oracle_lib):
    recs, st, want_p, want_f, want_b, want_l = _oracle(name)
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames, blocks, lines = ea.emu_pcm1_stitch_vis(emu, eng, recs, st)
    emu.sdv_engine_destroy(eng)
    assert rc == 0 and pairs.tobytes() == want_p.tobytes() and frames.tobytes() == want_f.tobytes()
    _check(name, blocks, lines, want_b, want_l)


def test_emu_feeds_in_calls_and_too_small(emu, oracle_lib):
    """The feeds over a stream cut into calls (frames complete in later calls), one feed alone, and buffers that are too small."""
    recs, st, want_p, want_f, want_b, want_l = _oracle("file_marks")
    eng = emu.sdv_engine_create(0)
    cuts = [0, len(recs) // 3, len(recs) // 3 + 7, len(recs)]
    got_b, got_l = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f, bl, ln = ea.emu_pcm1_stitch_vis(emu, eng, recs[a:b], st if a == 0 else None)
        assert rc == 0
        got_b.append(bl); got_l.append(ln)
    _check("file_marks", np.concatenate(got_b), np.concatenate(got_l), want_b, want_l)
    emu.sdv_engine_destroy(eng)
    eng = emu.sdv_engine_create(0)
    rc, p, f, bl, ln = ea.emu_pcm1_stitch_vis(emu, eng, recs, st, lines=False)
    assert rc == 0 and len(ln) == 0 and ea.emu_pcm1_stitch_vis.last_counts == (len(want_b), 0)
    _check("file_marks", bl, want_l, want_b, want_l)
    rc, p, f, bl, ln = ea.emu_pcm1_stitch_vis(emu, eng, recs, st, block_cap=5)
    assert rc != 0 and b"visualiser buffers too small" in emu.sdv_last_error(eng) and ea.emu_pcm1_stitch_vis.last_counts == (len(want_b), len(want_l))
    emu.sdv_engine_destroy(eng)


def _gpu_run(eng, recs, st, torch):
    from sdvpcmdecoder_amd import Pcm1StitchSettings
    eng.set_pcm1_stitch_settings(Pcm1StitchSettings.from_buffer_copy(bytes(st)))
    nfr = int((recs["service_type"] == p1.SRV_END_FRAME).sum()) + 2
    bl = torch.zeros((nfr * 16, 576), dtype=torch.uint8, device="cuda")
    ln = torch.zeros((nfr * 1470, 16), dtype=torch.uint8, device="cuda")
    eng.set_pcm1_stitch_block_output(bl); eng.set_pcm1_stitch_line_output(ln)
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 32)).cuda()
    p, f = eng.pcm1_stitch_frames(d)
    nb, nl = eng.pcm1_stitch_block_count(), eng.pcm1_stitch_line_count()
    return (p.cpu().numpy().reshape(-1).view(p1.PAIR_DTYPE), f.cpu().numpy().reshape(-1).view(p1.FRASM1_DTYPE),
            bl[:nb].cpu().numpy().reshape(-1).view(p1.BLOCK1_DTYPE), ln[:nl].cpu().numpy().reshape(-1).view(p1.ASM1_DTYPE))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    recs, st, want_p, want_f, want_b, want_l = _oracle(name)
    pairs, frames, blocks, lines = _gpu_run(Engine(0), recs, st, torch)
    assert pairs.tobytes() == want_p.tobytes() and frames.tobytes() == want_f.tobytes()
    _check(name, blocks, lines, want_b, want_l)


@pytest.mark.gpu
@pytest.mark.parametrize("name", p1.VIS_GOLDEN)
def test_gpu_matches_golden_from_reference(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    z = np.load(os.path.join(GOLD, "pcm1vis_" + name + ".npz"))
    recs, st = p1.make_input(name)
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"])
    pairs, frames, blocks, lines = _gpu_run(Engine(0), recs, st, torch)
    assert p1.comparable_blocks(blocks).tobytes() == np.ascontiguousarray(z["blocks"]).tobytes()
    assert lines[lines["flags"] != p1.P1S_SKIP].tobytes() == np.ascontiguousarray(z["lines"]).tobytes()
