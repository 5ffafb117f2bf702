"""PCM-1 back half (PCM1DataStitcher -> PCMSamplePair).
  oracle (oracle/pcm1.c)  vs  golden fixtures of the real reference, the reference's own CRC known answer, and - when the
                              reference build is loadable - the real PCM1DataStitcher run live on every scenario;
  HIP kernel source       vs  the oracle, on the SIMT emulator (CPU) and through the C-ABI on the GPU (-m gpu)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import engine_api as ea
import libs
import pcm1_api as p1
from stitch_api import PAIR_DTYPE

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _same(pairs, frames, want_p, want_f):
    return len(pairs) == len(want_p) and pairs.tobytes() == want_p.tobytes() and len(frames) == len(want_f) and frames.tobytes() == want_f.tobytes()


def _diff(pairs, frames, want_p, want_f):
    out = [f"pairs {len(pairs)} vs {len(want_p)}, frames {len(frames)} vs {len(want_f)}"]
    for i in range(min(len(frames), len(want_f))):
        if frames[i].tobytes() != want_f[i].tobytes():
            out.append(f" frame {i}: " + str([(n, frames[i][n], want_f[i][n]) for n in p1.FRASM1_DTYPE.names if np.any(frames[i][n] != want_f[i][n])]))
    n = min(len(pairs), len(want_p))
    d = np.nonzero((pairs[:n].view(np.uint8).reshape(n, 12) != want_p[:n].view(np.uint8).reshape(n, 12)).any(axis=1))[0]
    out.append(f" {len(d)} pairs differ, first at {d[:5]}")
    return "\n".join(out)


def _oracle(name):
    recs, st = p1.make_input(name)
    pairs, frames = p1.run_cpu(libs.load_oracle(), "orc_", recs, st)
    return recs, st, pairs, frames


# ---- the oracle is pinned ------------------------------------------------------------------------------------------
def test_crc_known_answers(oracle_lib):
    """PCM1Line::calcCRC: the reference's own test line (pcmtester.cpp:14-21 -> 0x9EB9), the silent line (CRC_SILENT) and 254
    random lines whose CRC the real reference computed (tests/golden/pcm1_crc.npz)."""
    oracle_lib.orc_pcm1_crc_words.restype = C.c_uint16
    z = np.load(os.path.join(GOLD, "pcm1_crc.npz"))
    words, crc = z["words"], z["crc"]
    assert crc[0] == 0x9EB9 and crc[1] == 0xECBF
    got = np.array([oracle_lib.orc_pcm1_crc_words(np.ascontiguousarray(w).ctypes.data_as(C.POINTER(C.c_uint16))) for w in words], dtype=np.uint16)
    assert (got == crc).all()
    assert (p1.crc_words(words) == crc).all()          # the vectorised copy the input generator uses


@pytest.mark.parametrize("name", p1.GOLDEN)
def test_oracle_matches_golden(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "pcm1_" + name + ".npz"))
    recs, st, pairs, frames = _oracle(name)
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"]), "regenerated input stream differs from the fixture's"
    assert bytes(st) == z["settings"].tobytes()
    want_p = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    want_f = np.ascontiguousarray(z["frames"]).view(p1.FRASM1_DTYPE).reshape(-1)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.ref
@pytest.mark.parametrize("name", list(p1.CASES))
def test_oracle_matches_live_reference(name, oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    recs, st, pairs, frames = _oracle(name)
    rp, rf = p1.run_cpu(libs.load_ref(), "ref_", recs, st)
    assert _same(pairs, frames, rp, rf), _diff(pairs, frames, rp, rf)


def test_clean_stream_is_a_permutation_of_the_lines(oracle_lib):
    """Property: on a clean tape every block is valid and the pair stream holds every sub-line's words exactly once."""
    recs, st = p1.make_input("clean")
    pairs, frames = p1.run_cpu(oracle_lib, "orc_", recs, st)
    assert (pairs["sample_flags"] == 3).all() and (frames["blocks_drop"] == 0).all() and len(pairs) == 3 * 1470

    def sample(w):
        w = w.astype(np.uint16)
        coarse = (w & 0x1000) != 0
        fine = (w << 4).astype(np.uint16)
        c = ((w & 0x0FFF) << 2).astype(np.uint16) | np.where((w & 0x0800) != 0, 0xC000, 0).astype(np.uint16)
        return np.where(coarse, c, fine).astype(np.uint16).view(np.int16)
    data = recs[recs["service_type"] == 0]
    want = sample(data["words"][:, :6].reshape(-1, 2))
    a = np.sort(want.view(np.uint32).reshape(-1))
    b = np.sort(np.ascontiguousarray(pairs["audio_word"]).view(np.uint32).reshape(-1))
    assert (a == b).all()


# ---- the kernels on the emulator -----------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emu(emu_lib):
    return ea.bind(emu_lib)


@pytest.mark.parametrize("name", list(p1.CASES))
def test_emu_matches_oracle(name, emu, oracle_lib):
    recs, st, want_p, want_f = _oracle(name)
    eng = emu.sdv_engine_create(0)
    rc, pairs, frames = ea.emu_pcm1_stitch(emu, eng, recs, st)
    emu.sdv_engine_destroy(eng)
    assert rc == 0
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def test_emu_streaming_calls_equal_one_call(emu, oracle_lib):
    """The stream may arrive in arbitrary pieces: records wait in the engine for their END_FRAME."""
    recs, st, want_p, want_f = _oracle("file_marks")
    eng = emu.sdv_engine_create(0)
    cuts = [0, 1, 2, 300, 496, 497, 1200, 1201, 2000, len(recs) - 1, len(recs)]
    got_p, got_f = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f = ea.emu_pcm1_stitch(emu, eng, recs[a:b], st if a == 0 else None, pair_cap=8000, frame_cap=16)
        assert rc == 0
        got_p.append(p.copy())
        got_f.append(f.copy())
    emu.sdv_engine_destroy(eng)
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def test_emu_edge_inputs(emu, oracle_lib):
    eng = emu.sdv_engine_create(0)
    rc, p, f = ea.emu_pcm1_stitch(emu, eng, np.zeros(0, dtype=p1.LINE1_DTYPE))            # empty
    assert rc == 0 and len(p) == 0 and len(f) == 0
    recs, st = p1.make_input("clean")
    end = int(np.nonzero(recs["service_type"] == p1.SRV_END_FRAME)[0][0])
    rc, p, f = ea.emu_pcm1_stitch(emu, eng, recs[:end])                                  # a frame without its END_FRAME: nothing yet
    assert rc == 0 and len(p) == 0 and len(f) == 0
    rc, p, f = ea.emu_pcm1_stitch(emu, eng, recs[end:end + 1])                           # ... now it completes
    want_p, want_f = p1.run_cpu(oracle_lib, "orc_", recs[:end + 1], st)
    assert rc == 0 and _same(p, f, want_p, want_f)
    lone = recs[end:end + 1].copy()                                                      # a lone END_FRAME: an all-padding frame
    lone["frame_number"] = 9
    rc, p, f = ea.emu_pcm1_stitch(emu, eng, lone)
    want_p, want_f = p1.run_cpu(oracle_lib, "orc_", lone, st)
    assert rc == 0 and len(p) == 1470 and _same(p, f, want_p, want_f)
    rc, p, f = ea.emu_pcm1_stitch(emu, eng, recs, pair_cap=100, frame_cap=8)             # output buffer too small: reported, sized
    assert rc != 0 and b"too small" in emu.sdv_last_error(eng)
    emu.sdv_engine_destroy(eng)


def test_emu_failed_call_leaves_the_stream_untouched(emu, oracle_lib):
    """A call that is refused (buffers too small) takes nothing: lines that waited for their END_FRAME still wait and the same
    lines can be handed over again."""
    recs, st, want_p, want_f = _oracle("file_marks")
    eng = emu.sdv_engine_create(0)
    cut = 700                                                                            # inside the second frame
    rc, p0, f0 = ea.emu_pcm1_stitch(emu, eng, recs[:cut], st, pair_cap=8000, frame_cap=16)
    assert rc == 0 and len(f0) >= 1
    rc, p, f = ea.emu_pcm1_stitch(emu, eng, recs[cut:], None, pair_cap=100, frame_cap=16)
    assert rc != 0 and b"too small" in emu.sdv_last_error(eng)
    rc, p1_, f1 = ea.emu_pcm1_stitch(emu, eng, recs[cut:], None, pair_cap=8000, frame_cap=16)   # the same lines again
    assert rc == 0
    emu.sdv_engine_destroy(eng)
    pairs, frames = np.concatenate([p0, p1_]), np.concatenate([f0, f1])
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.parametrize("name", ["manual_lost_many", "manual_lost_file_marks"])
def test_emu_field_buffers_outlive_calls(name, emu):
    """Manual line offsets over damaged fields, the stream cut into calls at every frame end and in the middle of frames: the lines earlier
    frames left in the field buffers are found whether those frames came with this call or with one before."""
    recs, st, want_p, want_f = _oracle(name)
    ends = np.nonzero(recs["service_type"] == p1.SRV_END_FRAME)[0]
    cuts = sorted(set([0, int(ends[0]) + 1, int(ends[1]) + 200, int(ends[3]) + 1, len(recs)]))
    eng = emu.sdv_engine_create(0)
    pairs, frames = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        rc, p, f = ea.emu_pcm1_stitch(emu, eng, recs[a:b], st if a == 0 else None, pair_cap=20000, frame_cap=32)
        assert rc == 0, emu.sdv_last_error(eng)
        pairs.append(p); frames.append(f)
    emu.sdv_engine_destroy(eng)
    pairs, frames = np.concatenate(pairs), np.concatenate(frames)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def test_emu_refuses_what_the_reference_never_finishes(emu):
    """A line of a later frame queued ahead of an END_FRAME: the reference pops its queue up to that line (pcm1datastitcher.cpp:1634-1660),
    finds the same END_FRAME again and assembles what is left of the frame without end (shown on the real reference: more frame reports
    than the stream has frames, until the driver's buffers are full).  The product refuses the stream."""
    recs, st = p1.make_input("clean")
    recs = recs.copy()
    recs["frame_number"][100] = 2
    if libs.ref_available():
        pairs, n_frames = p1.run_cpu(libs.load_ref(), "ref_", recs, st, overflow_ok=True)
        assert pairs is None and n_frames > 3
    eng = emu.sdv_engine_create(0)
    rc, p, f = ea.emu_pcm1_stitch(emu, eng, recs, st)
    assert rc != 0 and b"later frame" in emu.sdv_last_error(eng)
    emu.sdv_engine_destroy(eng)


# ---- the product on the GPU ------------------------------------------------------------------------------------------
def _gpu_run(eng, recs, st, torch, **kw):
    from sdvpcmdecoder_amd import Pcm1StitchSettings
    s = Pcm1StitchSettings.from_buffer_copy(bytes(st))
    eng.set_pcm1_stitch_settings(s)
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 32)).cuda()
    p, f = eng.pcm1_stitch_frames(d, **kw)
    return p.cpu().numpy().reshape(-1).view(PAIR_DTYPE), f.cpu().numpy().reshape(-1).view(p1.FRASM1_DTYPE)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(p1.CASES))
def test_gpu_matches_oracle(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    recs, st, want_p, want_f = _oracle(name)
    pairs, frames = _gpu_run(Engine(0), recs, st, torch)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
@pytest.mark.parametrize("name", p1.GOLDEN)
def test_gpu_matches_golden_from_reference(name):
    import torch
    from sdvpcmdecoder_amd import Engine
    z = np.load(os.path.join(GOLD, "pcm1_" + name + ".npz"))
    recs, st = p1.make_input(name)
    assert hashlib.sha256(recs.tobytes()).hexdigest() == str(z["input_sha256"])
    pairs, frames = _gpu_run(Engine(0), recs, st, torch)
    want_p = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    want_f = np.ascontiguousarray(z["frames"]).view(p1.FRASM1_DTYPE).reshape(-1)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
def test_gpu_large_batch_in_pieces():
    """2000 frames with damage, file marks in the middle of the batch, three calls cut mid-frame: equal to the oracle's run."""
    import torch
    from sdvpcmdecoder_amd import Engine
    a = p1.make_stream(700, seed=41, p_bad=0.03, header=2, new_file=True, end_file=True)
    b = p1.make_stream(1300, seed=42, p_bad=0.01, p_picked=0.02, new_file=True, first_frame=1000)
    recs = np.concatenate([a, b])
    st = p1.default_settings()
    want_p, want_f = p1.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = Engine(0)
    cuts = [0, 123457, 600001, len(recs)]
    got_p, got_f = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        p, f = _gpu_run(eng, recs[lo:hi], st, torch)
        got_p.append(p)
        got_f.append(f)
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22, 23])
def test_gpu_random_cuts(seed):
    """A damaged PCM-1 tape with file tags handed over in pieces cut at random record positions equals the oracle's run."""
    import torch
    from sdvpcmdecoder_amd import Engine
    rng = np.random.default_rng(seed)
    recs = p1.make_stream(int(rng.integers(40, 90)), seed=seed, p_bad=0.04, header=int(rng.integers(0, 4)), p_picked=0.02, new_file=True, end_file=True)
    st = p1.default_settings(field_order=int(rng.integers(1, 3)))
    want_p, want_f = p1.run_cpu(libs.load_oracle(), "orc_", recs, st)
    eng = Engine(0)
    cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(1, len(recs)), size=9, replace=False)) + [len(recs)]
    got_p, got_f = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        p, f = _gpu_run(eng, recs[a:b], st, torch)
        got_p.append(p.copy())
        got_f.append(f.copy())
    pairs, frames = np.concatenate(got_p), np.concatenate(got_f)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)


def bin_to_line_recs(bin_recs):
    """sdv_pcm1_bin_rec -> sdv_pcm1_line_rec in numpy: what sdv_pcm1_bin_to_line_recs does on the device."""
    out = np.zeros(len(bin_recs), dtype=p1.LINE1_DTYPE)
    for nm in ("frame_number", "line_number", "words", "calc_crc", "ref_level", "picked_bits_left", "picked_bits_right", "service_type"):
        out[nm] = bin_recs[nm]
    out["flags"] = bin_recs["flags"] & (p1.LF_BW_SET | p1.LF_FORCED_BAD)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_gpu_video_to_audio(mode):
    """The whole PCM-1 path on the device - video frames -> sdv_pcm1_binarize_frames -> sdv_pcm1_bin_to_line_recs ->
    sdv_pcm1_stitch_frames - against the oracle's two halves chained on the CPU."""
    import torch
    import pcm1_frames_api as fa
    from sdvpcmdecoder_amd import Engine, synth
    luma, _ = synth.pcm1_frames(5, seed=91 + mode, height=486, noise_sigma=4.0)
    orc = libs.load_oracle()
    want_recs, _ = fa.run_cpu(orc, "orc_", luma, mode, {})
    st = p1.default_settings()
    want_p, want_f = p1.run_cpu(orc, "orc_", bin_to_line_recs(want_recs), st)
    eng = Engine(0)
    eng.setPCMType(0)              # PCM_PCM1
    eng.setBinarizationMode(mode)
    lines, _ = eng.pcm1_binarize_frames(torch.from_numpy(luma).cuda())
    assert lines.cpu().numpy().tobytes() == want_recs.tobytes()
    conv = eng.pcm1_bin_to_line_recs(lines)
    assert conv.cpu().numpy().tobytes() == bin_to_line_recs(want_recs).tobytes()
    p, f = eng.pcm1_stitch_frames(conv)
    pairs = p.cpu().numpy().reshape(-1).view(PAIR_DTYPE)
    frames = f.cpu().numpy().reshape(-1).view(p1.FRASM1_DTYPE)
    assert _same(pairs, frames, want_p, want_f), _diff(pairs, frames, want_p, want_f)
    assert len(pairs) == 5 * 1470 and (pairs["sample_flags"] & 2).mean() > 0.95


def test_manual_offsets_always_hand_the_deinterleaver_one_field():
    """PCM1Deinterleaver::processBlock answers DI_RET_NO_DATA to a queue shorter than a field (pcm1deinterleaver.cpp:104, 119).  PCM1DataStitcher never
    hands it one: with automatic offsets the padding is made to fill the field, and with manual ones (int8, setOddLineOffset / setEvenLineOffset) the
    arithmetic of findFramePadding (pcm1datastitcher.cpp:809-923, uint16 fields, int expressions) comes out at exactly 735 sub-lines for every offset,
    every bottom line and every number of lines the frame wrote - walked here in the reference's own types.  That is why the engine has no such case."""
    def u16(x):
        return x & 0xFFFF

    def cdiv(a, b):             # C division: towards zero
        q = abs(a) // b
        return q if a >= 0 else -q
    for ofs in range(-128, 128):
        top_data = 2 * ofs + 1 if ofs > 0 else 1                # odd field; the even one: 2 ofs + 2 / 2 - the same arithmetic one line down
        top_pad = 0 if ofs > 0 else u16(0 - ofs)
        for bottom in range(0, 1300):
            for cnt in (0, 1, 7, 100, 244, 245):                # lines the frame wrote into the field buffer
                data = 3 * cnt
                bp = u16(u16(cdiv(bottom - top_data, 2) + 1) + top_pad)
                if bp > 245:
                    bp = u16(bp - 245)
                    b2 = u16(bottom - bp * 2)
                    data = u16(u16(cdiv(b2 - top_data, 2) + 1) * 3)
                bot = u16(cdiv(735 - data, 3) - top_pad)
                queue = 3 * top_pad + (data if data <= 735 else 0) + 3 * bot
                assert queue >= 735, (ofs, bottom, cnt, data, top_pad, bot)
