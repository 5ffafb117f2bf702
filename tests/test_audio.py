"""AudioProcessor + SamplesToWAV (PCMSamplePair stream -> masked PCMSamplePair stream -> WAV bytes).
  oracle (oracle/audio.c)  vs  golden fixtures of the real reference (pairs, sample indices, newSource positions, mask counts, the
                               WAV files the reference wrote) and - when the reference build is loadable - the real AudioProcessor
                               run live on every scenario;
  HIP kernel source        vs  the oracle, on the SIMT emulator (CPU) and through the C-ABI on the GPU (-m gpu)."""
import ctypes as C
import hashlib
import os
import tempfile

import numpy as np
import pytest

import audio_api as A
import engine_api as ea
import libs
from stitch_api import PAIR_DTYPE

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUPPORTED = list(A.CASES)        # (round 3: the worker's two dead ends are reproduced, nothing is refused any more)


def _diff(out, want):
    n = min(len(out), len(want))
    d = np.nonzero((out[:n].view(np.uint8).reshape(n, 12) != want[:n].view(np.uint8).reshape(n, 12)).any(axis=1))[0]
    return f"pairs {len(out)} vs {len(want)}; {len(d)} differ, first at {d[:8]}: got {out[d[:3]]} want {want[d[:3]]}"


def _oracle(name):
    pairs, mode, ends, stop = A.make_input(name)
    return (pairs, mode, ends, stop) + A.run_cpu(libs.load_oracle(), "orc_", pairs, mode, ends, stop)


# ---- the oracle is pinned ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", A.GOLDEN)
def test_oracle_matches_golden(name, oracle_lib):
    z = np.load(os.path.join(GOLD, "audio_" + name + ".npz"))
    pairs, mode, ends, stop, out, idx, pur, masked, hit = _oracle(name)
    assert hashlib.sha256(pairs.tobytes()).hexdigest() == str(z["input_sha256"]), "regenerated input stream differs from the fixture's"
    assert mode == int(z["mode"]) and stop == int(z["stop"]) and np.array_equal(ends, z["ends"])
    want = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    assert out.tobytes() == want.tobytes(), _diff(out, want)
    assert np.array_equal(idx, z["index"]) and np.array_equal(pur["first_pair"], z["purges"]) and masked == int(z["masked"]) and hit == (1 if name in A.DEAD_ENDS else 0)
    # the index is implied by the purge positions (what the C-ABI relies on), and the kinds / tag positions are consistent
    assert np.array_equal(idx, A.expected_index(len(out), pur["first_pair"]))
    tags = np.nonzero(pairs["service_type"])[0]
    if name not in A.DEAD_ENDS:         # (there a tag may pass without a purge, or never be read)
        assert np.array_equal(pur["tag_index"][pur["kind"] != A.PURGE_STOP], tags) and np.array_equal(pur["kind"][pur["kind"] != A.PURGE_STOP], pairs["service_type"][tags])
    # the WAV files the reference wrote
    files = dict(A.wav_files(oracle_lib, "orc_", out, pur))
    ref_files = {int(k[3:]): z[k].tobytes() for k in z.files if k.startswith("wav")}
    assert files == ref_files


@pytest.mark.ref
@pytest.mark.parametrize("name", list(A.CASES))
def test_oracle_matches_live_reference(name, oracle_lib):
    if not libs.ref_available():
        pytest.skip("reference build (oracle/_ref) not available")
    pairs, mode, ends, stop, out, idx, pur, masked, hit = _oracle(name)
    with tempfile.TemporaryDirectory() as d:
        r_out, r_idx, r_pur, r_masked, _ = A.run_cpu(libs.load_ref(), "ref_", pairs, mode, ends, stop, wav_dir=d)
        ref_files = {int(f[3:].split("_")[0]): open(os.path.join(d, f), "rb").read() for f in os.listdir(d)}
    assert out.tobytes() == r_out.tobytes(), _diff(out, r_out)
    assert np.array_equal(idx, r_idx) and np.array_equal(pur["first_pair"], r_pur) and masked == r_masked
    assert hit == (1 if name in A.DEAD_ENDS else 0)
    if stop:        # (without it the driver still has to stop the worker to end the run: the files then hold what stop() flushed as well)
        assert dict(A.wav_files(oracle_lib, "orc_", out, pur)) == ref_files


def _e2e_audio_fixture():
    z = np.load(os.path.join(GOLD, "e2e_ntsc_file_audio.npz"))
    want = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    return z, want, {int(k[3:]): z[k].tobytes() for k in z.files if k.startswith("wav")}


def test_oracle_whole_file_to_wav_matches_reference_golden(oracle_lib):
    """The pair stream of the end-to-end fixture (video -> both real workers, e2e_ntsc_file.npz) -> oracle AudioProcessor -> WAV bytes ==
    what the real AudioProcessor put out and the real SamplesToWAV wrote for it."""
    z, want, wavs = _e2e_audio_fixture()
    pairs = np.ascontiguousarray(np.load(os.path.join(GOLD, "e2e_ntsc_file.npz"))["pairs"]).view(PAIR_DTYPE).reshape(-1)
    assert hashlib.sha256(pairs.tobytes()).hexdigest() == str(z["input_sha256"])
    out, idx, pur, masked, hit = A.run_cpu(oracle_lib, "orc_", pairs, A.DROP_INTER_LIN_WORD, np.array([len(pairs)], dtype=np.uint64), 1)
    assert out.tobytes() == want.tobytes(), _diff(out, want)
    assert np.array_equal(pur["first_pair"], z["purges"]) and masked == int(z["masked"]) and masked > 0
    assert dict(A.wav_files(oracle_lib, "orc_", out, pur)) == wavs and len(wavs) == 1


def test_clean_stream_passes_unchanged(oracle_lib):
    """Property: a stream without invalid samples leaves as it came, behind one silent pair per file and without each file's last pair."""
    pairs, mode, ends, stop, out, idx, pur, masked, hit = _oracle("clean")
    data = pairs[pairs["service_type"] == 0]
    assert masked == 0 and len(out) == len(data) and (out[0]["audio_word"] == 0).all() and out[1:].tobytes() == data[:-1].tobytes()


def test_masking_is_idempotent(oracle_lib):
    """Property: what the worker put out holds no invalid sample, so a second pass alters nothing (it only shifts by its own silent pair)."""
    pairs, mode, ends, stop, out, idx, pur, masked, hit = _oracle("long_runs_lin")
    assert ((out["sample_flags"] & A.SF_WORD_VALID) != 0).all()
    again = A.tape(["N", out, "E"])
    out2, _, _, masked2, _ = A.run_cpu(oracle_lib, "orc_", again, mode, np.array([len(again)], dtype=np.uint64), 1)
    assert masked2 == 0 and out2[1:].tobytes() == out[:-1].tobytes()


def test_wav_header_known_answers(oracle_lib):
    """SamplesToWAV's header: RIFF sizes, rate and byte rate as updateHeader writes them (samples2wav.cpp:111-206)."""
    oracle_lib.orc_wav_header.argtypes = [C.c_void_p, C.c_uint64, C.c_uint16]
    for n, rate, want_rate in ((1, 44056, 44056), (1470, 44100, 44100), (100000, 48000, 44100)):
        h = np.zeros(44, dtype=np.uint8)
        oracle_lib.orc_wav_header(h.ctypes.data, n, rate)
        b = h.tobytes()
        assert b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and b[36:40] == b"data"
        assert int.from_bytes(b[4:8], "little") == 36 + 4 * n and int.from_bytes(b[40:44], "little") == 4 * n
        assert int.from_bytes(b[24:28], "little") == want_rate and int.from_bytes(b[28:32], "little") == 4 * want_rate
        assert b[20:24] == bytes([1, 0, 2, 0]) and b[32:36] == bytes([4, 0, 16, 0])


# ---- the kernels on the emulator -----------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emu(emu_lib):
    return A.bind_product(ea.bind(emu_lib))


@pytest.mark.parametrize("name", SUPPORTED)
def test_emu_matches_oracle(name, emu, oracle_lib):
    pairs, mode, ends, stop, want, idx, want_pur, want_masked, hit = _oracle(name)
    out, pur, masked = A.emu_run(emu, pairs, mode, ends, stop)
    assert out.tobytes() == want.tobytes(), _diff(out, want)
    assert pur.tobytes() == want_pur.tobytes() and masked == want_masked


def test_emu_follows_the_reference_into_its_dead_ends(emu, oracle_lib):
    emu.sdv_audio_stalled.argtypes = [C.c_void_p]
    eng = emu.sdv_engine_create(0)
    emu.sdv_set_audio_masking(eng, A.DROP_INTER_LIN_WORD)
    # a stream that starts with invalid samples and no NEW_FILE tag: the reference's window fills up and nothing can ever leave
    bad = A.audio(2000, 77, runs=[(0, 5, 2)])
    rc, out, pur, masked, n_out, n_pur = A.emu_audio(emu, eng, bad, 0)
    assert rc == 0 and n_out == 0 and n_pur == 0 and emu.sdv_audio_pending(eng) == 512 and emu.sdv_audio_stalled(eng) == 1
    want = A.run_cpu(oracle_lib, "orc_", bad, A.DROP_INTER_LIN_WORD, np.array([len(bad)], dtype=np.uint64), 0)
    assert want[4] == 1 and len(want[0]) == 0
    # ... nothing is taken any more, tags included
    rc, out, pur, masked, n_out, n_pur = A.emu_audio(emu, eng, A.tape(["N", A.audio(700, 3), "E"]), 0)
    assert rc == 0 and n_out == 0 and n_pur == 0 and emu.sdv_audio_pending(eng) == 512 and emu.sdv_audio_stalled(eng) == 1
    # ... until stop() purges the window as it is
    rc, out, pur, masked, n_out, n_pur = A.emu_audio(emu, eng, A.audio(10, 4), 1)
    want = A.run_cpu(oracle_lib, "orc_", bad, A.DROP_INTER_LIN_WORD, np.array([len(bad)], dtype=np.uint64), 1)
    assert rc == 0 and out.tobytes() == want[0].tobytes() and n_out == 511 and n_pur == 1 and pur[0]["kind"] == A.PURGE_STOP and pur[0]["first_pair"] == 511
    assert emu.sdv_audio_stalled(eng) == 0 and emu.sdv_audio_pending(eng) == 1
    # a failed call takes nothing, also when an END_FILE that does not purge has split it into spans
    pairs, mode, ends, stop = A.make_input("tiny_files_bad")
    emu.sdv_reset_audio(eng)
    rc, out, pur, masked, n_out, n_pur = A.emu_audio(emu, eng, pairs, 1, out_cap=300)
    assert rc == -1 and b"too small" in emu.sdv_last_error(eng) and emu.sdv_audio_pending(eng) == 0 and emu.sdv_audio_next_index(eng) == 0
    rc, out, pur, masked, n_out, n_pur = A.emu_audio(emu, eng, pairs, 1)
    want = A.run_cpu(oracle_lib, "orc_", pairs, mode, ends, stop)
    assert rc == 0 and out.tobytes() == want[0].tobytes() and pur.tobytes() == want[2].tobytes() and masked == want[3]
    # a tag that is neither NEW_FILE nor END_FILE (PCMSamplePair has no such type)
    odd = A.tape(["N", A.audio(100, 78), A.tag(7), A.audio(100, 79)])
    rc = A.emu_audio(emu, eng, odd, 0)[0]
    assert rc == -4 and b"neither NEW_FILE nor END_FILE" in emu.sdv_last_error(eng)
    # the engine is still usable
    pairs, mode, ends, stop = A.make_input("short_runs_lin")
    emu.sdv_reset_audio(eng)
    rc, out, pur, masked, _, _ = A.emu_audio(emu, eng, pairs, 1)
    want = A.run_cpu(oracle_lib, "orc_", pairs, mode, ends, stop)
    assert rc == 0 and out.tobytes() == want[0].tobytes()
    emu.sdv_engine_destroy(eng)


def test_emu_edge_inputs(emu, oracle_lib):
    eng = emu.sdv_engine_create(0)
    emu.sdv_set_audio_masking(eng, A.DROP_HOLD_WORD)
    rc, out, pur, masked, _, _ = A.emu_audio(emu, eng, np.zeros(0, dtype=PAIR_DTYPE), 0)           # empty
    assert rc == 0 and len(out) == 0 and len(pur) == 0 and emu.sdv_audio_pending(eng) == 0
    rc, out, pur, masked, _, _ = A.emu_audio(emu, eng, A.tag(A.SRV_NEW_FILE), 0)                   # a lone NEW_FILE: the silent pair waits
    assert rc == 0 and len(out) == 0 and len(pur) == 1 and pur[0]["kind"] == A.PURGE_NEW_FILE and emu.sdv_audio_pending(eng) == 1
    few = A.audio(100, 5, runs=[(50, 10, 2)])
    rc, out, pur, masked, _, _ = A.emu_audio(emu, eng, few, 0)                                    # fewer than 227 pairs: no scan yet
    assert rc == 0 and len(out) == 0 and emu.sdv_audio_pending(eng) == 101 and emu.sdv_audio_next_index(eng) == 0
    rc, out, pur, masked, _, _ = A.emu_audio(emu, eng, A.tag(A.SRV_END_FILE), 0)                   # the end of the file flushes it
    want = A.run_cpu(oracle_lib, "orc_", A.tape(["N", few, "E"]), A.DROP_HOLD_WORD, np.array([1, 101, 102], dtype=np.uint64), 0)
    assert rc == 0 and out.tobytes() == want[0].tobytes() and masked == want[3] and len(out) == 100 and emu.sdv_audio_pending(eng) == 1
    assert emu.sdv_audio_next_index(eng) == 0
    # output buffers too small: reported with the sizes, nothing taken, repeatable
    pairs, mode, ends, stop = A.make_input("two_files")
    emu.sdv_reset_audio(eng)
    rc, out, pur, masked, n_out, n_pur = A.emu_audio(emu, eng, pairs, 1, out_cap=100)
    assert rc == -1 and b"too small" in emu.sdv_last_error(eng) and n_out == 3400 and n_pur == 5 and emu.sdv_audio_pending(eng) == 0
    rc, out, pur, masked, n_out, n_pur = A.emu_audio(emu, eng, pairs, 1, purges_cap=2)
    assert rc == -1 and n_out == 3400 and n_pur == 5
    rc, out, pur, masked, n_out, n_pur = A.emu_audio(emu, eng, pairs, 1, out_cap=n_out, purges_cap=n_pur)
    want = A.run_cpu(oracle_lib, "orc_", pairs, A.DROP_HOLD_WORD, ends, 1)
    assert rc == 0 and out.tobytes() == want[0].tobytes() and pur.tobytes() == want[2].tobytes()
    assert emu.sdv_set_audio_masking(eng, 7) == -1
    emu.sdv_engine_destroy(eng)


def test_emu_in_place_and_staged_paths_agree(emu, oracle_lib):
    """With room for the whole burst the pairs are worked on in place in the caller's buffer; with exactly the room the output needs they go through
    a buffer of the engine.  Same result, burst by burst."""
    for name in ("short_runs_lin", "window_edges", "bursts_long_runs", "worn_tape", "no_first_tag"):
        pairs, mode, ends, stop, want, idx, want_pur, want_masked, hit = _oracle(name)
        eng = emu.sdv_engine_create(0)
        emu.sdv_set_audio_masking(eng, mode)
        outs, a = [], 0
        for k, b in enumerate(ends):
            b = int(b)
            st = 1 if (stop and k + 1 == len(ends)) else 0
            rc, o, _, _, n_out, n_pur = A.emu_audio(emu, eng, pairs[a:b], st, out_cap=0)          # refused: tells the size, takes nothing
            if rc == 0:                 # ... unless the burst puts nothing out
                assert n_out == 0
            else:
                assert rc == -1 and n_out > 0
                rc, o, p, m, n_out2, _ = A.emu_audio(emu, eng, pairs[a:b], st, out_cap=n_out, purges_cap=max(n_pur, 1))
                assert rc == 0 and n_out2 == n_out
            outs.append(o.copy()); a = b
        emu.sdv_engine_destroy(eng)
        out = np.concatenate(outs)
        assert out.tobytes() == want.tobytes(), name + ": " + _diff(out, want)


def test_emu_random_tapes(emu, oracle_lib):
    """Differential fuzz (tools/audio_fuzz.py): 60 random tapes - files of random length, dropout runs of random place / length / channel, random
    invalid words and blocks, tags in random order, random bursts, every masking mode - the kernels against the oracle.  (The same generator
    ran 150 tapes through the real AudioProcessor against the oracle, and 400 through the emulator, when this was written.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("audio_fuzz", os.path.join(os.path.dirname(GOLD), "..", "tools", "audio_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    res = [fz.check(emu, oracle_lib, seed) for seed in range(5000, 5060)]
    assert res.count("ok") >= 55


def test_emu_next_index_runs_on_across_calls(emu, oracle_lib):
    """sdv_audio_next_index before a call = PCMSample::index of the first pair that call puts out (the reference's own index, from the oracle run)."""
    for name in ("bursts_long_runs", "bursts_two_files", "bursts_small"):
        pairs, mode, ends, stop, want, idx, want_pur, want_masked, hit = _oracle(name)
        eng = emu.sdv_engine_create(0)
        emu.sdv_set_audio_masking(eng, mode)
        a = got = 0
        for k, b in enumerate(ends):
            b = int(b)
            before = emu.sdv_audio_next_index(eng)
            rc, o, p, m, _, _ = A.emu_audio(emu, eng, pairs[a:b], 1 if (stop and k + 1 == len(ends)) else 0)
            assert rc == 0
            if len(o) and (len(p) == 0 or p[0]["first_pair"] > 0):
                assert before == idx[got], (name, k)
            got += len(o); a = b
        emu.sdv_engine_destroy(eng)


def test_emu_call_in_place(emu, oracle_lib):
    """out_pairs == pairs: the output overwrites the input."""
    for name in ("short_runs_lin", "two_files", "worn_tape"):
        pairs, mode, ends, stop, want, idx, want_pur, want_masked, hit = _oracle(name)
        eng = emu.sdv_engine_create(0)
        emu.sdv_set_audio_masking(eng, mode)
        buf = np.concatenate([pairs, np.zeros(1024, dtype=PAIR_DTYPE)])
        pur = np.zeros(16, dtype=A.PURGE_DTYPE)
        n_out, n_pur, nm = C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
        rc = emu.sdv_audio_process(eng, buf.ctypes.data, len(pairs), 1, buf.ctypes.data, len(buf), C.byref(n_out), pur.ctypes.data, 16, C.byref(n_pur), C.byref(nm), None)
        emu.sdv_engine_destroy(eng)
        assert rc == 0 and buf[:n_out.value].tobytes() == want.tobytes() and nm.value == want_masked, name


def test_emu_many_files_in_one_call(emu, oracle_lib):
    """The stretches between tags are independent: 40 short files in one burst, one wave each."""
    rng = np.random.default_rng(5)
    parts = []
    for k in range(40):
        n = int(rng.integers(3, 1500))
        runs = [(int(rng.integers(0, max(1, n - 1))), int(rng.integers(1, 300)), int(rng.integers(0, 3))) for _ in range(3)]
        parts += ["N", A.audio(n, 100 + k, runs=runs)] + (["E"] if k % 3 else [])
    pairs = A.tape(parts)
    ends = np.array([len(pairs)], dtype=np.uint64)
    want = A.run_cpu(oracle_lib, "orc_", pairs, A.DROP_INTER_LIN_WORD, ends, 1)
    assert want[4] == 0
    out, pur, masked = A.emu_run(emu, pairs, A.DROP_INTER_LIN_WORD, ends, 1)
    assert out.tobytes() == want[0].tobytes(), _diff(out, want[0])
    assert pur.tobytes() == want[2].tobytes() and masked == want[3]


def test_emu_wav_pack_and_header(emu, oracle_lib):
    pairs, mode, ends, stop, out, idx, pur, masked, hit = _oracle("two_files")
    eng = emu.sdv_engine_create(0)
    pcm = np.zeros((len(out), 2), dtype="<i2")
    assert emu.sdv_wav_pack(eng, out.ctypes.data, len(out), pcm.ctypes.data, None) == 0
    assert np.array_equal(pcm, out["audio_word"])
    emu.sdv_engine_destroy(eng)
    assert dict(A.wav_files(emu, "sdv_", out, pur)) == dict(A.wav_files(oracle_lib, "orc_", out, pur))


# ---- the product on the GPU ------------------------------------------------------------------------------------------
def _gpu_run(eng, pairs, mode, ends, stop):
    import torch
    eng.set_audio_masking(mode)
    eng.reset_audio()
    outs, purs, masked, a, got = [], [], 0, 0, 0
    d = torch.from_numpy(np.ascontiguousarray(pairs).view(np.uint8).reshape(len(pairs), 12)).to("cuda:0")
    for k, b in enumerate(ends):
        b = int(b)
        o, p, m = eng.audio_process(d[a:b], stop=bool(stop and k + 1 == len(ends)))
        p = p.cpu().numpy().view(A.PURGE_DTYPE).reshape(-1).copy()
        p["first_pair"] += got
        p["tag_index"] += a
        o = o.cpu().numpy().view(PAIR_DTYPE).reshape(-1)
        outs.append(o); purs.append(p); masked += m
        got += len(o); a = b
    return np.concatenate(outs), np.concatenate(purs), masked


@pytest.mark.gpu
@pytest.mark.parametrize("name", SUPPORTED)
def test_gpu_matches_oracle(name, oracle_lib):
    from sdvpcmdecoder_amd import Engine
    pairs, mode, ends, stop, want, idx, want_pur, want_masked, hit = _oracle(name)
    eng = Engine(0)
    out, pur, masked = _gpu_run(eng, pairs, mode, ends, stop)
    assert out.tobytes() == want.tobytes(), _diff(out, want)
    assert pur.tobytes() == want_pur.tobytes() and masked == want_masked


@pytest.mark.gpu
@pytest.mark.parametrize("name", A.GOLDEN)
def test_gpu_matches_golden(name):
    """The product against what the real reference put out and wrote to disk (no oracle in between)."""
    import torch
    from sdvpcmdecoder_amd import Engine
    z = np.load(os.path.join(GOLD, "audio_" + name + ".npz"))
    pairs, mode, ends, stop = A.make_input(name)
    eng = Engine(0)
    out, pur, masked = _gpu_run(eng, pairs, mode, ends, stop)
    want = np.ascontiguousarray(z["pairs"]).view(PAIR_DTYPE).reshape(-1)
    assert out.tobytes() == want.tobytes(), _diff(out, want)
    assert np.array_equal(pur["first_pair"], z["purges"]) and masked == int(z["masked"])
    assert np.array_equal(A.expected_index(len(out), pur["first_pair"]), z["index"])
    d_out = torch.from_numpy(out.view(np.uint8).reshape(len(out), 12)).to("cuda:0")
    files = eng.wav_files(d_out, pur)
    assert files == {int(k[3:]): z[k].tobytes() for k in z.files if k.startswith("wav")}


@pytest.mark.gpu
def test_gpu_video_to_wav_matches_reference_golden():
    """The whole chain on the device, buffers handed from stage to stage: video -> sdv_binarize_frames -> sdv_stitch_frames ->
    sdv_audio_process -> sdv_wav_pack, against the file the real reference's four stages (VideoToDigital, STC007DataStitcher,
    AudioProcessor, SamplesToWAV) wrote for the same video."""
    import torch
    import test_stitch_kernel as tsk
    from sdvpcmdecoder_amd import Engine
    luma, _, _, _ = tsk._e2e_fixture()
    z, want, wavs = _e2e_audio_fixture()
    eng = Engine(0)
    lines, _ = eng.binarize_frames(torch.from_numpy(luma).cuda(), first_frame_no=1, new_file=True, end_file=True)
    eng.set_stitch_settings(tsk._settings(tsk.sa.default_settings()))
    pairs, _ = eng.stitch_frames(lines)
    eng.set_audio_masking(A.DROP_INTER_LIN_WORD)
    out, pur, masked = eng.audio_process(pairs, stop=True)
    got = out.cpu().numpy().view(PAIR_DTYPE).reshape(-1)
    assert got.tobytes() == want.tobytes(), _diff(got, want)
    assert masked == int(z["masked"])
    assert eng.wav_files(out, pur) == wavs


@pytest.mark.gpu
def test_gpu_output_buffer_grows_on_demand(oracle_lib):
    """Engine.audio_process with a buffer that is too small: the refused call takes nothing, the wrapper comes again with the sizes it was told."""
    import torch
    from sdvpcmdecoder_amd import Engine
    pairs, mode, ends, stop, want, idx, want_pur, want_masked, hit = _oracle("two_files")
    eng = Engine(0)
    eng.set_audio_masking(mode)
    d = torch.from_numpy(np.ascontiguousarray(pairs).view(np.uint8).reshape(len(pairs), 12)).to("cuda:0")
    o, p, m = eng.audio_process(d, stop=True, out_pairs=torch.empty((10, 12), dtype=torch.uint8, device="cuda:0"), out_purges=torch.empty((1, 16), dtype=torch.uint8, device="cuda:0"))
    assert o.cpu().numpy().tobytes() == want.tobytes() and p.cpu().numpy().tobytes() == want_pur.tobytes() and m == want_masked


@pytest.mark.gpu
def test_gpu_random_tapes(oracle_lib):
    """The differential fuzz of tools/audio_fuzz.py on the GPU: 300 random tapes against the oracle."""
    import importlib.util
    from sdvpcmdecoder_amd import Engine
    spec = importlib.util.spec_from_file_location("audio_fuzz", os.path.join(os.path.dirname(GOLD), "..", "tools", "audio_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    eng = Engine(0)
    n_ok = 0
    for seed in range(7000, 7300):
        pairs, mode, ends, stop = fz.random_case(seed)
        want = A.run_cpu(oracle_lib, "orc_", pairs, mode, ends, stop)
        if want[4]:
            continue
        out, pur, masked = _gpu_run(eng, pairs, mode, ends, stop)
        assert out.tobytes() == want[0].tobytes() and pur.tobytes() == want[2].tobytes() and masked == want[3], (seed, mode, list(ends), stop)
        n_ok += 1
    assert n_ok >= 280


@pytest.mark.gpu
def test_gpu_long_tape_properties(oracle_lib):
    """BASELINE's 10 000-frame size (14.7 M pairs): sparse damage; the oracle finishes this in a second, so it is compared in full,
    plus the size-independent properties: nothing invalid leaves, clean stretches pass unchanged, a second pass alters nothing."""
    import torch
    from sdvpcmdecoder_amd import Engine
    n = 14_700_000
    rng = np.random.default_rng(9)
    starts = np.sort(rng.integers(1000, n - 5000, 400))
    runs = [(int(s), int(rng.integers(1, 700)), int(rng.integers(0, 3))) for s in starts]
    data = A.audio(n, 9, runs=runs, tone=False)
    pairs = A.tape(["N", data, "E"])
    ends = np.array([len(pairs)], dtype=np.uint64)
    eng = Engine(0)
    out, pur, masked = _gpu_run(eng, pairs, A.DROP_INTER_LIN_WORD, ends, 1)
    want = A.run_cpu(oracle_lib, "orc_", pairs, A.DROP_INTER_LIN_WORD, ends, 1)
    assert out.tobytes() == want[0].tobytes() and pur.tobytes() == want[2].tobytes() and masked == want[3]
    assert ((out["sample_flags"] & A.SF_WORD_VALID) != 0).all() and len(out) == n
    untouched = (out["sample_flags"] & A.SF_WORD_MASKED) == 0
    assert (out["audio_word"][1:][untouched[1:]] == data["audio_word"][:-1][untouched[1:]]).all()
    out2, pur2, masked2 = _gpu_run(eng, A.tape(["N", out, "E"]), A.DROP_INTER_LIN_WORD, np.array([len(out) + 2], dtype=np.uint64), 1)
    assert masked2 == 0 and out2[1:].tobytes() == out[:-1].tobytes()
