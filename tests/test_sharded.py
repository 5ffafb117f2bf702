"""One tape sharded over two ranks (gloo, CPU, emulator build of the kernels): the concatenated per-rank output equals the
sequential decode of the whole tape by the oracle, whether the ranks' state predictions hold or have to be repaired."""
import os
import subprocess
import sys

import numpy as np
import pytest

import libs
import stitch_api as sa
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import synth
from sdvpcmdecoder_amd.sharded import shard_bounds

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_cover_the_tape():
    for n in (1, 7, 10000, 100003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


@pytest.mark.parametrize("n_frames,warmup,s_warm,expect_redo", [(6, 3, 2, False), (10, 3, 2, True), (6, 3, 0, False)])
def test_two_ranks_one_tape(tmp_path, emu_lib, oracle_lib, n_frames, warmup, s_warm, expect_redo):
    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(tmp_path), str(n_frames), str(warmup), str(s_warm)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    # the sequential truth: the whole file through the oracle's two workers
    luma, _, _ = synth.stc007_frames(n_frames, seed=41, noise_sigma=3.0)
    recs, _ = oracle_binarize(luma, mode=2, new_file=True, end_file=True)
    want_p, want_f = sa.run_cpu(libs.load_oracle(), "orc_", recs, sa.default_settings())
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    pairs = np.concatenate([np.ascontiguousarray(z["pairs"]).view(sa.PAIR_DTYPE).reshape(-1) for z in parts])
    frames = np.concatenate([np.ascontiguousarray(z["frames"]).view(sa.FRASM_DTYPE).reshape(-1) for z in parts])
    assert len(pairs) == len(want_p) and pairs.tobytes() == want_p.tobytes()
    assert len(frames) == len(want_f) and frames.tobytes() == want_f.tobytes()
    # a warm-up shorter than the predecessor's history cannot reproduce its coordinate history: the repair has to run
    assert (parts[1]["redo"][0] >= 1) == expect_redo, parts[1]["redo"]
    if s_warm == 0:         # a stitcher that starts cold cannot have guessed its predecessor's state: the range runs again from the true one
        assert parts[1]["redo"][1] >= 1, parts[1]["redo"]


def test_two_ranks_binarize_loop(tmp_path, emu_lib, oracle_lib):
    """What bench.py --gpus N times: batches of one continuing tape, each split over the ranks.  The geometry of the video changes
    from batch to batch, so ranks that simply carry on from their own previous state guess wrong and have to repair."""
    import ctypes as C
    n_frames, world = 4, 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(31500 + os.getpid() % 2000), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(tmp_path), str(n_frames), "-1", "0"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    lib = libs.load_oracle()
    lib.orc_v2d_new.restype = C.c_void_p
    h = C.c_void_p(lib.orc_v2d_new())
    lib.orc_v2d_set_mode.argtypes = [C.c_void_p, C.c_int]
    lib.orc_v2d_set_mode(h, 2)
    parts = [np.load(os.path.join(tmp_path, f"loop{r}.npz")) for r in range(world)]
    for batch in range(3):
        luma, _, _ = synth.stc007_frames(n_frames, seed=50 + batch, height=60, noise_sigma=3.0, x0=12 + 9 * batch, x1=700 - 5 * batch)
        want, _ = oracle_binarize(luma, handle=h, new_file=(batch == 0), first_frame_no=1 + batch * n_frames)
        got = np.concatenate([np.ascontiguousarray(z[f"b{batch}"]).view(libs.LINE_DTYPE).reshape(-1) for z in parts])
        assert got.tobytes() == want.tobytes(), f"batch {batch}"
    assert int(parts[1]["redo"]) >= 1


@pytest.mark.parametrize("fmt,n_frames,warmup,s_warm", [("pcm1", 6, 2, 2), ("pcm16x0", 6, 2, 2), ("pcm16x0_ei", 5, 1, 1), ("pcm1", 4, 0, 0), ("pcm16x0", 4, 0, 0)])
def test_two_ranks_one_pcm_tape(tmp_path, emu_lib, oracle_lib, fmt, n_frames, warmup, s_warm):
    """ShardedPcmDecoder: a PCM-1 / PCM-16x0 tape over two ranks equals the sequential decode by the oracle's two workers."""
    import dist_worker
    import pcm1_api as p1
    import pcm16_api as p16
    import pcm1_frames_api as p1f
    import pcm16_frames_api as p16f
    from test_pcm1 import bin_to_line_recs
    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(33500 + os.getpid() % 2000), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(tmp_path), str(n_frames), str(warmup), str(s_warm), fmt],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    luma = dist_worker.pcm_tape(fmt, n_frames)
    orc = libs.load_oracle()
    if fmt == "pcm1":
        recs, _ = p1f.run_cpu(orc, "orc_", luma, 2, dict(new_file=True, end_file=True))
        want_p, want_f = p1.run_cpu(orc, "orc_", bin_to_line_recs(recs), p1.default_settings())
        fdt = p1.FRASM1_DTYPE
    else:
        recs, _ = p16f.run_cpu(orc, "orc_", luma, 2, dict(new_file=True, end_file=True))
        want_p, want_f = p16.run_cpu(orc, "orc_", recs, p16.default_settings(format=1 if fmt == "pcm16x0_ei" else 0))
        fdt = p16.FRASM16_DTYPE
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    pairs = np.concatenate([np.ascontiguousarray(z["pairs"]).view(sa.PAIR_DTYPE).reshape(-1) for z in parts])
    frames = np.concatenate([np.ascontiguousarray(z["frames"]).view(fdt).reshape(-1) for z in parts])
    assert len(pairs) == len(want_p) and pairs.tobytes() == want_p.tobytes()
    assert len(frames) == len(want_f) and frames.tobytes() == want_f.tobytes()
    assert all(int(z["redo"][2]) >= 1 for z in parts)
    if warmup == 0:         # no warm-up, no prediction: the second rank has to take its predecessor's real state and decode again
        assert int(parts[1]["redo"][0]) >= 1 and (fmt == "pcm1" or int(parts[1]["redo"][1]) >= 1), parts[1]["redo"]


# ---- the same loop in C++ (examples/decode_tape_sharded.cpp) ---------------------------------------------------------------------------------
def _sequential_truth(luma):
    recs, _ = oracle_binarize(luma, mode=2, new_file=True, end_file=True)
    return sa.run_cpu(libs.load_oracle(), "orc_", recs, sa.default_settings())


@pytest.mark.parametrize("n_frames,warmup,s_warm,expect_redo", [(6, 3, 2, False), (10, 3, 2, True)])
def test_cpp_host_program_two_ranks_one_tape(tmp_path, emu_lib, oracle_lib, n_frames, warmup, s_warm, expect_redo):
    """The C++ host program of the sharded decode, built against the emulator build of the engine, two processes, the all-gather through
    files: warm-up, all-gather of the 120-byte / 3.8 KB states through sdv_get_/set_*_state, verification and repair - the concatenated
    output is the oracle's sequential decode of the whole file, and the same as the Python harness gives (test_two_ranks_one_tape)."""
    from sdvpcmdecoder_amd import build as b
    exe = b.build_example_sharded_emu()
    luma, _, _ = synth.stc007_frames(n_frames, seed=41, noise_sigma=3.0)
    n, h, w = luma.shape
    (tmp_path / "luma.raw").write_bytes(np.ascontiguousarray(luma).tobytes())
    os.makedirs(tmp_path / "comm")
    world = 2
    procs = [subprocess.Popen([exe, str(tmp_path / "luma.raw"), str(w), str(h), str(n), str(tmp_path / "out"), "file:" + str(tmp_path / "comm"), str(warmup), str(s_warm)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world)), stdout=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    want_p, want_f = _sequential_truth(luma)
    pairs = b"".join((tmp_path / f"out.rank{r}.pairs").read_bytes() for r in range(world))
    frames = b"".join((tmp_path / f"out.rank{r}.frames").read_bytes() for r in range(world))
    assert pairs == want_p.tobytes() and frames == want_f.tobytes()
    # a warm-up shorter than the predecessor's history cannot reproduce its coordinate history: the repair of the binarize stage has to run
    assert ("binarize 0," not in outs[1]) == expect_redo, outs[1]


@pytest.mark.gpu
def test_cpp_host_program_sharded_rccl_one_rank(tmp_path):
    """The product build on the GPU with one rank: ncclCommInitRank / ncclAllGather of the state blobs (the RCCL plumbing; more ranks need
    the multi-GPU node), output equal to the reference's for the file (tests/golden/e2e_ntsc_file.npz)."""
    from sdvpcmdecoder_amd import build as b
    import test_stitch_kernel as tsk
    exe = b.build_example_sharded()
    luma, z, want_p, want_f = tsk._e2e_fixture()
    n, h, w = luma.shape
    (tmp_path / "luma.raw").write_bytes(np.ascontiguousarray(luma).tobytes())
    out = subprocess.run([exe, str(tmp_path / "luma.raw"), str(w), str(h), str(n), str(tmp_path / "out"), "rccl"],
                         env=dict(os.environ, RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "1 all-gathers" in out.stdout or "2 all-gathers" in out.stdout
    assert (tmp_path / "out.rank0.pairs").read_bytes() == want_p.tobytes()
    assert (tmp_path / "out.rank0.frames").read_bytes() == want_f.tobytes()


@pytest.mark.gpu
def test_cpp_host_program_sharded_two_ranks_one_gpu(tmp_path):
    """Two ranks of the product build sharing the one GPU of the test box, the all-gather through files (RCCL wants a GPU per rank): the
    device-side loop with real hand-over between the ranks."""
    from sdvpcmdecoder_amd import build as b
    import test_stitch_kernel as tsk
    exe = b.build_example_sharded()
    luma, z, want_p, want_f = tsk._e2e_fixture()
    n, h, w = luma.shape
    (tmp_path / "luma.raw").write_bytes(np.ascontiguousarray(luma).tobytes())
    os.makedirs(tmp_path / "comm")
    procs = [subprocess.Popen([exe, str(tmp_path / "luma.raw"), str(w), str(h), str(n), str(tmp_path / "out"), "file:" + str(tmp_path / "comm")],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0"), stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert b"".join((tmp_path / f"out.rank{r}.pairs").read_bytes() for r in range(2)) == want_p.tobytes()
    assert b"".join((tmp_path / f"out.rank{r}.frames").read_bytes() for r in range(2)) == want_f.tobytes()


@pytest.mark.gpu
def test_cpp_host_program_sharded_noisy_tape_decodes_no_range_twice(tmp_path):
    """A tape that plays with noise on it: the binarizer's levels are what the first lines of the tape measured (sticky), not what a warm-up further
    down would measure.  Rank 0 publishes its state after its first frames, rank 1 warms up from it: no range is decoded twice, and the two parts
    are the sequential decode."""
    from sdvpcmdecoder_amd import build as b
    exe = b.build_example_sharded()
    n = 150                 # (75 frames per rank: the stitcher's statistics rings, 65 deep, are full where rank 1 takes over, as its warm-up assumes)
    luma, _, _ = synth.stc007_frames(n, seed=77, noise_sigma=5.0)
    recs, _ = oracle_binarize(luma, mode=2, new_file=True, end_file=True)
    want_p, want_f = sa.run_cpu(libs.load_oracle(), "orc_", recs, sa.default_settings())
    _, h, w = luma.shape
    (tmp_path / "luma.raw").write_bytes(np.ascontiguousarray(luma).tobytes())
    os.makedirs(tmp_path / "comm")
    procs = [subprocess.Popen([exe, str(tmp_path / "luma.raw"), str(w), str(h), str(n), str(tmp_path / "out"), "file:" + str(tmp_path / "comm")],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0"), stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ranges decoded again: binarize 0, stitch 0" in o for o in outs), outs
    assert b"".join((tmp_path / f"out.rank{r}.pairs").read_bytes() for r in range(2)) == want_p.tobytes()
    assert b"".join((tmp_path / f"out.rank{r}.frames").read_bytes() for r in range(2)) == want_f.tobytes()


@pytest.mark.gpu
def test_cpp_host_program_sharded_rccl_two_ranks(tmp_path):
    """Two ranks, a GPU each, ncclAllGather over the node's links (the multi-GPU node only: skipped on the one-GPU test boxes).  Run twice with the
    same output prefix and different run ids: the id file of the first run must not be picked up by the second (the rendezvous is per run)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from sdvpcmdecoder_amd import build as b
    import test_stitch_kernel as tsk
    exe = b.build_example_sharded()
    luma, z, want_p, want_f = tsk._e2e_fixture()
    n, h, w = luma.shape
    (tmp_path / "luma.raw").write_bytes(np.ascontiguousarray(luma).tobytes())
    for run in ("first", "second"):
        procs = [subprocess.Popen([exe, str(tmp_path / "luma.raw"), str(w), str(h), str(n), str(tmp_path / "out"), "rccl", "1", "1"],
                                  env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", SDV_RUN_ID=run, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
        outs = [p.communicate(timeout=600) for p in procs]
        assert all(p.returncode == 0 for p in procs), outs
        assert not (tmp_path / "out.ncclid").exists()           # removed once both ranks had joined
        pairs = b"".join((tmp_path / f"out.rank{r}.pairs").read_bytes() for r in range(2))
        frames = b"".join((tmp_path / f"out.rank{r}.frames").read_bytes() for r in range(2))
        assert pairs == want_p.tobytes() and frames == want_f.tobytes()
