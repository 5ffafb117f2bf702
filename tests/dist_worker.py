"""Worker of tests/test_sharded.py: one rank of a gloo process group decoding its range of one tape on the emulator build."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    import ctypes as C
    import torch.distributed as dist
    from sdvpcmdecoder_amd import synth
    from sdvpcmdecoder_amd.sharded import ShardedDecoder, torch_all_gather
    from emu_engine_adapter import EmuEngine
    import stitch_api as sa
    out_dir, n_frames, warmup, s_warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group(backend="gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    if warmup < 0:
        return loop_main(out_dir, n_frames, rank, world)
    if len(sys.argv) > 5:
        return pcm_main(out_dir, n_frames, warmup, s_warm, sys.argv[5], rank, world)
    luma, _, _ = synth.stc007_frames(n_frames, seed=41, noise_sigma=3.0)          # every rank renders the same tape ...
    eng = EmuEngine(C.CDLL(os.path.join(HERE, "emu", "libsdvpcm_emu.so")))
    eng.set_stitch_settings(sa.default_settings())
    dec = ShardedDecoder(eng, rank, world, torch_all_gather(None), height=luma.shape[1], warmup=warmup, stitch_warmup=s_warm)
    f0, f1 = dec.frames_needed(n_frames)
    pairs, frames = dec.decode(luma[f0:f1], n_frames, first_frame_no=1)           # ... and is only given its part of it
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pairs=pairs.view(np.uint8).reshape(len(pairs), 12),
             frames=frames.view(np.uint8).reshape(len(frames), 64), redo=np.array([dec.stats["binarize_redo"], dec.stats["stitch_redo"], dec.stats["gathers"]]))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def pcm_tape(fmt, n_frames):
    """The tape of the PCM-1 / PCM-16x0 sharding tests (every rank renders the same one)."""
    from sdvpcmdecoder_amd import synth
    if fmt == "pcm1":
        return synth.pcm1_frames(n_frames, seed=43, height=486, noise_sigma=3.0)[0]
    return synth.pcm16x0_tape_frames(n_frames, seed=44, ei=(fmt == "pcm16x0_ei"))[0]


def pcm_main(out_dir, n_frames, warmup, s_warm, fmt, rank, world):
    """ShardedPcmDecoder: one PCM-1 / PCM-16x0 tape over the ranks."""
    import ctypes as C
    import torch.distributed as dist
    from sdvpcmdecoder_amd.sharded import ShardedPcmDecoder, torch_all_gather
    from emu_engine_adapter import EmuEngine
    import pcm1_api as p1
    import pcm16_api as p16
    luma = pcm_tape(fmt, n_frames)
    eng = EmuEngine(C.CDLL(os.path.join(HERE, "emu", "libsdvpcm_emu.so")))
    eng.lib.sdv_set_pcm_type.argtypes = [C.c_void_p, C.c_int, C.c_int]
    st = p1.default_settings() if fmt == "pcm1" else p16.default_settings(format=1 if fmt == "pcm16x0_ei" else 0)
    dec = ShardedPcmDecoder(eng, rank, world, torch_all_gather(None), height=luma.shape[1], fmt="pcm1" if fmt == "pcm1" else "pcm16x0", stitch_settings=st,
                            warmup=warmup, stitch_warmup=s_warm)
    f0, f1 = dec.frames_needed(n_frames)
    pairs, frames = dec.decode(luma[f0:f1], n_frames, first_frame_no=1)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pairs=pairs.view(np.uint8).reshape(len(pairs), 12),
             frames=frames.view(np.uint8).reshape(len(frames), frames.dtype.itemsize),
             redo=np.array([dec.stats["binarize_redo"], dec.stats["stitch_redo"], dec.stats["gathers"]]))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def loop_main(out_dir, n_frames, rank, world):
    """ShardedBinarizeLoop: three batches of one continuing tape, each batch split over the ranks."""
    import ctypes as C
    import torch.distributed as dist
    from sdvpcmdecoder_amd import synth
    from sdvpcmdecoder_amd.sharded import ShardedBinarizeLoop, torch_all_gather, shard_bounds
    from emu_engine_adapter import EmuEngine
    eng = EmuEngine(C.CDLL(os.path.join(HERE, "emu", "libsdvpcm_emu.so")))
    loop = ShardedBinarizeLoop(eng, rank, world, torch_all_gather(None))
    lo, hi = shard_bounds(n_frames, rank, world)
    out = []
    for batch in range(3):
        luma, _, _ = synth.stc007_frames(n_frames, seed=50 + batch, height=60, noise_sigma=3.0, x0=12 + 9 * batch, x1=700 - 5 * batch)
        recs, stats = loop.step(luma[lo:hi], first_frame_no=1 + batch * n_frames + lo, new_file=(batch == 0))
        out.append(recs.copy())
    np.savez(os.path.join(out_dir, f"loop{rank}.npz"), redo=loop.redo, **{f"b{i}": o.view(np.uint8).reshape(len(o), 48) for i, o in enumerate(out)})
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
