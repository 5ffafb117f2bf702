"""Timing driver of the AudioProcessor stage: the PCMSamplePair stream of N frames (1470 pairs each, one file with its tags) in three
states of wear -> sdv_audio_process, `reps` timed calls each.  Prints wall time per call, pairs/s, frames/s and the algorithmic-bytes rate
(12 B read + 12 B written per pair); with `cpu` as third argument the real reference's AudioProcessor (oracle/_ref, when it loads)
and the oracle are timed on the first frames of the same tapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import audio_api as A, libs
from sdvpcmdecoder_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cpu = len(sys.argv) > 3 and sys.argv[3] == 'cpu'
npairs = n * 1470


def tapes():
    rng = np.random.default_rng(9)
    yield "clean", A.tape(["N", A.audio(npairs, 1, tone=False), "E"])
    starts = np.sort(rng.integers(1000, npairs - 5000, max(1, n // 25)))          # a dropout every 25 frames
    yield "dropouts", A.tape(["N", A.audio(npairs, 2, tone=False, runs=[(int(s), int(rng.integers(1, 700)), int(rng.integers(0, 3))) for s in starts]), "E"])
    yield "worn", A.tape(["N", A.audio(npairs, 3, tone=False, p_bad=0.01), "E"])    # an invalid word in every window


eng = Engine(0)
eng.set_audio_masking(A.DROP_INTER_LIN_WORD)
out_p = torch.empty((npairs + 1024, 12), dtype=torch.uint8, device='cuda')
out_u = torch.empty((16, 16), dtype=torch.uint8, device='cuda')
for name, pairs in tapes():
    d = torch.from_numpy(pairs.view(np.uint8).reshape(len(pairs), 12)).cuda()
    for it in range(reps):
        eng.reset_audio()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o, p, m = eng.audio_process(d, stop=True, out_pairs=out_p, out_purges=out_u)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} n={n} it={it}: wall {dt*1e3:.3f} ms, {npairs/dt/1e6:.0f} M pairs/s, {n/dt/1e3:.0f} k frames/s, {24*npairs/dt/1e9:.0f} GB/s algorithmic, out {o.shape[0]} masked {m}", flush=True)
    if cpu:
        k = min(n, 200) * 1470
        sample = A.tape(["N", pairs[1:1 + k], "E"])
        ends = np.array([len(sample)], dtype=np.uint64)
        got = o[:k].cpu().numpy().view(A.PAIR_DTYPE).reshape(-1)
        for label, lib, prefix in (("oracle", libs.load_oracle(), "orc_"),) + ((("reference", libs.load_ref(), "ref_"),) if libs.ref_available() else ()):
            t0 = time.perf_counter()
            r = A.run_cpu(lib, prefix, sample, A.DROP_INTER_LIN_WORD, ends, 1)
            dt = time.perf_counter() - t0
            same = r[0][:k - 600].tobytes() == got[:k - 600].tobytes()
            print(f"  cpu {label}: {k} pairs in {dt*1e3:.1f} ms = {k/dt/1e6:.2f} M pairs/s, {k/1470/dt/1e3:.1f} k frames/s (incl. the driver's idle wait for the reference); head matches GPU: {same}", flush=True)
