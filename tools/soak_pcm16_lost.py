"""Soak run on the GPU box: random PCM-16x0 tapes (SI / EI) with sub-lines lost and doubled at random places - frames of the wrong size, conv_queue's
remainder carried from frame to frame - cut into calls at random places, against the CPU oracle.  usage: soak_pcm16_lost.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import libs, pcm16_api as p16
from sdvpcmdecoder_amd import Engine, Pcm16x0StitchSettings
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
orc = libs.load_oracle()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    ei = bool(rng.integers(0, 2))
    n = int(rng.integers(6, 60))
    kw = dict(seed=seed0 + case, ei=ei, cut=(int(rng.integers(0, 9)), int(rng.integers(0, 9))), tail_cut=(int(rng.integers(0, 6)), int(rng.integers(0, 6))),
              p_bad=float(rng.choice([0.0, 0.02, 0.1])), new_file=bool(rng.integers(0, 2)), end_file=bool(rng.integers(0, 2)))
    recs, _ = p16.make_stream(n, **kw)
    recs = p16.mangle(recs, seed=case, drop=int(rng.integers(0, 40 * n // 6)), dup=int(rng.integers(0, 10)))
    st = p16.default_settings(format=2 if ei else 1, p_correction=int(rng.integers(0, 2)), use_ecc=int(rng.integers(0, 2)))
    t0 = time.time()
    want_p, want_f = p16.run_cpu(orc, "orc_", recs, st)
    t_cpu = time.time() - t0
    eng = Engine(0)
    eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(1, len(recs) - 1), size=int(rng.integers(0, 4)), replace=False)) + [len(recs)]
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 36)).cuda()
    ps, fs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        p, f = eng.pcm16x0_stitch_frames(d[a:b].contiguous())
        ps.append(p.cpu().numpy()); fs.append(f.cpu().numpy())
    ok = np.concatenate(ps).tobytes() == want_p.tobytes() and np.concatenate(fs).tobytes() == want_f.tobytes()
    print(f"case {case}: {'ei' if ei else 'si'} {n} frames, {len(recs)} records in {len(cuts) - 1} calls, {len(want_p)} pairs, cpu {t_cpu:.1f}s -> {'OK' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
print("soak ok")
