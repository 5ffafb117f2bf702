"""Soak run on the GPU box: random PCM-1 tapes stitched with manual line offsets - long fields with lines lost, so that the stitcher reads its
field buffers past what a frame wrote (lines of earlier frames) - cut into calls at random places, against the CPU oracle.
usage: soak_pcm1_manual.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import libs, pcm1_api as p1
from sdvpcmdecoder_amd import Engine, Pcm1StitchSettings
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
orc = libs.load_oracle()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    lines = (int(rng.integers(236, 330)), int(rng.integers(236, 330)))
    nf = int(rng.integers(3, 40))
    recs = p1.make_stream(nf, seed=seed0 + case, lines=lines, p_bad=0.03, header=int(rng.integers(0, 3)), new_file=bool(case % 3 == 0), end_file=bool(case % 3 == 0),
                          noise_lines=int(rng.choice([0, 0, 4])))
    st = p1.default_settings(auto_offset=int(rng.random() < 0.2), odd_offset=int(rng.integers(-8, 9)), even_offset=int(rng.integers(-8, 9)), use_ecc=int(rng.integers(0, 2)),
                             field_order=int(rng.integers(1, 3)))
    share = float(rng.choice([0.0, 0.02, 0.1, 0.3, 0.6]))
    recs = recs[~((recs["service_type"] == 0) & (rng.random(len(recs)) < share))]
    want_p, want_f = p1.run_cpu(orc, "orc_", recs, st)
    eng = Engine(0)
    eng.set_pcm1_stitch_settings(Pcm1StitchSettings.from_buffer_copy(bytes(st)))
    cuts = [0] + sorted(int(x) for x in rng.choice(np.arange(1, len(recs) - 1), size=int(rng.integers(0, 5)), replace=False)) + [len(recs)]
    d = torch.from_numpy(np.ascontiguousarray(recs).view(np.uint8).reshape(len(recs), 32)).cuda()
    ps, fs = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        p, f = eng.pcm1_stitch_frames(d[a:b].contiguous())
        ps.append(p.cpu().numpy()); fs.append(f.cpu().numpy())
    ok = np.concatenate(ps).tobytes() == want_p.tobytes() and np.concatenate(fs).tobytes() == want_f.tobytes()
    print(f"case {case}: {nf} frames of {lines} lines, {share:.0%} lost, offsets auto={st.auto_offset} {st.odd_offset}/{st.even_offset}, {len(cuts) - 1} calls, {len(want_p)} pairs -> {'OK' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
print("soak ok")
