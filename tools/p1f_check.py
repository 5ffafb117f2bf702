"""PCM-1 / PCM-16x0 frame drivers on the GPU against the oracle on the input of the *_frames_prof.py scripts (full-height frames, all modes).
usage: p1f_check.py [frames]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import libs
import pcm1_frames_api as p1, pcm16_frames_api as p16
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
orc = libs.load_oracle()
for name, pf, gen, call in (("pcm1", p1, synth.pcm1_frames, "pcm1_binarize_frames"), ("pcm16x0", p16, synth.pcm16x0_frames, "pcm16x0_binarize_frames")):
    base, _ = gen(8, seed=530, height=486, noise_sigma=4.0)
    luma = np.tile(base, ((n + 7) // 8, 1, 1))[:n]
    d = torch.from_numpy(luma).to("cuda:0")
    for mode in (0, 1, 2):
        eng = Engine(0)
        eng.setBinarizationMode(mode)
        lines, stats = getattr(eng, call)(d, first_frame_no=1, new_file=True)
        torch.cuda.synchronize()
        got = lines.cpu().numpy().tobytes()
        want, wstats = pf.run_cpu(orc, "orc_", luma, mode, dict(new_file=True))
        dt = want.dtype
        g = np.frombuffer(got, dtype=dt)
        same = got == want.tobytes() and stats.cpu().numpy().tobytes() == wstats.tobytes()
        print(name, "mode", mode, "same" if same else "DIFF", "valid", int(((want["flags"] & 64) != 0).sum()), "of", len(want), "gpu valid", int(((g["flags"] & 64) != 0).sum()), flush=True)
        if not same:
            for i in range(min(len(g), len(want))):
                if g[i].tobytes() != want[i].tobytes():
                    print(i, g[i], want[i], sep="\n"); break
