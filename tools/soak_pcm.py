"""Soak run of the marker-less formats on the GPU box: random PCM-1 / PCM-16x0 (SI, EI) tapes - noise, dropouts, random lengths - through
sdv_decode_frames with the audio stage behind it, against the oracle's three workers run one after the other.  Not part of the test suite
(the oracle's frame drivers take a second per dozen frames); prints one line per case and exits non-zero on the first difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import libs, audio_api as au
import pcm1_api as p1, pcm16_api as p16, pcm1_frames_api as p1f, pcm16_frames_api as p16f
from test_pcm1 import bin_to_line_recs
from stitch_api import PAIR_DTYPE
from sdvpcmdecoder_amd import Engine, synth, Pcm16x0StitchSettings, Pcm1StitchSettings

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
max_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
orc = libs.load_oracle()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    fmt = ("pcm1", "si", "ei")[case % 3]
    n = int(rng.integers(6, max_frames))
    sigma = float(rng.choice([0.0, 3.0, 5.0]))
    mode = int(rng.choice([0, 1, 2]))
    a_mode = int(rng.integers(1, 7))
    if fmt == "pcm1":
        luma = synth.pcm1_frames(n, seed=seed0 + case, height=486, noise_sigma=sigma)[0].copy()
    else:
        luma = synth.pcm16x0_tape_frames(n, seed=seed0 + case, ei=(fmt == "ei"), noise_sigma=sigma)[0].copy()
    for f in rng.choice(np.arange(1, n), size=int(rng.integers(0, 4 + n // 10)), replace=False):      # lost lines
        luma[int(f), rng.integers(10, 470, size=int(rng.integers(1, 12)))] = 20
    t0 = time.time()
    kw = dict(new_file=True, end_file=True)
    if fmt == "pcm1":
        recs, stats = p1f.run_cpu(orc, "orc_", luma, mode, kw)
        st = p1.default_settings()
        want_p, want_f = p1.run_cpu(orc, "orc_", bin_to_line_recs(recs), st)
    else:
        recs, stats = p16f.run_cpu(orc, "orc_", luma, mode, kw)
        st = p16.default_settings(format=p16.FORMAT_EI if fmt == "ei" else p16.FORMAT_SI)
        want_p, want_f = p16.run_cpu(orc, "orc_", recs, st)
    w_audio = au.run_cpu(orc, "orc_", want_p, a_mode, np.array([len(want_p)], dtype=np.uint64), 1)
    t_cpu = time.time() - t0
    eng = Engine(0)
    eng.setBinarizationMode(mode)
    pcm_type = 0 if fmt == "pcm1" else 1
    eng.setPCMType(pcm_type)
    if fmt == "pcm1":
        eng.set_pcm1_stitch_settings(Pcm1StitchSettings.from_buffer_copy(bytes(st)))
    else:
        eng.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    d = torch.from_numpy(luma).cuda()
    gp, gf, gs = eng.decode_frames(pcm_type, d, first_frame_no=1, new_file=True, end_file=True)
    ok = gp.cpu().numpy().tobytes() == want_p.tobytes() and gf.cpu().numpy().tobytes() == want_f.tobytes() and gs.cpu().numpy().tobytes() == stats.tobytes()
    eng2 = Engine(0)
    eng2.setBinarizationMode(mode); eng2.setPCMType(pcm_type); eng2.set_audio_masking(a_mode)
    if fmt == "pcm1":
        eng2.set_pcm1_stitch_settings(Pcm1StitchSettings.from_buffer_copy(bytes(st)))
    else:
        eng2.set_pcm16x0_stitch_settings(Pcm16x0StitchSettings.from_buffer_copy(bytes(st)))
    ga, _, _, pur, masked = eng2.decode_frames(pcm_type, d, first_frame_no=1, new_file=True, end_file=True, with_audio=True, audio_stop=True)
    ok_a = w_audio[4] != 0 or (ga.cpu().numpy().tobytes() == w_audio[0].tobytes() and masked == w_audio[3])
    print(f"case {case}: {fmt} {n} frames sigma {sigma} mode {mode} audio mode {a_mode} pairs {len(want_p)} masked {w_audio[3]} cpu {t_cpu:.1f}s -> {'OK' if ok and ok_a else 'MISMATCH'}", flush=True)
    if not (ok and ok_a):
        sys.exit(1)
print("soak ok")
