"""PCM-1 front half on the GPU (sdv_pcm1_binarize_lines): lines per second with every line preset from a decoded neighbour (a tape
that plays) and with nothing preset (the marker-less coordinate search on every line).  Usage: pcm1_front_prof.py [frames] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
LPF = 490                                            # PCM-1 lines of an NTSC frame (245 per field)
base, words = synth.pcm1_random_lines(LPF * 8, seed=21, x0=5, x1=713, noise_sigma=4.0)
d_base = torch.from_numpy(base).cuda()
n = frames * LPF
luma = d_base.repeat((n + len(base) - 1) // len(base), 1)[:n].contiguous()
# every line its own pixels (the tile would sit in the caches): +-3 of noise on top, far inside the decision levels
g = torch.Generator(device="cuda"); g.manual_seed(5)
for i in range(0, n, 1 << 18):
    blk = luma[i:i + (1 << 18)]
    blk.copy_((blk.to(torch.int16) + torch.randint(-3, 4, blk.shape, generator=g, device="cuda", dtype=torch.int16)).clamp_(0, 255).to(torch.uint8))
eng = Engine(0); eng.setBinarizationMode(2)
cold_n = min(n, 20 * LPF)
t0 = time.perf_counter(); recs = eng.pcm1_binarize_lines(luma[:cold_n]); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
r = recs.cpu().numpy().reshape(-1).view(np.dtype([("frame", "<u4"), ("line", "<u2"), ("words", "<u2", (7,)), ("crc", "<u2"), ("start", "<i2"), ("stop", "<i2"),
                                                  ("lv", "u1", (5,)), ("hs", "u1", (2,)), ("srv", "u1"), ("pk", "u1", (2,)), ("flags", "u1"), ("_p", "u1", (3,))]))
ok = (r["flags"] & 64) != 0
print(f"cold: {cold_n} lines, {int(ok.sum())} with a valid CRC, words equal the generator's: {bool((r['words'] == words[np.arange(cold_n) % len(base)]).all())}", flush=True)
st = np.zeros(n, dtype=np.dtype([("black", "u1"), ("white", "u1"), ("ref", "u1"), ("_p", "u1"), ("start", "<i2"), ("stop", "<i2"), ("d", "u1"), ("_p2", "u1")]))
k = int(np.flatnonzero(ok)[0])
st["black"], st["white"], st["ref"], st["start"], st["stop"] = r["lv"][k][0], r["lv"][k][1], r["lv"][k][3], r["start"][k], r["stop"][k]
d_st = torch.from_numpy(st.view(np.uint8).reshape(n, 10)).cuda()
out = torch.empty((n, 40), dtype=torch.uint8, device="cuda")
for name, args, cnt in (("warm", (luma, d_st), n), ("cold", (luma[:cold_n], None), cold_n)):
    eng.pcm1_binarize_lines(*args, out_lines=out[:cnt]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.pcm1_binarize_lines(*args, out_lines=out[:cnt])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    h = out[:cnt].cpu().numpy().reshape(-1).view(r.dtype)
    print(f"{name}: {cnt} lines in {dt * 1e3:.3f} ms = {cnt / dt / 1e6:.2f} M lines/s = {cnt / dt / LPF / 1e3:.1f} k frames/s, {cnt * 720 / dt / 1e9:.1f} GB/s of luma; "
          f"valid CRC {int(((h['flags'] & 64) != 0).sum())}, by preset {int(((h['flags'] & 4) != 0).sum())}", flush=True)
