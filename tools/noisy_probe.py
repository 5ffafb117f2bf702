"""How the scheduler behaves on tapes the steady-state model predicts badly: heavy noise (every few lines need the general path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for sigma in [float(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["4", "12", "18"])]:
    luma, _ = synth.stc007_frames_torch(n, seed=3, device='cuda', noise_sigma=sigma, cyclic=True)
    eng = Engine(0); eng.set_profiling(True)
    eng.binarize_frames(luma, first_frame_no=1, new_file=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.binarize_frames(luma, first_frame_no=1 + n, new_file=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    i = eng.run_info()
    print(f"sigma {sigma}: wall {dt:.1f} ms, kernel {i.kernel_ms:.1f} ms, rounds {i.rounds}, frame decodes {i.frames_launched} ({i.frames_general} by the full kernel) for {n} frames", flush=True)
