"""Does the lean kernel's launch time follow the buffer (where the frames lie) or the process?  One tape, cloned into fresh allocations; per clone the average
of 20 launches (engine's event pair)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import Engine, synth
n = 10000
dev = "cuda"
luma0, _ = synth.stc007_frames_torch(n, seed=2, device=dev, noise_sigma=4.0, cyclic=True)
eng = Engine(0); eng.set_profiling(True); eng.setBinarizationMode(2)
out_lines = torch.empty((n * 489 + 1, 48), dtype=torch.uint8, device=dev)
out_stats = torch.empty((n, 32), dtype=torch.uint8, device=dev)
keep = []
def run(luma, tag):
    eng.reset_stream()
    eng.binarize_frames(luma, first_frame_no=1, new_file=True, out_lines=out_lines, out_stats=out_stats)
    fno = 1 + n
    for _ in range(3):
        eng.binarize_frames(luma, first_frame_no=fno, out_lines=out_lines[1:], out_stats=out_stats); fno += n
    k = 0.0
    for _ in range(20):
        eng.binarize_frames(luma, first_frame_no=fno, out_lines=out_lines[1:], out_stats=out_stats); fno += n
        k += eng.run_info().kernel_ms
    print(f"{tag}: ptr {luma.data_ptr():#x} launch {k / 20:.4f} ms", flush=True)
run(luma0, "original")
for i in range(5):
    pad = torch.empty(((i + 1) * 37 * 4096 + 512 * i,), dtype=torch.uint8, device=dev); keep.append(pad)
    c = luma0.clone(); keep.append(c)
    run(c, f"clone {i}")
run(luma0, "original again")
