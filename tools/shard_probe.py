"""What a warm-up predicts against what the predecessor really hands over (binarize chain state, 120 B; stitch state blob), on the benchmark's tape.
usage: shard_probe.py [frames per rank] [noise sigma] [warm-up frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sdvpcmdecoder_amd import Engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 20
luma, _ = synth.stc007_frames_torch(2 * n, seed=3, device="cuda", noise_sigma=sigma, cyclic=True)
eng = Engine(0); eng.setBinarizationMode(2)
eng.reset_stream()
eng.binarize_frames(luma[:n], first_frame_no=1, new_file=True)
real = np.frombuffer(eng.get_chain_state(), dtype=np.uint8)
eng.reset_stream()
eng.binarize_frames(luma[n - warm:n], first_frame_no=1 + n - warm, new_file=False)
pred = np.frombuffer(eng.get_chain_state(), dtype=np.uint8)
d = np.nonzero(real != pred)[0]
print("binarize: bytes that differ", d.tolist())
print(" real", real.view(np.uint32)[sorted(set(d // 4))].tolist() if len(d) else "", "\n pred", pred.view(np.uint32)[sorted(set(d // 4))].tolist() if len(d) else "")
print(" real state:", real[:12].tolist(), "n_last/n_long", real[10:12].tolist())
print(" pred state:", pred[:12].tolist())
# the protocol of ShardedDecoder: rank 0's presets after its first `warm` frames, then the warm-up from a reset worker with those presets
eng.reset_stream()
eng.binarize_frames(luma[:warm], first_frame_no=1, new_file=True)
early = eng.get_chain_state()
eng.reset_stream()
eng.set_chain_state(early[:10] + eng.get_chain_state()[10:])
eng.binarize_frames(luma[n - warm:n], first_frame_no=1 + n - warm, new_file=False)
pred2 = np.frombuffer(eng.get_chain_state(), dtype=np.uint8)
d2 = np.nonzero(real != pred2)[0]
print("with rank 0's early presets: bytes that differ", d2.tolist())
print(" early:", list(early[:12]), "\n pred :", pred2[:12].tolist(), "\n real :", real[:12].tolist())
eng.reset_stream(); eng.set_chain_state(early)
eng.binarize_frames(luma[n - warm:n], first_frame_no=1 + n - warm, new_file=False)
pred3 = np.frombuffer(eng.get_chain_state(), dtype=np.uint8)
print("with rank 0's whole early state: bytes that differ", np.nonzero(real != pred3)[0].tolist(), pred3[:12].tolist())
