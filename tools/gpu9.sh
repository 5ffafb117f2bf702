cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python - <<'PY'
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import deint_api as da
from sdvpcmdecoder_amd import Engine, synth
rng=np.random.default_rng(1)
nb=4_900_000   # 10k NTSC frames worth of blocks
n=nb+112
audio=rng.integers(0,1<<14,size=(n,6),dtype=np.uint32)
w9=synth.interleave_stream(audio)
lines=da.make_lines(w9, rng=rng, p_bad=0.02, p_corrupt_valid=0.0)
eng=Engine(0); st=eng.default_deint_settings()
d=torch.from_numpy(lines.view(np.uint8).reshape(n,24)).to('cuda:0')
out=torch.empty((nb,72),dtype=torch.uint8,device='cuda:0')
for _ in range(3): eng.deinterleave_blocks(d, st, nb, out=out)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): eng.deinterleave_blocks(d, st, nb, out=out)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/10
bytes_alg = nb*(24+72)
print(f"deint kernel: {ms:.3f} ms per {nb} blocks = {nb/490/ms*1e3:.0f} frames/s-equivalent, {bytes_alg/ms/1e6:.0f} GB/s algorithmic (24 B line read + 72 B block write per block)")
PY
