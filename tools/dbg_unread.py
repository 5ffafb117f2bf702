import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_cases
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import Engine, LINE_DTYPE, synth
from test_gpu_parity import _unreadable_cells
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
luma, _, _ = synth.stc007_frames(40, seed=77, noise_sigma=4.0)
luma = np.ascontiguousarray(_unreadable_cells(luma)[:n])
want, ws = oracle_binarize(luma, mode=2)
eng = Engine(0); eng.setBinarizationMode(2)
lines, stats = eng.binarize_frames(torch.from_numpy(luma).to("cuda:0"), first_frame_no=1, new_file=True)
torch.cuda.synchronize()
i = eng.run_info()
got = lines.cpu().numpy().view(LINE_DTYPE).reshape(-1)
print("rounds", i.rounds, "sweeps", i.sweeps, "equal", got.tobytes() == want.tobytes())
if got.tobytes() != want.tobytes(): print(golden_cases.diff_report(got, want, limit=3))
