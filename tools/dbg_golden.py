import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_cases
from sdvpcmdecoder_amd import Engine, LINE_DTYPE
name = sys.argv[1] if len(sys.argv) > 1 else "ntsc_clean_normal"
mode, luma, want, want_stats = golden_cases.load(name)
eng = Engine(0); eng.setBinarizationMode(mode)
d = torch.from_numpy(np.ascontiguousarray(luma)).to("cuda:0")
lines, stats = eng.binarize_frames(d, first_frame_no=1, new_file=True)
torch.cuda.synchronize()
i = eng.run_info()
print("rounds", i.rounds, "launched", i.frames_launched, "general", i.frames_general, "sweeps", i.sweeps)
got = lines.cpu().numpy().view(LINE_DTYPE).reshape(-1)
a8 = got.view(np.uint8).reshape(len(got), -1); b8 = want.view(np.uint8).reshape(len(want), -1)
bad = np.unique(np.nonzero(a8 != b8)[0])
print(len(bad), "differ; first", bad[:5], "last", bad[-5:] if len(bad) else None)
for k in list(range(0, 4)):
    print(k, "got ", got[k]); print(k, "want", want[k])
