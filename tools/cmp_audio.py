#!/usr/bin/env python3
"""Developer aid: oracle/audio.c against the real AudioProcessor (oracle/_ref) on every case of tests/audio_api.py."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, libs, audio_api as A
orc = libs.load_oracle(); ref = libs.load_ref()
bad = 0
names = sys.argv[1:] or list(A.CASES)
for name in names:
    pairs, mode, ends, stop = A.make_input(name)
    t = time.time()
    o = A.run_cpu(orc, "orc_", pairs, mode, ends, stop)
    r = A.run_cpu(ref, "ref_", pairs, mode, ends, stop)
    ok = o[0].tobytes() == r[0].tobytes() and np.array_equal(o[1], r[1]) and np.array_equal(o[2]["first_pair"], r[2]) and o[3] == r[3]
    ok_idx = np.array_equal(o[1], A.expected_index(len(o[0]), o[2]["first_pair"]))
    print(name, "OK" if ok else "DIFF", len(o[0]), len(r[0]), "purges", list(o[2]["first_pair"]), list(r[2]), "masked", o[3], r[3], "unsup", o[4], "idx", ok_idx, "%.1fs" % (time.time() - t))
    if not ok:
        bad += 1
        n = min(len(o[0]), len(r[0]))
        d = np.nonzero((o[0][:n] != r[0][:n]))[0]
        print("   first diffs", d[:10], o[0][d[:3]], r[0][d[:3]])
print("bad", bad)
