#!/usr/bin/env python3
"""Dev tool: run the oracle and the real reference per-line binarizer on the same synthetic lines."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from sdvpcmdecoder_amd import synth
import libs

def run(n, seed, mode, chain, **kw):
    luma, w9 = synth.random_lines(n, seed=seed, **kw)
    o = libs.BinApi(libs.load_oracle(), "orc_"); r = libs.BinApi(libs.load_ref(), "ref_")
    o.set_mode(mode); r.set_mode(mode)
    bad = 0; ok = 0; t_o = t_r = 0.0
    for i in range(n):
        t0 = time.perf_counter(); ro = o.process_px(luma[i], 1, i + 1); t1 = time.perf_counter()
        rr = r.process_px(luma[i], 1, i + 1); t2 = time.perf_counter()
        t_o += t1 - t0; t_r += t2 - t1
        a, b = libs.rec_tuple(ro[1]), libs.rec_tuple(rr[1])
        if ro[0] != rr[0] or a != b:
            bad += 1
            if bad <= 5:
                print("MISMATCH line", i, "\n orc", ro[0], a, "\n ref", rr[0], b)
        if rr[1].flags & 64: ok += 1
        if chain:
            if rr[1].flags & 64:
                o.set_good_from_last(); r.set_good_from_last()
        else:
            o.reset_good(); r.reset_good()
    print(f"mode={mode} chain={chain} kw={kw}: n={n} crc_ok={ok} mismatches={bad}  t_orc={t_o/n*1e6:.1f}us t_ref={t_r/n*1e6:.1f}us")
    return bad

if __name__ == "__main__":
    tot = 0
    for mode in (0, 1, 2, 3):
        for chain in (0, 1):
            tot += run(40, 1, mode, chain)
            tot += run(40, 2, mode, chain, noise_sigma=12.0, blur=2)
            tot += run(40, 3, mode, chain, noise_sigma=30.0, blur=3, shift=np.random.default_rng(5).integers(-3, 4, 40))
    print("TOTAL mismatches", tot)
