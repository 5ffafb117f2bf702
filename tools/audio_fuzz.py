#!/usr/bin/env python3
"""Differential fuzz of the audio stage: random tapes (files of random length, dropout runs of random place / length / channel, random
invalid words and blocks, tags in random order, random bursts, every masking mode) through the kernels on the SIMT emulator against the
oracle - and, with `ref` as second argument, the oracle against the real AudioProcessor as well.   tools/audio_fuzz.py [n_cases] [ref]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes as C
import numpy as np
import audio_api as A, engine_api as ea, libs
from sdvpcmdecoder_amd import build as b


def random_case(seed):
    rng = np.random.default_rng(seed)
    parts = []
    if rng.random() < 0.85:
        parts.append("N")
    for f in range(int(rng.integers(1, 4))):
        n = int(rng.choice([rng.integers(3, 40), rng.integers(200, 260), rng.integers(500, 530), rng.integers(900, 4000), rng.integers(4000, 12000)]))
        runs = []
        for _ in range(int(rng.integers(0, 8))):
            ln = int(rng.choice([rng.integers(1, 4), rng.integers(4, 40), rng.integers(180, 260), rng.integers(260, 1500)]))
            runs.append((int(rng.integers(0, max(1, n))), ln, int(rng.integers(0, 3))))
        if rng.random() < 0.3:      # around the window grid
            k = int(rng.integers(1, max(2, n // 509 + 1)))
            runs.append((max(0, 509 * k - int(rng.integers(0, 6))), int(rng.integers(1, 300)), int(rng.integers(0, 3))))
        a = A.audio(n, seed * 31 + f, runs=runs, p_bad=float(rng.choice([0, 0, 0.002, 0.02, 0.2])), p_block=float(rng.choice([0, 0, 0.03])),
                    rate=int(rng.choice([44056, 44100])), tone=bool(rng.random() < 0.7))
        if rng.random() < 0.1:
            a["sample_flags"][rng.random((n, 2)) < 0.01] |= A.SF_WORD_MASKED       # foreign input: masked flags, valid or not
        parts.append(a)
        r = rng.random()
        parts += ["E", "N"] if r < 0.6 else ["N"] if r < 0.75 else ["E"] if r < 0.9 else []
    pairs = A.tape(parts)
    n = len(pairs)
    cuts = sorted(set(int(x) for x in rng.integers(1, n + 1, int(rng.integers(0, 5)))) | {n})
    return pairs, int(rng.integers(0, 7)), np.array(cuts, dtype=np.uint64), int(rng.integers(0, 2))


def check(emu, orc, seed, ref=None):
    pairs, mode, ends, stop = random_case(seed)
    want = A.run_cpu(orc, "orc_", pairs, mode, ends, stop)
    if ref is not None:
        r = A.run_cpu(ref, "ref_", pairs, mode, ends, stop)
        assert want[0].tobytes() == r[0].tobytes() and np.array_equal(want[1], r[1]) and np.array_equal(want[2]["first_pair"], r[2]) and want[3] == r[3], ("oracle vs reference", seed)
    if want[4]:
        return "refused"        # something the product refuses: the emulator run would return SDV_ERR_UNSUPPORTED
    out, pur, masked = A.emu_run(emu, pairs, mode, ends, stop)
    assert out.tobytes() == want[0].tobytes() and pur.tobytes() == want[2].tobytes() and masked == want[3], ("emulator vs oracle", seed, mode, list(ends), stop)
    return "ok"


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ref = libs.load_ref() if len(sys.argv) > 2 and sys.argv[2] == "ref" else None
    emu = A.bind_product(ea.bind(C.CDLL(b.build_emu())))
    orc = libs.load_oracle()
    res = {}
    for seed in range(1000, 1000 + n):
        r = check(emu, orc, seed, ref)
        res[r] = res.get(r, 0) + 1
    print(res)
