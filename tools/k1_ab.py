"""K1 A/B across builds of the library on ONE box: the clean 10 000-frame NTSC tape of the benchmark through sdv_binarize_frames of each library
given on the command line, alternating, `reps` rounds of `steps` timed calls each (hipEvents around the calls on the null stream).  Raw ctypes: only
entry points every ABI version has.  usage: k1_ab.py reps lib1.so lib2.so ..."""
import ctypes as C, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import synth
reps = int(sys.argv[1]); paths = sys.argv[2:]
n, W, H = 10000, 720, 486
luma, _ = synth.stc007_frames_torch(n, seed=2, device="cuda", width=W, height=H, noise_sigma=4.0, cyclic=True)
recs = torch.empty((n * (H + 3) + 1, 48), dtype=torch.uint8, device="cuda")
stats = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
engines = []
for p in paths:
    lib = C.CDLL(os.path.abspath(p))
    lib.sdv_engine_create.restype = C.c_void_p; lib.sdv_engine_create.argtypes = [C.c_int]
    lib.sdv_binarize_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint,
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_set_mode.argtypes = [C.c_void_p, C.c_int]
    e = C.c_void_p(lib.sdv_engine_create(0)); lib.sdv_set_mode(e, 2)
    def call(first, flags, lib=lib, e=e):
        rc = lib.sdv_binarize_frames(e, luma.data_ptr(), W, W * H, W, H, n, first, flags, recs.data_ptr() + (0 if flags & 1 else 48), recs.shape[0], stats.data_ptr(), n, None)
        assert rc == 0, rc
    call(1, 1); call(1 + n, 0); call(1 + 2 * n, 0)
    engines.append((p, call))
steps = 10
times = {p: [] for p in paths}
fno = 1 + 3 * n
for r in range(reps):
    for p, call in engines:
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            call(fno, 0); fno += n
        e1.record(); torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1) / steps)
for p in paths:
    t = times[p]
    print(f"{os.path.basename(p):24s} ms per 10 000-frame call: median {statistics.median(t):.4f}  min {min(t):.4f}  max {max(t):.4f}   runs " + " ".join(f"{x:.3f}" for x in t))
