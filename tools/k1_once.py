"""A few sdv_binarize_frames calls over the clean 10 000-frame NTSC tape of the benchmark through ONE build of the library (raw ctypes, like
tools/k1_ab.py): the program rocprofv3 profiles when builds are compared counter by counter.  usage: k1_once.py lib.so [calls]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sdvpcmdecoder_amd import synth
path = sys.argv[1]; calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, W, H = 10000, 720, 486
luma, _ = synth.stc007_frames_torch(n, seed=2, device="cuda", width=W, height=H, noise_sigma=4.0, cyclic=True)
recs = torch.empty((n * (H + 3) + 1, 48), dtype=torch.uint8, device="cuda")
stats = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
lib = C.CDLL(os.path.abspath(path))
lib.sdv_engine_create.restype = C.c_void_p; lib.sdv_engine_create.argtypes = [C.c_int]
lib.sdv_binarize_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint,
                                    C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
lib.sdv_set_mode.argtypes = [C.c_void_p, C.c_int]
e = C.c_void_p(lib.sdv_engine_create(0)); lib.sdv_set_mode(e, 2)
fno = 1
for i in range(calls + 1):
    flags = 1 if i == 0 else 0
    rc = lib.sdv_binarize_frames(e, luma.data_ptr(), W, W * H, W, H, n, fno, flags, recs.data_ptr() + (0 if flags & 1 else 48), recs.shape[0], stats.data_ptr(), n, None)
    assert rc == 0, rc
    fno += n
torch.cuda.synchronize()
print("ok", path)
