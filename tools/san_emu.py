import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes as C, numpy as np
import libs, engine_api as ea, pcm1_api as p1, stitch_api as sa, stitch_cases as sc
from oracle_run import oracle_binarize
from sdvpcmdecoder_amd import synth
lib = ea.bind(C.CDLL(os.path.join(ROOT, 'build', 'san') + '/libsdvpcm_emu.so'))
orc = libs.load_oracle()
for name in list(p1.CASES):
    recs, st = p1.make_input(name)
    wp, wf = p1.run_cpu(orc, "orc_", recs, st)
    eng = lib.sdv_engine_create(0)
    rc, p, f = ea.emu_pcm1_stitch(lib, eng, recs, st)
    assert rc == 0 and p.tobytes() == wp.tobytes() and f.tobytes() == wf.tobytes(), name
    lib.sdv_engine_destroy(eng)
print("pcm1 emu ok")
for name in ("ntsc_bad5", "ntsc_burst300", "f1_16bit_bad5", "ntsc_drift"):
    recs, st = sc.make_input(name, lambda luma: oracle_binarize(luma, mode=2))
    wp, wf = sa.run_cpu(orc, "orc_", recs, st)
    eng = lib.sdv_engine_create(0)
    rc, p, f = ea.emu_stitch(lib, eng, recs, st)
    assert rc == 0 and p.tobytes() == wp.tobytes() and f.tobytes() == wf.tobytes(), name
    lib.sdv_engine_destroy(eng)
print("stitch emu ok")
luma, _, _ = synth.stc007_frames(3, seed=21, height=96, noise_sigma=8.0, blur=1)
luma = luma.copy(); luma[:, 33::17] = 16
want, wst = oracle_binarize(luma, mode=2, first_frame_no=1, new_file=True)
eng = lib.sdv_engine_create(0)
lib.sdv_set_mode(eng, 2)
rc, recs, stats = ea.emu_binarize(lib, eng, luma)
assert rc == 0 and recs.tobytes() == want.tobytes()
print("binarize emu ok")
# small rounds that settle their sweeps themselves (sdv_k_stc007_frames_fat), the sweeps' comparisons on bit planes
from test_gpu_parity import _unreadable_cells
luma = _unreadable_cells(synth.stc007_frames(6, seed=78, noise_sigma=4.0, height=120, lines_per_field=60)[0], every=23)
luma[:, 50::31, :] = 16
want, wst = oracle_binarize(np.ascontiguousarray(luma), mode=2)
eng = lib.sdv_engine_create(0)
lib.sdv_set_mode(eng, 2)
rc, recs, stats = ea.emu_binarize(lib, eng, luma)
assert rc == 0 and recs.tobytes() == want.tobytes()
lib.sdv_engine_destroy(eng)
print("binarize emu (sweeps in small rounds) ok")
import ctypes as C
import pcm1_front_api as pf
lib.sdv_engine_create.restype = C.c_void_p
for name in ("clean_fast", "cut_bits_draft", "noisy_header", "window_moves", "garbage", "forced_coords"):
    luma1, run = pf.make_case(name)
    seq, _, _ = pf.run_lines(orc, "orc_bin1_", luma1, **run)
    states = pf.states_from_records(seq)
    keep = np.ones(len(luma1), dtype=bool)
    want1 = pf.run_lines_with_states(orc, "orc_bin1_", luma1[keep], states[keep], mode=run["mode"], coord_search=run.get("coord_search", True), preset=run["preset"])
    eng = C.c_void_p(lib.sdv_engine_create(0))
    rc, got1 = pf.run_engine_lines(lib, eng, luma1[keep], states[keep], mode=run["mode"], coord_search=run.get("coord_search", True), preset=run["preset"])
    assert rc == 0 and got1.tobytes() == want1.tobytes(), name
    lib.sdv_engine_destroy(eng)
print("pcm1 front emu ok")

# round 2: the frame drivers (MODE_INSANE and the parallel walk included) and the PCM-16x0 back half on the emulator
import pcm1_frames_api as f1, pcm16_frames_api as f16, pcm16_api as p16
for pf in (f1, f16):
    for name in ("clean_normal", "cut_bits_normal", "dropouts_fast", "jitter_draft", "garbage", "insane_jitter", "insane_flag_matters"):
        luma2, mode, st = pf.make_input(name)
        want2, wst2 = pf.run_cpu(orc, "orc_", luma2, mode, st)
        eng = C.c_void_p(lib.sdv_engine_create(0))
        rc, got2, st2 = pf.run_engine(lib, eng, luma2, mode, st)
        assert rc == 0 and got2.tobytes() == want2.tobytes() and st2.tobytes() == wst2.tobytes(), name
        lib.sdv_engine_destroy(eng)
print("frame drivers emu ok")
for name in ("si_clean", "si_bad10", "si_picked_forced"):
    recs, st = p16.make_input(name)
    wp, wf = p16.run_cpu(orc, "orc_", recs, st)
    eng = C.c_void_p(lib.sdv_engine_create(0))
    rc, p, f = ea.emu_pcm16_stitch(lib, eng, recs, st)
    assert rc == 0 and p.tobytes() == wp.tobytes() and f.tobytes() == wf.tobytes(), name
    lib.sdv_engine_destroy(eng)
print("pcm16x0 stitch emu ok")
# round 2, SURVEY 8f: AudioProcessor on the emulator (plan with leaps, window chains, emit, WAV packing)
import audio_api as au
au.bind_product(lib)
for name in sorted(au.CASES):
    pairs, mode, ends, stop = au.make_input(name)
    want = au.run_cpu(orc, "orc_", pairs, mode, ends, stop)
    out, pur, masked = au.emu_run(lib, pairs, mode, ends, stop)
    assert out.tobytes() == want[0].tobytes() and pur.tobytes() == want[2].tobytes() and masked == want[3], name
    au.wav_files(lib, "sdv_", out, pur)
print("audio emu ok")
