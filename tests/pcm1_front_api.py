"""PCM-1 front half (Binarizer::processLine with a PCM1Line output): record dtype and the runner shared by the oracle and the
reference driver (same C signatures under the prefixes orc_bin1_ / ref_bin1_)."""
import ctypes as C

import numpy as np

import libs

BIN1_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (7,)), ("calc_crc", "<u2"),
                       ("data_start", "<i2"), ("data_stop", "<i2"),
                       ("black_level", "u1"), ("white_level", "u1"), ("ref_low", "u1"), ("ref_level", "u1"), ("ref_high", "u1"),
                       ("hysteresis_depth", "u1"), ("shift_stage", "u1"), ("service_type", "u1"),
                       ("picked_bits_left", "u1"), ("picked_bits_right", "u1"), ("flags", "u1"), ("_pad", "u1", (3,))])
assert BIN1_DTYPE.itemsize == 40
LF_COORDS_SWEEPED, LF_BY_EXT_TUNE, LF_BW_SET, LF_COORDS_SET, LF_FORCED_BAD, LF_CRC_VALID = 2, 4, 8, 16, 32, 64


def run_lines(lib, prefix, luma, mode=1, coord_search=True, preset=None, feedback="good", services=None, doubled=False, empty=None,
              first_line=1, frame=1):
    """One Binarizer over the rows of `luma`.  feedback: "good" = setGoodParameters(last line) after every line (what the frame
    driver does for a line with a valid CRC), "none" = every line from scratch, "reset" = setGoodParameters(NULL) before each."""
    f = lambda name: getattr(lib, prefix + name)
    f("new").restype = C.c_void_p
    h = C.c_void_p(f("new")())
    f("set_mode").argtypes = [C.c_void_p, C.c_int]
    f("set_coord_search").argtypes = [C.c_void_p, C.c_int]
    f("set_preset").argtypes = [C.c_void_p, C.c_void_p]
    f("reset_good").argtypes = [C.c_void_p]
    f("set_good_from_last").argtypes = [C.c_void_p]
    f("scan_done").argtypes = [C.c_void_p]
    f("free").argtypes = [C.c_void_p]
    proc = f("process")
    proc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint16, C.c_int, C.c_int, C.c_int, C.c_void_p]
    f("set_mode")(h, mode)
    f("set_coord_search")(h, 1 if coord_search else 0)
    if preset is not None:
        f("set_preset")(h, C.byref(preset))
    luma = np.ascontiguousarray(luma)
    n = luma.shape[0]
    out = np.zeros(n, dtype=BIN1_DTYPE)
    rets = np.zeros(n, dtype=np.int32)
    scans = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        if feedback == "reset":
            f("reset_good")(h)
        srv = 0 if services is None else int(services[i])
        emp = 0 if empty is None else int(empty[i])
        rets[i] = proc(h, luma[i].ctypes.data, luma.shape[1], frame, first_line + i, srv, 1 if doubled else 0, emp, out[i:i + 1].ctypes.data)
        scans[i] = f("scan_done")(h)
        if feedback == "good":
            f("set_good_from_last")(h)
    f("free")(h)
    return out, rets, scans


# ---- seeded scenarios: name -> (generator kwargs, runner kwargs, preset overrides) -------------------------------------------------
def _preset(**kw):
    p = libs.default_preset()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


CASES = {
    "clean_fast": (dict(n=16, seed=401), dict(mode=1, feedback="good"), {}),
    "clean_scratch_normal": (dict(n=8, seed=402), dict(mode=2, feedback="none"), {}),
    "cut_bits_draft": (dict(n=12, seed=403, x0=-9, x1=726, noise_sigma=3.0), dict(mode=0, feedback="good"), {}),
    "cut_bits_normal": (dict(n=12, seed=404, x0=-14, x1=731, noise_sigma=2.0), dict(mode=2, feedback="good"), {}),
    "cut_left_only": (dict(n=10, seed=405, x0=-12, x1=700, noise_sigma=2.0), dict(mode=1, feedback="reset"), {}),
    "cut_right_only": (dict(n=10, seed=406, x0=10, x1=728, noise_sigma=2.0), dict(mode=1, feedback="reset"), {}),
    "no_bit_picker": (dict(n=8, seed=407, x0=-9, x1=726), dict(mode=1, feedback="good"), dict(left_bit_pick=0, right_bit_pick=0)),
    "noisy_header": (dict(n=14, seed=408, x0=7, x1=709, noise_sigma=8.0, blur=1, header_every=5), dict(mode=1, feedback="good"), {}),
    "heavy_noise": (dict(n=12, seed=409, x0=5, x1=713, noise_sigma=22.0, blur=2), dict(mode=2, feedback="good"), {}),
    "low_contrast": (dict(n=10, seed=410, black=60, white=95, noise_sigma=2.0), dict(mode=1, feedback="good"), {}),
    "dark": (dict(n=8, seed=411, black=16, white=40), dict(mode=1, feedback="good"), {}),
    "wide_1440_doubled": (dict(n=6, seed=412, width=1440, x0=9, x1=1431, noise_sigma=3.0), dict(mode=1, feedback="good", doubled=True), {}),
    "narrow_640": (dict(n=8, seed=413, width=640, x0=3, x1=636, noise_sigma=3.0), dict(mode=1, feedback="good"), {}),
    "no_coord_search": (dict(n=8, seed=414), dict(mode=1, feedback="good", coord_search=False), {}),
    "search_disabled_in_preset": (dict(n=8, seed=415), dict(mode=1, feedback="good"), dict(en_coord_search=0)),
    "forced_coords": (dict(n=8, seed=416, x0=6, x1=712), dict(mode=1, feedback="good"), dict(en_force_coords=1, horiz_start=6, horiz_stop=8)),
    "forced_coords_wrong": (dict(n=6, seed=417, x0=6, x1=712), dict(mode=1, feedback="good"), dict(en_force_coords=1, horiz_start=30, horiz_stop=2)),
    "garbage": (dict(n=10, seed=418, garbage=True), dict(mode=1, feedback="good"), {}),
    "flat_and_services": (dict(n=12, seed=419, flat=(2, 7), noise_sigma=2.0), dict(mode=1, feedback="good", services={4: 4, 5: 5, 9: 3}, empty=(10,)), {}),
    "window_moves": (dict(n=16, seed=420, noise_sigma=3.0, jump_at=8, jump_to=(15, 700)), dict(mode=1, feedback="good"), {}),
    # MODE_INSANE: every line that does not read from what was handed on runs the reference level sweep (a coordinate search per level)
    "insane_scratch": (dict(n=3, seed=431, noise_sigma=2.0, black=40, white=120), dict(mode=3, feedback="none"), {}),
    "insane_cut_right": (dict(n=4, seed=432, x0=10, x1=728, noise_sigma=2.0), dict(mode=3, feedback="reset"), {}),
    "insane_cut_left": (dict(n=4, seed=433, x0=-12, x1=700, noise_sigma=2.0), dict(mode=3, feedback="reset"), {}),
    "insane_noisy_header": (dict(n=12, seed=434, x0=7, x1=709, noise_sigma=8.0, blur=1, header_every=5), dict(mode=3, feedback="good"), {}),
    "insane_low_contrast": (dict(n=6, seed=435, black=60, white=95, noise_sigma=2.0), dict(mode=3, feedback="none"), {}),
    "insane_level_range": (dict(n=6, seed=436, noise_sigma=4.0), dict(mode=3, feedback="none"), dict(min_ref_lvl=96, max_ref_lvl=112)),
    "insane_forced_coords": (dict(n=5, seed=437, x0=6, x1=712), dict(mode=3, feedback="none"), dict(en_force_coords=1, horiz_start=6, horiz_stop=8)),
    "insane_garbage": (dict(n=4, seed=438, garbage=True), dict(mode=3, feedback="good"), {}),
    "insane_window_moves": (dict(n=10, seed=439, noise_sigma=3.0, jump_at=5, jump_to=(15, 700)), dict(mode=3, feedback="good"), {}),
    "insane_few_valid": (dict(n=3, seed=440, black=50, white=100, noise_sigma=2.0), dict(mode=3, feedback="none"), dict(min_valid_crcs=60)),
}
GOLDEN = ("clean_fast", "cut_bits_normal", "noisy_header", "heavy_noise", "forced_coords", "flat_and_services", "window_moves",
          "insane_scratch", "insane_cut_left", "insane_noisy_header", "insane_few_valid")


def make_case(name):
    from sdvpcmdecoder_amd import synth
    gen, run, pre = CASES[name]
    gen = dict(gen)
    n = gen.pop("n"); seed = gen.pop("seed")
    garbage = gen.pop("garbage", False); flat = gen.pop("flat", ()); jump_at = gen.pop("jump_at", None); jump_to = gen.pop("jump_to", None)
    luma, words = synth.pcm1_random_lines(n, seed=seed, **gen)
    rng = np.random.default_rng(seed + 1000)
    if garbage:
        luma = rng.integers(0, 256, size=luma.shape).astype(np.uint8)
    for i in flat:
        luma[i] = 40
    if jump_at is not None:
        gen2 = dict(gen); gen2["x0"], gen2["x1"] = jump_to
        luma2, _ = synth.pcm1_random_lines(n, seed=seed, **gen2)
        luma[jump_at:] = luma2[jump_at:]
    run = dict(run)
    srv = run.pop("services", None)
    if srv is not None:
        a = np.zeros(n, dtype=np.int32)
        for k, v in srv.items():
            a[k] = v
        run["services"] = a
    emp = run.pop("empty", None)
    if emp is not None:
        e = np.zeros(n, dtype=np.int32); e[list(emp)] = 1
        run["empty"] = e
    run["preset"] = _preset(**pre)
    return np.ascontiguousarray(luma), run


def run_engine_lines(lib, eng, luma, states=None, mode=1, coord_search=True, preset=None, doubled=False, first_line=1, frame=1, line_step=1):
    """sdv_pcm1_binarize_lines on host buffers (the emulator build); one launch for all rows, row i preset with states[i]."""
    f = lib.sdv_pcm1_binarize_lines
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint, C.c_int,
                  C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_set_mode.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_set_mode(eng, mode)
    lib.sdv_set_bin_preset.argtypes = [C.c_void_p, C.POINTER(libs.BinPreset)]
    p = preset if preset is not None else libs.default_preset()
    lib.sdv_set_bin_preset(eng, C.byref(p))
    luma = np.ascontiguousarray(luma)
    out = np.zeros(luma.shape[0], dtype=BIN1_DTYPE)
    st = None if states is None else np.ascontiguousarray(states)
    rc = f(eng, luma.ctypes.data, luma.shape[1], luma.shape[1], luma.shape[0], None if st is None else st.ctypes.data, frame, first_line, line_step,
           2 if doubled else 0, 1 if coord_search else 0, out.ctypes.data, len(out), None)
    return rc, out


STATE_DTYPE = np.dtype([("black", "u1"), ("white", "u1"), ("ref", "u1"), ("sweep_flag", "u1"), ("start", "<i2"), ("stop", "<i2"), ("doubled", "u1"), ("_p2", "u1")])
assert STATE_DTYPE.itemsize == 10


def run_lines_with_states(lib, prefix, luma, states, mode=1, coord_search=True, preset=None, doubled=False, first_line=1, frame=1, line_step=1):
    """The oracle / the reference, every line on a Binarizer preset with states[i] (set_state) - the per-line contract of the engine entry."""
    f = lambda name: getattr(lib, prefix + name)
    f("new").restype = C.c_void_p
    h = C.c_void_p(f("new")())
    f("set_mode").argtypes = [C.c_void_p, C.c_int]
    f("set_coord_search").argtypes = [C.c_void_p, C.c_int]
    f("set_preset").argtypes = [C.c_void_p, C.c_void_p]
    f("set_state").argtypes = [C.c_void_p, C.c_void_p]
    f("free").argtypes = [C.c_void_p]
    proc = f("process")
    proc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint16, C.c_int, C.c_int, C.c_int, C.c_void_p]
    f("set_mode")(h, mode)
    f("set_coord_search")(h, 1 if coord_search else 0)
    if preset is not None:
        f("set_preset")(h, C.byref(preset))
    luma = np.ascontiguousarray(luma)
    states = np.ascontiguousarray(states)
    out = np.zeros(luma.shape[0], dtype=BIN1_DTYPE)
    for i in range(luma.shape[0]):
        f("set_state")(h, states[i:i + 1].ctypes.data)
        proc(h, luma[i].ctypes.data, luma.shape[1], frame, first_line + i * line_step, 0, 1 if doubled else 0, 0, out[i:i + 1].ctypes.data)
    f("free")(h)
    return out


def states_from_records(recs, mode=1):
    """What setGoodParameters(previous line) leaves in the Binarizer before each line (binarizer.cpp:353-377): the levels and
    coordinates of the last line with a valid CRC (ignoring the forced-bad mark), nothing before the first one.  And the sticky
    do_ref_lvl_sweep member: a line that found levels and was not read with the preset tuning went through :1104 and left it at
    "the mode is MODE_INSANE"."""
    st = np.zeros(len(recs), dtype=STATE_DTYPE)
    cur = np.zeros(1, dtype=STATE_DTYPE)[0]
    cur["start"], cur["stop"] = -32768, 32767
    for i in range(len(recs)):
        st[i] = cur
        r = recs[i]
        w = r["words"]
        if (int(r["flags"]) & LF_BW_SET) and not (int(r["flags"]) & LF_BY_EXT_TUNE):
            cur = cur.copy()
            cur["sweep_flag"] = 1 if mode == 3 else 0
        hdr = tuple(int(x) for x in w) == (0x0666, 0x0CCC, 0x1999, 0x1333, 0x0666, 0x0CCC, 0xCCCC)
        if int(r["calc_crc"]) == int(w[6]) or hdr:
            cur = cur.copy()
            cur["ref"] = r["ref_level"]
            s, e = int(r["data_start"]), int(r["data_stop"])
            if s != -32768 and e != 32767 and s < e:
                cur["start"], cur["stop"], cur["doubled"] = s, e, 1 if (int(r["flags"]) & 128) else 0
            else:
                cur["start"], cur["stop"], cur["doubled"] = -32768, 32767, 0
            b, wht = int(r["black_level"]), int(r["white_level"])
            if b < wht and b < 160 and wht > 28 and wht != 0:
                cur["black"], cur["white"] = b, wht
            else:
                cur["black"], cur["white"] = 0, 0
    return st


def states_for_run(recs, run):
    """Per-line presets equivalent to the sequential run `run` that produced `recs`: everything for feedback "good", only the
    sticky sweep flag (which no feedback mode touches) for "none" / "reset"."""
    st = states_from_records(recs, run["mode"])
    if run.get("feedback") != "good":
        st["black"] = st["white"] = st["ref"] = 0
        st["start"], st["stop"], st["doubled"] = -32768, 32767, 0
    return st
