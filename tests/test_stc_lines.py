"""sdv_binarize_lines: Binarizer::processLine with an STC007Line as output, a line at a time (binarizer.h:361; the per-line contract the reference's
VideoToDigital itself binds: setSource / setOutput / processLine, videotodigital.cpp:834-1003).  The oracle's per-line driver against the real
Binarizer (oracle/_ref, in this container), the kernel under the SIMT emulator and - gpu-marked - the HIP library through the C-ABI against the oracle."""
import ctypes as C

import numpy as np
import pytest

import libs
from pcm1_front_api import STATE_DTYPE
from sdvpcmdecoder_amd import synth

LF_BY_EXT_TUNE, LF_BW_SET, LF_CRC_VALID = 4, 8, 64


def _lines(seed=5, noise=4.0, width=720):
    """Lines of a tape that plays, lines beside their coordinates, lines with an unreadable cell, lost lines, black lines with a white spot."""
    luma, _, _ = synth.stc007_frames(2, seed=seed, height=48, noise_sigma=noise, width=width)
    rows = luma.reshape(-1, width).copy()
    rng = np.random.default_rng(seed)
    for i in range(7, len(rows), 9):
        rows[i] = np.roll(rows[i], int(rng.integers(-3, 4)))
    for i in range(4, len(rows), 11):
        x = 12 + int(rng.integers(4, 132)) * (width - 24) // 137
        rows[i, x:x + 5] = np.clip(230 - rows[i, x:x + 5].astype(np.int16), 0, 255).astype(np.uint8)
    rows[13] = 16; rows[40] = 16; rows[41, 300:330] = 220
    return np.ascontiguousarray(rows)


def _bind(lib, prefix):
    f = lambda name: getattr(lib, prefix + name)
    f("new").restype = C.c_void_p
    f("set_mode").argtypes = [C.c_void_p, C.c_int]
    f("set_state").argtypes = [C.c_void_p, C.c_void_p]
    f("free").argtypes = [C.c_void_p]
    f("process").argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint16, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return f


def _per_line(lib, prefix, rows, states, mode, doubled=False, first_line=1, line_step=2, frame=7):
    """Every line on a Binarizer of its own, preset with states[i].  The sticky do_ref_lvl_sweep member has no setter in the reference: where the state
    says it is set, the Binarizer has decoded a line that sets it (the first line of the set, cold) before it is given the state."""
    f = _bind(lib, prefix)
    out = np.zeros(len(rows), dtype=libs.LINE_DTYPE)
    scratch = np.zeros(1, dtype=libs.LINE_DTYPE)
    for i in range(len(rows)):
        h = C.c_void_p(f("new")())
        f("set_mode")(h, mode)
        if states is not None and states[i]["sweep_flag"]:
            f("process")(h, rows[0].ctypes.data, rows.shape[1], frame, 1, 0, 1 if doubled else 0, 0, scratch.ctypes.data)
            assert (int(scratch[0]["flags"]) & LF_BW_SET) and not (int(scratch[0]["flags"]) & LF_BY_EXT_TUNE), "the priming line did not reach the level search"
        if states is not None:
            f("set_state")(h, states[i:i + 1].ctypes.data)
        f("process")(h, rows[i].ctypes.data, rows.shape[1], frame, first_line + i * line_step, 0, 1 if doubled else 0, 0, out[i:i + 1].ctypes.data)
        f("free")(h)
    return out


def _oracle_lines(rows, states, mode, **kw):
    """The oracle, the flag set directly (orc_bin_set_state_full)."""
    lib = libs.load_oracle()
    f = _bind(lib, "orc_bin_")
    lib.orc_bin_set_state_full.argtypes = [C.c_void_p, C.c_void_p]
    first_line, line_step, frame, doubled = kw.get("first_line", 1), kw.get("line_step", 2), kw.get("frame", 7), kw.get("doubled", False)
    out = np.zeros(len(rows), dtype=libs.LINE_DTYPE)
    for i in range(len(rows)):
        h = C.c_void_p(f("new")())
        f("set_mode")(h, mode)
        if states is not None:
            lib.orc_bin_set_state_full(h, states[i:i + 1].ctypes.data)
        f("process")(h, rows[i].ctypes.data, rows.shape[1], frame, first_line + i * line_step, 0, 1 if doubled else 0, 0, out[i:i + 1].ctypes.data)
        f("free")(h)
    return out


def _states_behind(recs, mode):
    """What a worker that hands every line that read on to its Binarizer (setGoodParameters, binarizer.cpp:353-377) has preset before each line: levels and
    coordinates of the last line whose CRC held; do_ref_lvl_sweep as a line that went through the level search leaves it in this mode (:1104-1128)."""
    st = np.zeros(len(recs), dtype=STATE_DTYPE)
    cur = np.zeros(1, dtype=STATE_DTYPE)[0]
    cur["start"], cur["stop"] = -32768, 32767
    for i in range(len(recs)):
        st[i] = cur
        r = recs[i]
        cur = cur.copy()
        if (int(r["flags"]) & LF_BW_SET) and not (int(r["flags"]) & LF_BY_EXT_TUNE):
            cur["sweep_flag"] = 1 if mode in (2, 3) else 0
        if int(r["calc_crc"]) == int(r["words"][8]):
            cur["black"], cur["white"], cur["ref"] = r["black_level"], r["white_level"], r["ref_level"]
            cur["start"], cur["stop"] = r["data_start"], r["data_stop"]
    return st


def _engine_lines_host(lib, eng, rows, states, mode, doubled=False, first_line=1, line_step=2, frame=7):
    f = lib.sdv_binarize_lines
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint16, C.c_uint16, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_set_mode.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_set_mode(eng, mode)
    out = np.zeros(len(rows), dtype=libs.LINE_DTYPE)
    st = None if states is None else np.ascontiguousarray(states)
    rc = f(eng, rows.ctypes.data, rows.shape[1], rows.shape[1], len(rows), None if st is None else st.ctypes.data, frame, first_line, line_step,
           2 if doubled else 0, out.ctypes.data, len(out), None)
    return rc, out


@pytest.mark.parametrize("mode", [1, 2])
def test_oracle_per_line_driver_equals_the_real_binarizer(oracle_lib, mode):
    ref = libs.load_ref()
    if ref is None:
        pytest.skip("the reference build (oracle/_ref) exists only where /root/reference does")
    rows = _lines()
    for states in (None, "behind"):
        if states == "behind":
            states = _states_behind(_oracle_lines(rows, None, mode), mode)
        want = _per_line(ref, "ref_bin_", rows, states, mode)
        got = _oracle_lines(rows, states, mode)
        assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_emu_lines_equal_the_oracle(emu_lib, oracle_lib, mode):
    rows = _lines(seed=6)
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    cold = _oracle_lines(rows, None, mode)
    rc, got = _engine_lines_host(emu_lib, eng, rows, None, mode)
    assert rc == 0 and got.tobytes() == cold.tobytes()
    states = _states_behind(cold, mode)
    want = _oracle_lines(rows, states, mode)
    rc, got = _engine_lines_host(emu_lib, eng, rows, states, mode)
    assert rc == 0 and got.tobytes() == want.tobytes()
    assert int(((want["flags"] & LF_BY_EXT_TUNE) != 0).sum()) > len(rows) // 2          # most lines read from what they were given
    emu_lib.sdv_engine_destroy(eng)


def test_emu_lines_refuse_bad_arguments(emu_lib):
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    rows = np.zeros((2, 100), dtype=np.uint8)
    rc, _ = _engine_lines_host(emu_lib, eng, rows, None, 2)
    assert rc == 3 and b"137" in emu_lib.sdv_last_error(eng)                            # LB_RET_SHORT_LINE
    f = emu_lib.sdv_binarize_lines
    rows = np.zeros((2, 720), dtype=np.uint8); out = np.zeros(1, dtype=libs.LINE_DTYPE)
    assert f(eng, None, 720, 720, 2, None, 1, 1, 1, 0, out.ctypes.data, 1, None) == 1   # LB_RET_NULL_VIDEO
    assert f(eng, rows.ctypes.data, 720, 720, 2, None, 1, 1, 1, 0, None, 2, None) == 2  # LB_RET_NULL_PCM
    assert f(eng, rows.ctypes.data, 720, 720, 2, None, 1, 1, 1, 0, out.ctypes.data, 1, None) == -1 and b"2 line records" in emu_lib.sdv_last_error(eng)
    emu_lib.sdv_engine_destroy(eng)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,doubled", [(1, False), (2, False), (2, True), (3, False)])
def test_gpu_lines_equal_the_oracle(oracle_lib, mode, doubled):
    import torch
    from sdvpcmdecoder_amd import Engine
    rows = _lines(seed=8, width=1440 if doubled else 720) if not doubled else np.ascontiguousarray(np.repeat(_lines(seed=8), 2, axis=1))
    eng = Engine(0); eng.setBinarizationMode(mode)
    d = torch.from_numpy(rows).cuda()
    cold = _oracle_lines(rows, None, mode, doubled=doubled)
    got = eng.binarize_lines(d, None, frame_number=7, first_line=1, line_step=2, doubled=doubled).cpu().numpy()
    assert got.tobytes() == cold.tobytes()
    states = _states_behind(cold, mode)
    want = _oracle_lines(rows, states, mode, doubled=doubled)
    got = eng.binarize_lines(d, torch.from_numpy(states.view(np.uint8).reshape(len(states), 10)).cuda(), frame_number=7, first_line=1, line_step=2, doubled=doubled).cpu().numpy()
    assert got.tobytes() == want.tobytes()
    # ... and a strided view of a frame (every second row: one field), many lines
    luma, _, _ = synth.stc007_frames(6, seed=9, noise_sigma=4.0)
    if not doubled:
        field = torch.from_numpy(luma).cuda().view(-1, 720)[::2]
        want = _oracle_lines(np.ascontiguousarray(luma.reshape(-1, 720)[::2]), None, mode)
        got = eng.binarize_lines(field, None, frame_number=7, first_line=1, line_step=2).cpu().numpy()
        assert got.tobytes() == want.tobytes()
    eng.close()
