"""ctypes binding of the C-ABI in include/sdvpcm.h, usable with either the product library
(sdvpcmdecoder_amd/libsdvpcm_hip.so, device pointers) or the test-only emulator build
(tests/emu/libsdvpcm_emu.so, host pointers)."""
import ctypes as C
import numpy as np
import libs

STATS_DTYPE = np.dtype([("frame_id", "<u4"), ("line_length", "<u2"), ("lines_odd", "<u2"), ("lines_even", "<u2"),
                        ("lines_pcm_odd", "<u2"), ("lines_pcm_even", "<u2"), ("lines_bad_odd", "<u2"),
                        ("lines_bad_even", "<u2"), ("lines_dup_odd", "<u2"), ("lines_dup_even", "<u2"),
                        ("data_start", "<i2"), ("data_stop", "<i2"), ("data_from_doubled", "u1"),
                        ("data_not_sure", "u1"), ("_pad", "u1", (4,))])
assert STATS_DTYPE.itemsize == 32


class RunInfo(C.Structure):
    _fields_ = [("frames", C.c_uint32), ("rounds", C.c_uint32), ("frames_launched", C.c_uint32), ("frames_general", C.c_uint32),
                ("kernel_ms", C.c_float), ("sweeps", C.c_uint32), ("frames_met", C.c_uint32)]


class StitchInfo(C.Structure):
    _fields_ = [("steps", C.c_uint32), ("rounds", C.c_uint32), ("steps_launched", C.c_uint32), ("pipelined", C.c_uint32),
                ("device_ms", C.c_float), ("direct_frames", C.c_uint32)]


def bind(lib):
    lib.sdv_engine_create.restype = C.c_void_p
    lib.sdv_engine_create.argtypes = [C.c_int]
    lib.sdv_engine_destroy.argtypes = [C.c_void_p]
    lib.sdv_last_error.restype = C.c_char_p
    lib.sdv_last_error.argtypes = [C.c_void_p]
    lib.sdv_set_mode.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_set_bin_preset.argtypes = [C.c_void_p, C.POINTER(libs.BinPreset)]
    lib.sdv_set_check_line_dup.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_set_pcm_type.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.sdv_reset_stream.argtypes = [C.c_void_p]
    lib.sdv_get_run_info.argtypes = [C.c_void_p, C.POINTER(RunInfo)]
    lib.sdv_get_chain_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.sdv_set_chain_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.sdv_records_per_frame.restype = C.c_size_t
    lib.sdv_records_per_frame.argtypes = [C.c_int]
    lib.sdv_binarize_frames.restype = C.c_int
    lib.sdv_binarize_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                        C.c_uint32, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    import deint_api as da
    lib.sdv_deinterleave_blocks.restype = C.c_int
    lib.sdv_deinterleave_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(da.DeintSettings), C.c_void_p, C.c_size_t, C.c_void_p]
    import stitch_api as sa
    lib.sdv_default_stitch_settings.argtypes = [C.POINTER(sa.StitchSettings)]
    lib.sdv_set_stitch_settings.argtypes = [C.c_void_p, C.POINTER(sa.StitchSettings)]
    lib.sdv_reset_stitcher.argtypes = [C.c_void_p]
    lib.sdv_get_stitch_info.argtypes = [C.c_void_p, C.POINTER(StitchInfo)]
    lib.sdv_stitch_frames.restype = C.c_int
    lib.sdv_stitch_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    import pcm1_api as p1
    lib.sdv_default_pcm1_stitch_settings.argtypes = [C.POINTER(p1.Pcm1Settings)]
    lib.sdv_set_pcm1_stitch_settings.argtypes = [C.c_void_p, C.POINTER(p1.Pcm1Settings)]
    lib.sdv_pcm1_stitch_frames.restype = C.c_int
    lib.sdv_pcm1_stitch_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                           C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    return lib


def emu_stitch(lib, eng, recs, settings=None, pair_cap=None, frame_cap=None):
    """Host-memory call (emulator build only): one sdv_stitch_frames call over `recs`."""
    import stitch_api as sa
    if settings is not None:
        assert lib.sdv_set_stitch_settings(eng, C.byref(settings)) == 0
    recs = np.ascontiguousarray(recs)
    nfr = int((recs["service_type"] == 5).sum()) + 2
    pair_cap = pair_cap or nfr * 2400 + 16
    frame_cap = frame_cap or nfr * 3
    pairs = np.zeros(pair_cap, dtype=sa.PAIR_DTYPE)
    frames = np.zeros(frame_cap, dtype=sa.FRASM_DTYPE)
    npairs, nframes = C.c_size_t(0), C.c_size_t(0)
    rc = lib.sdv_stitch_frames(eng, recs.ctypes.data, len(recs), pairs.ctypes.data, pair_cap, C.byref(npairs),
                               frames.ctypes.data, frame_cap, C.byref(nframes), None)
    return rc, pairs[:npairs.value], frames[:nframes.value]


def emu_pcm1_stitch(lib, eng, recs, settings=None, pair_cap=None, frame_cap=None):
    """Host-memory call (emulator build only): one sdv_pcm1_stitch_frames call over `recs`."""
    import pcm1_api as p1
    import stitch_api as sa
    if settings is not None:
        assert lib.sdv_set_pcm1_stitch_settings(eng, C.byref(settings)) == 0
    recs = np.ascontiguousarray(recs)
    nfr = int((recs["service_type"] == 5).sum()) + 2
    pair_cap = pair_cap or nfr * 1472 + 16
    frame_cap = frame_cap or nfr + 8
    pairs = np.zeros(pair_cap, dtype=sa.PAIR_DTYPE)
    frames = np.zeros(frame_cap, dtype=p1.FRASM1_DTYPE)
    npairs, nframes = C.c_size_t(0), C.c_size_t(0)
    rc = lib.sdv_pcm1_stitch_frames(eng, recs.ctypes.data if len(recs) else None, len(recs), pairs.ctypes.data, pair_cap, C.byref(npairs),
                                    frames.ctypes.data, frame_cap, C.byref(nframes), None)
    return rc, pairs[:min(npairs.value, pair_cap)], frames[:min(nframes.value, frame_cap)]


def emu_pcm1_stitch_vis(lib, eng, recs, settings=None, blocks=True, lines=True, block_cap=None, line_cap=None):
    """... with the visualiser's feeds switched on (sdv_set_pcm1_stitch_block_output / _line_output): (rc, pairs, frames, blocks, sub-lines)."""
    import pcm1_api as p1
    for nm in ("sdv_set_pcm1_stitch_block_output", "sdv_set_pcm1_stitch_line_output"):
        getattr(lib, nm).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for nm in ("sdv_pcm1_stitch_block_count", "sdv_pcm1_stitch_line_count"):
        getattr(lib, nm).restype = C.c_size_t
        getattr(lib, nm).argtypes = [C.c_void_p]
    nfr = int((recs["service_type"] == 5).sum()) + 2
    bl = np.zeros(block_cap if block_cap is not None else nfr * 16, dtype=p1.BLOCK1_DTYPE)
    ln = np.zeros(line_cap if line_cap is not None else nfr * 1470, dtype=p1.ASM1_DTYPE)
    assert lib.sdv_set_pcm1_stitch_block_output(eng, bl.ctypes.data if blocks else None, len(bl)) == 0
    assert lib.sdv_set_pcm1_stitch_line_output(eng, ln.ctypes.data if lines else None, len(ln)) == 0
    rc, pairs, frames = emu_pcm1_stitch(lib, eng, recs, settings)
    nb, nl = lib.sdv_pcm1_stitch_block_count(eng), lib.sdv_pcm1_stitch_line_count(eng)
    emu_pcm1_stitch_vis.last_counts = (nb, nl)          # what the call made (or needed)
    assert lib.sdv_set_pcm1_stitch_block_output(eng, None, 0) == 0 and lib.sdv_set_pcm1_stitch_line_output(eng, None, 0) == 0
    return rc, pairs, frames, bl[:min(nb, len(bl))].copy(), ln[:min(nl, len(ln))].copy()


def emu_pcm16_stitch_vis(lib, eng, recs, settings=None, block_cap=None):
    """... with the visualiser's block feed switched on (sdv_set_pcm16x0_stitch_block_output): (rc, pairs, frames, blocks)."""
    import pcm16_api as p16
    lib.sdv_set_pcm16x0_stitch_block_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_pcm16x0_stitch_block_count.restype = C.c_size_t
    lib.sdv_pcm16x0_stitch_block_count.argtypes = [C.c_void_p]
    nfr = int((recs["service_type"] == 5).sum()) + 2
    bl = np.zeros(block_cap if block_cap is not None else nfr * 800 + 16, dtype=p16.VBLOCK16_DTYPE)
    assert lib.sdv_set_pcm16x0_stitch_block_output(eng, bl.ctypes.data, len(bl)) == 0
    rc, pairs, frames = emu_pcm16_stitch(lib, eng, recs, settings)
    nb = lib.sdv_pcm16x0_stitch_block_count(eng)
    emu_pcm16_stitch_vis.last_count = nb
    assert lib.sdv_set_pcm16x0_stitch_block_output(eng, None, 0) == 0
    return rc, pairs, frames, bl[:min(nb, len(bl))].copy()


def emu_pcm16_stitch_lines(lib, eng, recs, settings=None, line_cap=None):
    """... with the assembled sub-lines switched on (sdv_set_pcm16x0_stitch_line_output): (rc, pairs, frames, lines)."""
    lib.sdv_set_pcm16x0_stitch_line_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_pcm16x0_stitch_line_count.restype = C.c_size_t
    lib.sdv_pcm16x0_stitch_line_count.argtypes = [C.c_void_p]
    nfr = int((recs["service_type"] == 5).sum()) + 2
    ln = np.zeros(line_cap if line_cap is not None else nfr * 3000 + 16, dtype=recs.dtype)
    assert lib.sdv_set_pcm16x0_stitch_line_output(eng, ln.ctypes.data, len(ln)) == 0
    rc, pairs, frames = emu_pcm16_stitch(lib, eng, recs, settings)
    nl = lib.sdv_pcm16x0_stitch_line_count(eng)
    emu_pcm16_stitch_lines.last_count = nl
    assert lib.sdv_set_pcm16x0_stitch_line_output(eng, None, 0) == 0
    return rc, pairs, frames, ln[:min(nl, len(ln))].copy()


def emu_pcm16_stitch(lib, eng, recs, settings=None, pair_cap=None, frame_cap=None):
    """Host-memory call (emulator build only): one sdv_pcm16x0_stitch_frames call over `recs`."""
    import pcm16_api as p16
    import stitch_api as sa
    lib.sdv_set_pcm16x0_stitch_settings.argtypes = [C.c_void_p, C.c_void_p]
    lib.sdv_pcm16x0_stitch_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                              C.POINTER(C.c_size_t), C.c_void_p]
    if settings is not None:
        assert lib.sdv_set_pcm16x0_stitch_settings(eng, C.byref(settings)) == 0
    recs = np.ascontiguousarray(recs)
    nfr = int((recs["service_type"] == 5).sum()) + 2
    pair_cap = pair_cap or nfr * 1472 + 16
    frame_cap = frame_cap or nfr + 8
    pairs = np.zeros(pair_cap, dtype=sa.PAIR_DTYPE)
    frames = np.zeros(frame_cap, dtype=p16.FRASM16_DTYPE)
    npairs, nframes = C.c_size_t(0), C.c_size_t(0)
    rc = lib.sdv_pcm16x0_stitch_frames(eng, recs.ctypes.data if len(recs) else None, len(recs), pairs.ctypes.data, pair_cap, C.byref(npairs),
                                       frames.ctypes.data, frame_cap, C.byref(nframes), None)
    return rc, pairs[:min(npairs.value, pair_cap)], frames[:min(nframes.value, frame_cap)]


def emu_binarize(lib, eng, luma, first_frame_no=1, flags=1, row_stride=None, misalign=0):
    """Host-memory call (emulator build only).  row_stride / misalign: the same pixels in a padded, shifted buffer."""
    n, h, w = luma.shape
    if row_stride is not None or misalign:
        rs = row_stride or w
        buf = np.full(n * h * rs + misalign + 64, 0x5A, dtype=np.uint8)
        view = buf[misalign:misalign + n * h * rs].reshape(n, h, rs)
        view[:, :, :w] = luma
        nrec = n * (h + 3) + (1 if flags & 1 else 0) + (h + 4 if flags & 4 else 0)
        recs = np.zeros(nrec, dtype=libs.LINE_DTYPE)
        stats = np.zeros(n + (1 if flags & 4 else 0), dtype=STATS_DTYPE)
        rc = lib.sdv_binarize_frames(eng, buf.ctypes.data + misalign, rs, rs * h, w, h, n, first_frame_no, flags, recs.ctypes.data, len(recs),
                                     stats.ctypes.data, len(stats), None)
        return rc, recs, stats
    nrec = n * (h + 3) + (1 if flags & 1 else 0) + (h + 4 if flags & 4 else 0)
    recs = np.zeros(nrec, dtype=libs.LINE_DTYPE)
    stats = np.zeros(n + (1 if flags & 4 else 0), dtype=STATS_DTYPE)
    luma = np.ascontiguousarray(luma)
    rc = lib.sdv_binarize_frames(eng, luma.ctypes.data, w, w * h, w, h, n, first_frame_no, flags, recs.ctypes.data, len(recs),
                                 stats.ctypes.data, len(stats), None)
    return rc, recs, stats
