"""Runs the CPU oracle (oracle/liborc.so) at the VideoToDigital level. Test-side only."""
import ctypes as C
import numpy as np
import libs


def oracle_binarize(luma, mode=2, first_frame_no=1, new_file=True, doubled=False, preset=None, check_line_dup=True,
                    m2=False, handle=None, return_state=False, end_file=False):
    lib = libs.load_oracle()
    lib.orc_v2d_new.restype = C.c_void_p
    lib.orc_v2d_run.restype = C.c_long
    lib.orc_v2d_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p]
    lib.orc_v2d_delete.argtypes = [C.c_void_p]
    lib.orc_v2d_set_mode.argtypes = [C.c_void_p, C.c_int]
    lib.orc_v2d_set_check_line_dup.argtypes = [C.c_void_p, C.c_int]
    lib.orc_v2d_set_m2.argtypes = [C.c_void_p, C.c_int]
    lib.orc_v2d_set_preset.argtypes = [C.c_void_p, C.POINTER(libs.BinPreset)]
    lib.orc_v2d_get_state.argtypes = [C.c_void_p, C.c_void_p]
    own = handle is None
    h = C.c_void_p(lib.orc_v2d_new()) if own else handle
    if own:
        lib.orc_v2d_set_mode(h, mode)
        lib.orc_v2d_set_check_line_dup(h, int(check_line_dup))
        lib.orc_v2d_set_m2(h, int(m2))
        if preset is not None:
            lib.orc_v2d_set_preset(h, C.byref(preset))
    luma = np.ascontiguousarray(luma, dtype=np.uint8)
    n, hgt, w = luma.shape
    recs = np.zeros(n * (hgt + 3) + (1 if new_file else 0) + (hgt + 4 if end_file else 0), dtype=libs.LINE_DTYPE)
    stats = np.zeros((n + (1 if end_file else 0), 32), dtype=np.uint8)
    got = lib.orc_v2d_run(h, luma.ctypes.data, w, w, hgt, n, first_frame_no, int(new_file) | (2 if end_file else 0), int(doubled), recs.ctypes.data,
                          stats.ctypes.data)
    assert got == len(recs)
    state = None
    if return_state:
        state = np.zeros(120, dtype=np.uint8)
        lib.orc_v2d_get_state(h, state.ctypes.data)
    if own:
        lib.orc_v2d_delete(h)
    return (recs, stats, state) if return_state else (recs, stats)
