"""ctypes loaders for the test-side libraries: the CPU oracle (oracle/liborc.so) and, when it has
been built in this container, the real reference (oracle/_ref/libsdvref.so)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class LineRec(C.Structure):
    _fields_ = [("frame_number", C.c_uint32), ("line_number", C.c_uint16), ("words", C.c_uint16 * 9),
                ("calc_crc", C.c_uint16), ("data_start", C.c_int16), ("data_stop", C.c_int16),
                ("marker_start_bg_coord", C.c_uint16), ("marker_start_ed_coord", C.c_uint16),
                ("marker_stop_ed_coord", C.c_uint16),
                ("black_level", C.c_uint8), ("white_level", C.c_uint8), ("ref_low", C.c_uint8),
                ("ref_level", C.c_uint8), ("ref_high", C.c_uint8), ("hysteresis_depth", C.c_uint8),
                ("shift_stage", C.c_uint8), ("service_type", C.c_uint8), ("mark_st_stage", C.c_uint8),
                ("mark_ed_stage", C.c_uint8), ("flags", C.c_uint8), ("word_state", C.c_uint8)]


assert C.sizeof(LineRec) == 48

LINE_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (9,)),
                       ("calc_crc", "<u2"), ("data_start", "<i2"), ("data_stop", "<i2"),
                       ("marker_start_bg_coord", "<u2"), ("marker_start_ed_coord", "<u2"),
                       ("marker_stop_ed_coord", "<u2"),
                       ("black_level", "u1"), ("white_level", "u1"), ("ref_low", "u1"), ("ref_level", "u1"),
                       ("ref_high", "u1"), ("hysteresis_depth", "u1"), ("shift_stage", "u1"),
                       ("service_type", "u1"), ("mark_st_stage", "u1"), ("mark_ed_stage", "u1"),
                       ("flags", "u1"), ("word_state", "u1")])
assert LINE_DTYPE.itemsize == 48


class BinState(C.Structure):
    _fields_ = [("in_def_black", C.c_uint8), ("in_def_white", C.c_uint8), ("in_def_reference", C.c_uint8),
                ("_pad", C.c_uint8), ("in_def_start", C.c_int16), ("in_def_stop", C.c_int16),
                ("in_def_from_doubled", C.c_uint8), ("_pad2", C.c_uint8)]


class BinPreset(C.Structure):
    _fields_ = [(n, C.c_uint8) for n in ("max_black_lvl", "min_white_lvl", "min_contrast", "min_ref_lvl",
                                          "max_ref_lvl", "min_valid_crcs", "mark_max_dist", "left_bit_pick",
                                          "right_bit_pick", "en_force_coords", "en_coord_search",
                                          "en_first_line_dup", "en_good_no_marker", "_pad")] + \
               [("horiz_start", C.c_int16), ("horiz_stop", C.c_int16)]


def default_preset() -> BinPreset:
    return BinPreset(160, 28, 10, 7, 240, 5, 6, 4, 2, 0, 1, 1, 1, 0, 0, 0)


def rec_tuple(r: LineRec):
    return tuple(list(getattr(r, n)) if n == "words" else getattr(r, n) for n, _ in LineRec._fields_)


def _bind_bin_api(lib, prefix):
    g = lambda n: getattr(lib, prefix + n)
    g("bin_new").restype = C.c_void_p
    g("bin_free").argtypes = [C.c_void_p]
    g("bin_set_mode").argtypes = [C.c_void_p, C.c_int]
    g("bin_set_coord_search").argtypes = [C.c_void_p, C.c_int]
    g("bin_set_preset").argtypes = [C.c_void_p, C.POINTER(BinPreset)]
    g("bin_reset_good").argtypes = [C.c_void_p]
    g("bin_set_good_from_last").argtypes = [C.c_void_p]
    g("bin_set_state").argtypes = [C.c_void_p, C.POINTER(BinState)]
    g("bin_process").argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint16, C.c_int, C.c_int,
                                 C.c_int, C.POINTER(LineRec)]
    g("bin_process").restype = C.c_int
    g("crc_stc007").argtypes = [C.POINTER(C.c_uint16)]
    g("crc_stc007").restype = C.c_uint16


class BinApi:
    """Uniform wrapper over the per-line binarizer entry points of either library."""

    def __init__(self, lib, prefix):
        self.lib, self.p = lib, prefix
        self.h = getattr(lib, prefix + "bin_new")()

    def __getattr__(self, name):
        f = getattr(self.lib, self.p + "bin_" + name)
        return lambda *a: f(self.h, *a)

    def process_px(self, px, frame=0, line=1, service=0, doubled=0, empty=0):
        rec = LineRec()
        if px is None:
            ret = getattr(self.lib, self.p + "bin_process")(self.h, None, 0, frame, line, service, doubled, empty, C.byref(rec))
        else:
            px = np.ascontiguousarray(px, dtype=np.uint8)
            ret = getattr(self.lib, self.p + "bin_process")(self.h, px.ctypes.data, px.shape[0], frame, line, service,
                                                           doubled, empty, C.byref(rec))
        return ret, rec

    def close(self):
        getattr(self.lib, self.p + "bin_free")(self.h)


_orc = None
_ref = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def load_oracle():
    global _orc
    if _orc is None:
        path = os.path.join(ROOT, "oracle", "liborc.so")
        if not os.path.exists(path):
            build_oracle()
        _orc = C.CDLL(path)
        _bind_bin_api(_orc, "orc_")
    return _orc


def ref_available() -> bool:
    return os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsdvref.so"))


def load_ref():
    """The reference build links the image's Qt (/opt/conda/lib); preload the system libstdc++ first so the
    older one next to Qt is not picked up."""
    global _ref
    if _ref is None:
        for cand in ("/usr/lib/x86_64-linux-gnu/libstdc++.so.6",):
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
        _ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsdvref.so"))
        _bind_bin_api(_ref, "ref_")
    return _ref
