"""Python mirror of the reference's operator interface for the binarize path.

`Engine` exposes the VideoToDigital slots a Qt front-end drives (videotodigital.h:137-147) with the same
names and argument meaning - setPCMType, setBinarizationMode, setCheckLineDup, setFineSettings,
setDefaultFineSettings - plus `binarize_frames`, the batch replacement of the doBinarize worker loop.
Everything computes on the GPU through the C-ABI; torch is only used for device memory."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG, "libsdvpcm_hip.so")

LINE_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (9,)),
                       ("calc_crc", "<u2"), ("data_start", "<i2"), ("data_stop", "<i2"),
                       ("marker_start_bg_coord", "<u2"), ("marker_start_ed_coord", "<u2"),
                       ("marker_stop_ed_coord", "<u2"),
                       ("black_level", "u1"), ("white_level", "u1"), ("ref_low", "u1"), ("ref_level", "u1"),
                       ("ref_high", "u1"), ("hysteresis_depth", "u1"), ("shift_stage", "u1"),
                       ("service_type", "u1"), ("mark_st_stage", "u1"), ("mark_ed_stage", "u1"),
                       ("flags", "u1"), ("word_state", "u1")])
STATS_DTYPE = np.dtype([("frame_id", "<u4"), ("line_length", "<u2"), ("lines_odd", "<u2"), ("lines_even", "<u2"),
                        ("lines_pcm_odd", "<u2"), ("lines_pcm_even", "<u2"), ("lines_bad_odd", "<u2"),
                        ("lines_bad_even", "<u2"), ("lines_dup_odd", "<u2"), ("lines_dup_even", "<u2"),
                        ("data_start", "<i2"), ("data_stop", "<i2"), ("data_from_doubled", "u1"),
                        ("data_not_sure", "u1"), ("_pad", "u1", (4,))])
assert LINE_DTYPE.itemsize == 48 and STATS_DTYPE.itemsize == 32

# enums of include/sdvpcm.h
PCM_PCM1, PCM_PCM16X0, PCM_STC007 = 0, 1, 2
TYPE_M2 = 3                      # VideoToDigital::TYPE_M2 (videotodigital.h:77)
MODE_DRAFT, MODE_FAST, MODE_NORMAL, MODE_INSANE = 0, 1, 2, 3
FLAG_NEW_FILE, FLAG_DOUBLED, FLAG_END_FILE = 1, 2, 4
FRAME_EMPTY = 1                 # sdv_set_frame_flags: SDV_FRAME_EMPTY
VIS_STC007_LINES, VIS_PCM1_LINES, VIS_PCM16X0_LINES, VIS_STC007_BLOCKS_NTSC, VIS_STC007_BLOCKS_PAL, VIS_STC007_ASM_NTSC, VIS_STC007_ASM_PAL = 0, 1, 2, 3, 4, 5, 6   # SDV_VIS_*


class BinPreset(C.Structure):
    """bin_preset_t (binarizer.h:163-186)"""
    _fields_ = [(n, C.c_uint8) for n in ("max_black_lvl", "min_white_lvl", "min_contrast", "min_ref_lvl",
                                          "max_ref_lvl", "min_valid_crcs", "mark_max_dist", "left_bit_pick",
                                          "right_bit_pick", "en_force_coords", "en_coord_search",
                                          "en_first_line_dup", "en_good_no_marker", "_pad")] + \
               [("horiz_start", C.c_int16), ("horiz_stop", C.c_int16)]


class DeintSettings(C.Structure):
    """STC007Deinterleaver settings (stc007deinterleaver.h:151-161)"""
    _fields_ = [("res_mode", C.c_uint8), ("ignore_crc", C.c_uint8), ("force_ecc_check", C.c_uint8),
                ("en_p_code", C.c_uint8), ("en_q_code", C.c_uint8), ("en_cwd", C.c_uint8), ("_pad", C.c_uint8 * 2)]


DEINT_LINE_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (8,)),
                             ("word_crc_ok", "u1"), ("flags", "u1")])
BLOCK_DTYPE = np.dtype([("w_frame", "<u4", (8,)), ("w_line", "<u2", (8,)), ("words", "<u2", (8,)),
                        ("line_crc", "u1"), ("cwd_fixed", "u1"), ("word_valid", "u1"), ("resolution", "u1"),
                        ("audio_state", "u1"), ("cwd_applied", "u1"), ("sample_rate", "<u2")])
assert DEINT_LINE_DTYPE.itemsize == 24 and BLOCK_DTYPE.itemsize == 72


class StitchSettings(C.Structure):
    """STC007DataStitcher settings (slots stc007datastitcher.h:331-350)"""
    _fields_ = [("video_standard", C.c_uint8), ("field_order", C.c_uint8), ("enable_p", C.c_uint8), ("enable_q", C.c_uint8),
                ("enable_cwd", C.c_uint8), ("m2_format", C.c_uint8), ("resolution_preset", C.c_uint8), ("max_unch_14", C.c_uint8),
                ("max_unch_16", C.c_uint8), ("use_ecc", C.c_uint8), ("mask_seams", C.c_uint8), ("broke_mask", C.c_uint8),
                ("top_line_fix", C.c_uint8), ("_pad", C.c_uint8), ("sample_rate_preset", C.c_uint16)]


class Pcm1StitchSettings(C.Structure):
    """PCM1DataStitcher settings (slots pcm1datastitcher.h:183-190)"""
    _fields_ = [("field_order", C.c_uint8), ("auto_offset", C.c_uint8), ("use_ecc", C.c_uint8), ("odd_offset", C.c_int8),
                ("even_offset", C.c_int8), ("_pad", C.c_uint8 * 3)]


class Pcm16x0StitchSettings(C.Structure):
    """PCM16X0DataStitcher settings (slots pcm16x0datastitcher.h:304-314)"""
    _fields_ = [("format", C.c_uint8), ("field_order", C.c_uint8), ("p_correction", C.c_uint8), ("use_ecc", C.c_uint8),
                ("mask_seams", C.c_uint8), ("broke_mask", C.c_uint8), ("sample_rate_preset", C.c_uint16)]


class StitchInfo(C.Structure):
    _fields_ = [("steps", C.c_uint32), ("rounds", C.c_uint32), ("steps_launched", C.c_uint32), ("pipelined", C.c_uint32),
                ("device_ms", C.c_float), ("direct_frames", C.c_uint32)]


PAIR_DTYPE = np.dtype([("audio_word", "<i2", (2,)), ("sample_flags", "u1", (2,)), ("sample_rate", "<u2"),
                       ("emphasis", "u1"), ("service_type", "u1"), ("_pad", "<u2")])
assert PAIR_DTYPE.itemsize == 12 and C.sizeof(StitchSettings) == 16


class RunInfo(C.Structure):
    _fields_ = [("frames", C.c_uint32), ("rounds", C.c_uint32), ("frames_launched", C.c_uint32), ("frames_general", C.c_uint32),
                ("kernel_ms", C.c_float), ("sweeps", C.c_uint32), ("frames_met", C.c_uint32)]


_lib = None


def _check_out(t, rec_bytes, device, name):
    """Caller-supplied output tensors: (n, rec_bytes) uint8, contiguous, on the frames' device."""
    import torch
    if not (t.is_cuda and t.device == device and t.dtype == torch.uint8 and t.dim() == 2 and t.shape[1] == rec_bytes and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous torch.uint8 CUDA tensor of shape (n, {rec_bytes}) on {device}")


def load_library(path: str | None = None):
    """Loads libsdvpcm_hip.so. Raises (never falls back) when the HIP extension is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SDVPCM_LIB") or _LIB_PATH   # SDVPCM_LIB: alternative builds of the same HIP library (tuning experiments)
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(p)
    lib.sdv_engine_create.restype = C.c_void_p
    lib.sdv_engine_create.argtypes = [C.c_int]
    lib.sdv_engine_destroy.argtypes = [C.c_void_p]
    lib.sdv_last_error.restype = C.c_char_p
    lib.sdv_last_error.argtypes = [C.c_void_p]
    lib.sdv_abi_version.restype = C.c_int
    lib.sdv_default_bin_preset.argtypes = [C.POINTER(BinPreset)]
    lib.sdv_set_mode.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_set_bin_preset.argtypes = [C.c_void_p, C.POINTER(BinPreset)]
    lib.sdv_set_check_line_dup.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_set_pcm_type.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.sdv_reset_stream.argtypes = [C.c_void_p]
    lib.sdv_set_frame_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_needs_double_width.argtypes = [C.c_int]
    lib.sdv_double_width.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_vis_canvas_size.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.sdv_vis_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.sdv_vis_render_lines.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    lib.sdv_vis_render_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_set_stitch_block_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_stitch_block_count.restype = C.c_size_t
    lib.sdv_stitch_block_count.argtypes = [C.c_void_p]
    for nm in ("sdv_set_pcm1_stitch_block_output", "sdv_set_pcm1_stitch_line_output", "sdv_set_pcm16x0_stitch_block_output", "sdv_set_pcm16x0_stitch_line_output"):
        getattr(lib, nm).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    for nm in ("sdv_pcm1_stitch_block_count", "sdv_pcm1_stitch_line_count", "sdv_pcm16x0_stitch_block_count", "sdv_pcm16x0_stitch_line_count"):
        getattr(lib, nm).restype = C.c_size_t
        getattr(lib, nm).argtypes = [C.c_void_p]
    lib.sdv_set_stitch_line_output.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_stitch_line_count.restype = C.c_size_t
    lib.sdv_stitch_line_count.argtypes = [C.c_void_p]
    lib.sdv_stitch_line_counts.restype = C.c_size_t
    lib.sdv_stitch_line_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_vis_render_asm_lines.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_get_run_info.argtypes = [C.c_void_p, C.POINTER(RunInfo)]
    lib.sdv_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_get_chain_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.sdv_set_chain_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.sdv_records_per_frame.restype = C.c_size_t
    lib.sdv_records_per_frame.argtypes = [C.c_int]
    lib.sdv_binarize_frames.restype = C.c_int
    lib.sdv_binarize_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                        C.c_uint32, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_binarize_records.restype = C.c_size_t
    lib.sdv_binarize_records.argtypes = [C.c_int, C.c_int, C.c_uint]
    lib.sdv_pcm1_binarize_frames.restype = C.c_int
    lib.sdv_pcm1_binarize_frames.argtypes = lib.sdv_binarize_frames.argtypes
    lib.sdv_pcm16x0_binarize_frames.restype = C.c_int
    lib.sdv_pcm16x0_binarize_frames.argtypes = lib.sdv_binarize_frames.argtypes
    lib.sdv_pcm16x0_binarize_records.restype = C.c_size_t
    lib.sdv_pcm16x0_binarize_records.argtypes = [C.c_int, C.c_int, C.c_uint]
    if hasattr(lib, "sdv_binarize_lines"):      # (SDVPCM_LIB may name a build of an earlier ABI: tuning comparisons across rounds)
        lib.sdv_binarize_lines.restype = C.c_int
        lib.sdv_binarize_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint16, C.c_uint16,
                                           C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_pcm1_binarize_lines.restype = C.c_int
    lib.sdv_pcm1_binarize_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint16,
                                            C.c_uint16, C.c_uint, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.sdv_pcm16x0_binarize_lines.restype = C.c_int
    lib.sdv_pcm16x0_binarize_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint16,
                                               C.c_uint16, C.c_uint, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.sdv_default_deint_settings.argtypes = [C.POINTER(DeintSettings)]
    lib.sdv_deinterleave_blocks.restype = C.c_int
    lib.sdv_deinterleave_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(DeintSettings), C.c_void_p, C.c_size_t,
                                            C.c_void_p]
    lib.sdv_default_stitch_settings.argtypes = [C.POINTER(StitchSettings)]
    lib.sdv_set_stitch_settings.argtypes = [C.c_void_p, C.POINTER(StitchSettings)]
    lib.sdv_reset_stitcher.argtypes = [C.c_void_p]
    lib.sdv_get_stitch_info.argtypes = [C.c_void_p, C.POINTER(StitchInfo)]
    lib.sdv_default_pcm1_stitch_settings.argtypes = [C.POINTER(Pcm1StitchSettings)]
    lib.sdv_set_pcm1_stitch_settings.argtypes = [C.c_void_p, C.POINTER(Pcm1StitchSettings)]
    lib.sdv_pcm1_stitch_frames.restype = C.c_int
    lib.sdv_pcm1_stitch_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                           C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    lib.sdv_pcm1_bin_to_line_recs.restype = C.c_int
    lib.sdv_pcm1_bin_to_line_recs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.sdv_default_pcm16x0_stitch_settings.argtypes = [C.POINTER(Pcm16x0StitchSettings)]
    lib.sdv_set_pcm16x0_stitch_settings.argtypes = [C.c_void_p, C.POINTER(Pcm16x0StitchSettings)]
    lib.sdv_pcm16x0_stitch_frames.restype = C.c_int
    lib.sdv_pcm16x0_stitch_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                              C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    lib.sdv_stitch_frames.restype = C.c_int
    lib.sdv_stitch_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]
    lib.sdv_stitch_state_size.restype = C.c_size_t
    lib.sdv_get_stitch_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_set_stitch_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_saturate_stitch_stats.argtypes = [C.c_void_p]
    lib.sdv_pcm16x0_chain_state_size.restype = C.c_size_t
    lib.sdv_get_pcm16x0_chain_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_set_pcm16x0_chain_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_pcm16x0_stitch_state_size.restype = C.c_size_t
    lib.sdv_get_pcm16x0_stitch_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_set_pcm16x0_stitch_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.sdv_saturate_pcm16x0_stitch_stats.argtypes = [C.c_void_p]
    lib.sdv_set_audio_masking.argtypes = [C.c_void_p, C.c_int]
    lib.sdv_reset_audio.argtypes = [C.c_void_p]
    lib.sdv_audio_pending.restype = C.c_size_t
    lib.sdv_audio_pending.argtypes = [C.c_void_p]
    lib.sdv_audio_stalled.restype = C.c_int
    lib.sdv_audio_stalled.argtypes = [C.c_void_p]
    lib.sdv_audio_next_index.restype = C.c_uint64
    lib.sdv_audio_next_index.argtypes = [C.c_void_p]
    lib.sdv_audio_process.restype = C.c_int
    lib.sdv_audio_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_void_p]
    lib.sdv_wav_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.sdv_wav_header.argtypes = [C.c_void_p, C.c_uint64, C.c_uint16]
    lib.sdv_wav_header.restype = None
    lib.sdv_decode_frames.restype = C.c_int
    lib.sdv_decode_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint,
                                      C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                      C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_void_p]
    if path is None:
        _lib = lib
    return lib


AUDIO_PURGE_DTYPE = __import__("numpy").dtype([("first_pair", "<u8"), ("tag_index", "<u4"), ("kind", "u1"), ("_pad", "u1", (3,))])


class Engine:
    """One decode engine per GPU (per process rank)."""

    def __init__(self, device: int = 0, lib=None):
        # The tensors this class takes are torch's: let torch load and bring up the HIP runtime it ships before the library is loaded
        # and makes its first HIP call - the other way round two runtimes end up in the process and one of them finds no device
        # ("No HIP GPUs are available" / "no HIP device available").
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        self.lib = lib or load_library()
        self.device = device
        self._h = self.lib.sdv_engine_create(device)
        if not self._h:
            raise RuntimeError("sdv_engine_create failed: " + self.lib.sdv_last_error(None).decode())
        self._h = C.c_void_p(self._h)

    def close(self):
        if self._h:
            self.lib.sdv_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"sdvpcm error {rc}: " + self.lib.sdv_last_error(self._h).decode())

    # ---- VideoToDigital slots (videotodigital.h:137-147) ----
    def setPCMType(self, in_pcm: int = PCM_STC007):
        if in_pcm == TYPE_M2:
            self._check(self.lib.sdv_set_pcm_type(self._h, PCM_STC007, 1))
        else:
            self._check(self.lib.sdv_set_pcm_type(self._h, in_pcm, 0))

    def setBinarizationMode(self, in_mode: int = MODE_NORMAL):
        self._check(self.lib.sdv_set_mode(self._h, in_mode))

    def setCheckLineDup(self, flag: bool = True):
        self._check(self.lib.sdv_set_check_line_dup(self._h, int(flag)))

    def getDefaultFineSettings(self) -> BinPreset:
        p = BinPreset()
        self.lib.sdv_default_bin_preset(C.byref(p))
        return p

    def setFineSettings(self, in_set: BinPreset):
        self._check(self.lib.sdv_set_bin_preset(self._h, C.byref(in_set)))

    def setDefaultFineSettings(self):
        self.setFineSettings(self.getDefaultFineSettings())

    def reset_stream(self):
        self._check(self.lib.sdv_reset_stream(self._h))

    def set_frame_flags(self, flags):
        """Per-frame marks for the next frame entry call (binarize_frames / pcm1_binarize_frames / pcm16x0_binarize_frames / decode_frames):
        flags[i] & FRAME_EMPTY = frame i of that call was dropped by the video input (VideoInFFMPEG::insertDummyFrame(false, true))."""
        import numpy as np
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(self.lib.sdv_set_frame_flags(self._h, fl.ctypes.data if len(fl) else None, len(fl)))

    def needs_double_width(self, width: int) -> bool:
        return bool(self.lib.sdv_needs_double_width(width))

    def double_width(self, luma, stream=None):
        """The integer 2x width doubler (sdv_double_width): luma (..., width) uint8 CUDA tensor with contiguous rows -> (..., 2 * width)."""
        import torch
        assert luma.is_cuda and luma.dtype == torch.uint8 and luma.stride(-1) == 1 and luma.is_contiguous()
        w = luma.shape[-1]
        rows = luma.numel() // w
        out = torch.empty(luma.shape[:-1] + (2 * w,), dtype=torch.uint8, device=luma.device)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(luma.device).cuda_stream)
        self._check(self.lib.sdv_double_width(self._h, C.c_void_p(luma.data_ptr()), w, w, rows, C.c_void_p(out.data_ptr()), 2 * w, sptr))
        return out

    # ---- visualiser feed (RenderPCM's canvas of binarized lines) ----
    def vis_canvas_size(self, kind: int):
        w, h = C.c_uint32(0), C.c_uint32(0)
        self._check(self.lib.sdv_vis_canvas_size(kind, C.byref(w), C.byref(h)))
        return w.value, h.value

    def vis_reset(self, kind: int, stream=None):
        import torch
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._check(self.lib.sdv_vis_reset(self._h, kind, sptr))

    def vis_render_lines(self, kind: int, recs, n_frames: int, stream=None):
        """sdv_vis_render_lines: `recs` = (n, record size) uint8 CUDA tensor of the kind's line records holding `n_frames` whole frames
        -> (n_frames, height, width) int32 CUDA tensor of 32-bit pixels (view it as uint32 on the host)."""
        import torch
        assert recs.is_cuda and recs.dtype == torch.uint8 and recs.is_contiguous()
        w, h = self.vis_canvas_size(kind)
        out = torch.empty((max(n_frames, 1), h, w), dtype=torch.int32, device=recs.device)
        got = C.c_size_t(0)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(recs.device).cuda_stream)
        self._check(self.lib.sdv_vis_render_lines(self._h, kind, C.c_void_p(recs.data_ptr()), recs.shape[0], C.c_void_p(out.data_ptr()), n_frames,
                                                  C.byref(got), sptr))
        return out[:got.value]

    def set_stitch_block_output(self, blocks):
        """sdv_set_stitch_block_output: `blocks` = (cap, 72) uint8 CUDA tensor that the following stitch_frames calls fill with their data
        blocks (None: off).  The tensor must stay alive while it is set."""
        self._block_out = blocks
        if blocks is None:
            self._check(self.lib.sdv_set_stitch_block_output(self._h, None, 0))
        else:
            assert blocks.is_cuda and blocks.is_contiguous() and blocks.shape[1] == 72
            self._check(self.lib.sdv_set_stitch_block_output(self._h, C.c_void_p(blocks.data_ptr()), blocks.shape[0]))

    def stitch_block_count(self) -> int:
        return int(self.lib.sdv_stitch_block_count(self._h))

    def set_stitch_line_output(self, lines):
        """sdv_set_stitch_line_output: `lines` = (cap, 32) uint8 CUDA tensor that the following stitch_frames calls fill with the assembled lines
        they hand to the visualiser (None: off).  The tensor must stay alive while it is set."""
        self._line_out = lines
        if lines is None:
            self._check(self.lib.sdv_set_stitch_line_output(self._h, None, 0))
        else:
            assert lines.is_cuda and lines.is_contiguous() and lines.shape[1] == 32
            self._check(self.lib.sdv_set_stitch_line_output(self._h, C.c_void_p(lines.data_ptr()), lines.shape[0]))

    def stitch_line_count(self) -> int:
        return int(self.lib.sdv_stitch_line_count(self._h))

    def stitch_line_counts(self):
        """Lines per stitcher turn of the last stitch_frames call (numpy uint32)."""
        import numpy as np
        n = int(self.lib.sdv_stitch_line_counts(self._h, None, 0))
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self.lib.sdv_stitch_line_counts(self._h, out.ctypes.data, n)
        return out[:n]

    def vis_render_asm_lines(self, kind: int, lines, frame_lines, stream=None):
        """sdv_vis_render_asm_lines: `lines` = (n, 32) uint8 CUDA tensor of sdv_asm_line_rec, frame_lines = lines per frame (host sequence)
        -> (n_frames, height, width) int32 CUDA tensor."""
        import numpy as np
        import torch
        assert lines.is_cuda and lines.dtype == torch.uint8 and lines.is_contiguous()
        per = np.ascontiguousarray(np.asarray(frame_lines, dtype=np.uint32))
        w, h = self.vis_canvas_size(kind)
        out = torch.empty((max(len(per), 1), h, w), dtype=torch.int32, device=lines.device)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(lines.device).cuda_stream)
        self._check(self.lib.sdv_vis_render_asm_lines(self._h, kind, C.c_void_p(lines.data_ptr()), lines.shape[0], per.ctypes.data, len(per),
                                                      C.c_void_p(out.data_ptr()), len(per), sptr))
        return out[:len(per)]

    def vis_render_blocks(self, kind: int, blocks, frame_blocks, stream=None):
        """sdv_vis_render_blocks: `blocks` = (n, 72) uint8 CUDA tensor of sdv_block_rec, frame_blocks = blocks per frame (host sequence)
        -> (n_frames, height, width) int32 CUDA tensor."""
        import numpy as np
        import torch
        assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous()
        per = np.ascontiguousarray(np.asarray(frame_blocks, dtype=np.uint32))
        w, h = self.vis_canvas_size(kind)
        out = torch.empty((max(len(per), 1), h, w), dtype=torch.int32, device=blocks.device)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(blocks.device).cuda_stream)
        self._check(self.lib.sdv_vis_render_blocks(self._h, kind, C.c_void_p(blocks.data_ptr()), blocks.shape[0], per.ctypes.data, len(per),
                                                   C.c_void_p(out.data_ptr()), len(per), sptr))
        return out[:len(per)]

    # ---- stream state as bytes (checkpoints, hand-over between the GPUs of a sharded stream) ----
    def get_chain_state(self) -> bytes:
        buf = C.create_string_buffer(120)
        self._check(self.lib.sdv_get_chain_state(self._h, buf))
        return buf.raw

    def set_chain_state(self, state: bytes):
        assert len(state) == 120
        self._check(self.lib.sdv_set_chain_state(self._h, C.create_string_buffer(state, 120)))

    def get_stitch_state(self) -> bytes:
        n = self.lib.sdv_stitch_state_size()
        buf = C.create_string_buffer(n)
        self._check(self.lib.sdv_get_stitch_state(self._h, buf, n))
        return buf.raw

    def set_stitch_state(self, state: bytes):
        self._check(self.lib.sdv_set_stitch_state(self._h, C.create_string_buffer(state, len(state)), len(state)))

    def saturate_stitch_stats(self):
        self._check(self.lib.sdv_saturate_stitch_stats(self._h))

    def get_pcm16x0_chain_state(self) -> bytes:
        n = self.lib.sdv_pcm16x0_chain_state_size()
        buf = C.create_string_buffer(n)
        self._check(self.lib.sdv_get_pcm16x0_chain_state(self._h, buf, n))
        return buf.raw

    def set_pcm16x0_chain_state(self, state: bytes):
        self._check(self.lib.sdv_set_pcm16x0_chain_state(self._h, C.create_string_buffer(state, len(state)), len(state)))

    def get_pcm16x0_stitch_state(self) -> bytes:
        n = self.lib.sdv_pcm16x0_stitch_state_size()
        buf = C.create_string_buffer(n)
        self._check(self.lib.sdv_get_pcm16x0_stitch_state(self._h, buf, n))
        return buf.raw

    def set_pcm16x0_stitch_state(self, state: bytes):
        self._check(self.lib.sdv_set_pcm16x0_stitch_state(self._h, C.create_string_buffer(state, len(state)), len(state)))

    def saturate_pcm16x0_stitch_stats(self):
        self._check(self.lib.sdv_saturate_pcm16x0_stitch_stats(self._h))

    def set_profiling(self, on: bool = True):
        self._check(self.lib.sdv_set_profiling(self._h, int(on)))

    def run_info(self) -> RunInfo:
        info = RunInfo()
        self.lib.sdv_get_run_info(self._h, C.byref(info))
        return info

    # ---- STC007Deinterleaver (stc007deinterleaver.h:163-176) ----
    def default_deint_settings(self) -> DeintSettings:
        st = DeintSettings()
        self.lib.sdv_default_deint_settings(C.byref(st))
        return st

    def deinterleave_blocks(self, lines, settings: DeintSettings, n_blocks: int | None = None, out=None, stream=None):
        """processBlock(line_shift) for line_shift = 0..n_blocks-1 over `lines` (torch.uint8 CUDA tensor (n_lines, 24)).
        Returns torch.uint8 CUDA tensor (n_blocks, 72) of sdv_block_rec."""
        import torch
        assert lines.is_cuda and lines.dtype == torch.uint8 and lines.dim() == 2 and lines.shape[1] == 24 and lines.is_contiguous()
        n_lines = lines.shape[0]
        if n_blocks is None:
            n_blocks = max(0, n_lines - 112)
        if out is None:
            out = torch.empty((n_blocks, 72), dtype=torch.uint8, device=lines.device)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(lines.device).cuda_stream)
        self._check(self.lib.sdv_deinterleave_blocks(self._h, C.c_void_p(lines.data_ptr()), n_lines, C.byref(settings),
                                                     C.c_void_p(out.data_ptr()), n_blocks, sptr))
        return out

    # ---- STC007DataStitcher (stc007datastitcher.h:331-350) ----
    def default_stitch_settings(self) -> StitchSettings:
        st = StitchSettings()
        self.lib.sdv_default_stitch_settings(C.byref(st))
        return st

    def set_stitch_settings(self, st: StitchSettings):
        self._check(self.lib.sdv_set_stitch_settings(self._h, C.byref(st)))

    def reset_stitcher(self):
        self._check(self.lib.sdv_reset_stitcher(self._h))

    def stitch_info(self) -> StitchInfo:
        info = StitchInfo()
        self.lib.sdv_get_stitch_info(self._h, C.byref(info))
        return info

    def stitch_frames(self, lines, out_pairs=None, out_frames=None, stream=None):
        """doFrameReassemble over a span of the binarized line stream: `lines` is a torch.uint8 CUDA tensor (n_records, 48)
        (what binarize_frames returns).  Returns (pairs, frames): torch.uint8 CUDA tensors (n_pairs, 12) of sdv_sample_pair
        and (n_frames, 64) of sdv_frame_asm - views of out_pairs / out_frames when those are given."""
        import torch
        assert lines.is_cuda and lines.dtype == torch.uint8 and lines.dim() == 2 and lines.shape[1] == 48 and lines.is_contiguous()
        n = lines.shape[0]
        if out_pairs is None:
            out_pairs = torch.empty((n * 13 // 4 + 8192, 12), dtype=torch.uint8, device=lines.device)   # 3 pairs per assembled line + padding
        if out_frames is None:
            out_frames = torch.empty((n // 8 + 64, 64), dtype=torch.uint8, device=lines.device)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(lines.device).cuda_stream)
        npairs, nframes = C.c_size_t(0), C.c_size_t(0)
        rc = self.lib.sdv_stitch_frames(self._h, C.c_void_p(lines.data_ptr()), n, C.c_void_p(out_pairs.data_ptr()), out_pairs.shape[0],
                                        C.byref(npairs), C.c_void_p(out_frames.data_ptr()), out_frames.shape[0], C.byref(nframes), sptr)
        self._check(rc)
        return out_pairs[:npairs.value], out_frames[:nframes.value]

    # ---- PCM-1 back half (PCM1DataStitcher) ----
    def default_pcm1_stitch_settings(self) -> Pcm1StitchSettings:
        st = Pcm1StitchSettings()
        self.lib.sdv_default_pcm1_stitch_settings(C.byref(st))
        return st

    def set_pcm1_stitch_settings(self, st: Pcm1StitchSettings):
        self._check(self.lib.sdv_set_pcm1_stitch_settings(self._h, C.byref(st)))

    def pcm1_stitch_frames(self, lines, out_pairs=None, out_frames=None, stream=None):
        """PCM1DataStitcher::doFrameReassemble over a span of the PCM-1 line stream: `lines` is a torch.uint8 CUDA tensor
        (n_records, 32) of sdv_pcm1_line_rec.  Returns (pairs, frames): torch.uint8 CUDA tensors (n_pairs, 12) of sdv_sample_pair
        and (n_frames, 52) of sdv_frame_asm_pcm1."""
        import torch
        assert lines.is_cuda and lines.dtype == torch.uint8 and lines.dim() == 2 and lines.shape[1] == 32 and lines.is_contiguous()
        n = lines.shape[0]
        own_pairs, own_frames = out_pairs is None, out_frames is None
        if out_pairs is None:
            out_pairs = torch.empty((n * 3 + 4096, 12), dtype=torch.uint8, device=lines.device)     # 3 pairs per line + padding lines
        if out_frames is None:
            out_frames = torch.empty((n // 64 + 64, 52), dtype=torch.uint8, device=lines.device)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(lines.device).cuda_stream)
        npairs, nframes = C.c_size_t(0), C.c_size_t(0)
        for attempt in (0, 1):
            rc = self.lib.sdv_pcm1_stitch_frames(self._h, C.c_void_p(lines.data_ptr()), n, C.c_void_p(out_pairs.data_ptr()), out_pairs.shape[0],
                                                 C.byref(npairs), C.c_void_p(out_frames.data_ptr()), out_frames.shape[0], C.byref(nframes), sptr)
            # buffers of our own guess were too small (a frame that lost most of its lines still decodes to 1470 pairs; lines that waited in the engine
            # complete frames): the refused call took nothing and reported the sizes - once more with those
            if rc == -1 and attempt == 0 and (npairs.value > out_pairs.shape[0] or nframes.value > out_frames.shape[0]) and \
                    (own_pairs or npairs.value <= out_pairs.shape[0]) and (own_frames or nframes.value <= out_frames.shape[0]):
                if own_pairs and npairs.value > out_pairs.shape[0]:
                    out_pairs = torch.empty((npairs.value, 12), dtype=torch.uint8, device=lines.device)
                if own_frames and nframes.value > out_frames.shape[0]:
                    out_frames = torch.empty((nframes.value, 52), dtype=torch.uint8, device=lines.device)
                continue
            break
        self._check(rc)
        return out_pairs[:npairs.value], out_frames[:nframes.value]

    def set_pcm1_stitch_block_output(self, blocks):
        """sdv_set_pcm1_stitch_block_output: `blocks` = (cap, 576) uint8 CUDA tensor the following pcm1_stitch_frames calls fill with their
        PCM1DataBlocks, 16 per frame (None: off).  The tensor must stay alive while it is set."""
        self._p1_block_out = blocks
        if blocks is None:
            self._check(self.lib.sdv_set_pcm1_stitch_block_output(self._h, None, 0))
        else:
            assert blocks.is_cuda and blocks.is_contiguous() and blocks.shape[1] == 576
            self._check(self.lib.sdv_set_pcm1_stitch_block_output(self._h, C.c_void_p(blocks.data_ptr()), blocks.shape[0]))

    def pcm1_stitch_block_count(self) -> int:
        return int(self.lib.sdv_pcm1_stitch_block_count(self._h))

    def set_pcm1_stitch_line_output(self, lines):
        """sdv_set_pcm1_stitch_line_output: `lines` = (cap, 16) uint8 CUDA tensor for the sub-lines of the stitcher's queue, 1470 per frame."""
        self._p1_line_out = lines
        if lines is None:
            self._check(self.lib.sdv_set_pcm1_stitch_line_output(self._h, None, 0))
        else:
            assert lines.is_cuda and lines.is_contiguous() and lines.shape[1] == 16
            self._check(self.lib.sdv_set_pcm1_stitch_line_output(self._h, C.c_void_p(lines.data_ptr()), lines.shape[0]))

    def pcm1_stitch_line_count(self) -> int:
        return int(self.lib.sdv_pcm1_stitch_line_count(self._h))

    def set_pcm16x0_stitch_block_output(self, blocks):
        """sdv_set_pcm16x0_stitch_block_output: `blocks` = (cap, 32) uint8 CUDA tensor the following pcm16x0_stitch_frames calls fill with their
        PCM16X0DataBlocks (None: off).  The tensor must stay alive while it is set."""
        self._p16_block_out = blocks
        if blocks is None:
            self._check(self.lib.sdv_set_pcm16x0_stitch_block_output(self._h, None, 0))
        else:
            assert blocks.is_cuda and blocks.is_contiguous() and blocks.shape[1] == 32
            self._check(self.lib.sdv_set_pcm16x0_stitch_block_output(self._h, C.c_void_p(blocks.data_ptr()), blocks.shape[0]))

    def pcm16x0_stitch_block_count(self) -> int:
        return int(self.lib.sdv_pcm16x0_stitch_block_count(self._h))

    def set_pcm16x0_stitch_line_output(self, lines):
        """sdv_set_pcm16x0_stitch_line_output: `lines` = (cap, 36) uint8 CUDA tensor the following pcm16x0_stitch_frames calls fill with the assembled
        sub-lines of their frames as sdv_pcm16x0_bin_rec, an END_FRAME record behind every frame's (None: off) - what vis_render_lines(PCM16X0_LINES)
        draws as the reference's re-assembled window.  The tensor must stay alive while it is set."""
        self._p16_line_out = lines
        if lines is None:
            self._check(self.lib.sdv_set_pcm16x0_stitch_line_output(self._h, None, 0))
        else:
            assert lines.is_cuda and lines.is_contiguous() and lines.shape[1] == 36
            self._check(self.lib.sdv_set_pcm16x0_stitch_line_output(self._h, C.c_void_p(lines.data_ptr()), lines.shape[0]))

    def pcm16x0_stitch_line_count(self) -> int:
        return int(self.lib.sdv_pcm16x0_stitch_line_count(self._h))

    def pcm1_bin_to_line_recs(self, bin_recs, out=None, stream=None):
        """The records pcm1_binarize_frames returns ((n, 40) sdv_pcm1_bin_rec) as the records pcm1_stitch_frames takes ((n, 32) sdv_pcm1_line_rec)."""
        import torch
        _check_out(bin_recs, 40, bin_recs.device, "bin_recs")
        n = bin_recs.shape[0]
        if out is None:
            out = torch.empty((n, 32), dtype=torch.uint8, device=bin_recs.device)
        _check_out(out, 32, bin_recs.device, "out")
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(bin_recs.device).cuda_stream)
        self._check(self.lib.sdv_pcm1_bin_to_line_recs(self._h, C.c_void_p(bin_recs.data_ptr()), n, C.c_void_p(out.data_ptr()), sptr))
        return out[:n]

    def default_pcm16x0_stitch_settings(self) -> Pcm16x0StitchSettings:
        st = Pcm16x0StitchSettings()
        self.lib.sdv_default_pcm16x0_stitch_settings(C.byref(st))
        return st

    def set_pcm16x0_stitch_settings(self, st: Pcm16x0StitchSettings):
        """Applies the settings and starts a fresh PCM16X0DataStitcher (histories and waiting sub-lines are dropped)."""
        self._check(self.lib.sdv_set_pcm16x0_stitch_settings(self._h, C.byref(st)))

    def pcm16x0_stitch_frames(self, lines, out_pairs=None, out_frames=None, stream=None):
        """PCM16X0DataStitcher::doFrameReassemble over a span of the PCM-16x0 sub-line stream: `lines` is a torch.uint8 CUDA tensor
        (n_records, 36) of sdv_pcm16x0_bin_rec (what pcm16x0_binarize_frames returns).  Returns (pairs, frames): torch.uint8 CUDA tensors
        (n_pairs, 12) of sdv_sample_pair and (n_frames, 56) of sdv_frame_asm_pcm16x0."""
        import torch
        assert lines.is_cuda and lines.dtype == torch.uint8 and lines.dim() == 2 and lines.shape[1] == 36 and lines.is_contiguous()
        n = lines.shape[0]
        own_pairs, own_frames = out_pairs is None, out_frames is None
        if out_pairs is None:
            out_pairs = torch.empty((n + n // 8 + 4096, 12), dtype=torch.uint8, device=lines.device)     # a pair per sub-line + padding
        if out_frames is None:
            out_frames = torch.empty((n // 64 + 64, 56), dtype=torch.uint8, device=lines.device)
        _check_out(out_pairs, 12, lines.device, "out_pairs")
        _check_out(out_frames, 56, lines.device, "out_frames")
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(lines.device).cuda_stream)
        npairs, nframes = C.c_size_t(0), C.c_size_t(0)
        for attempt in (0, 1):
            rc = self.lib.sdv_pcm16x0_stitch_frames(self._h, C.c_void_p(lines.data_ptr()), n, C.c_void_p(out_pairs.data_ptr()), out_pairs.shape[0],
                                                    C.byref(npairs), C.c_void_p(out_frames.data_ptr()), out_frames.shape[0], C.byref(nframes), sptr)
            # buffers of our own guess were too small (short fields still decode to 1470 pairs a frame): the refused call took
            # nothing and reported the sizes - once more with those
            if rc == -1 and attempt == 0 and (npairs.value > out_pairs.shape[0] or nframes.value > out_frames.shape[0]) and \
                    (own_pairs or npairs.value <= out_pairs.shape[0]) and (own_frames or nframes.value <= out_frames.shape[0]):
                if npairs.value > out_pairs.shape[0]:
                    out_pairs = torch.empty((npairs.value, 12), dtype=torch.uint8, device=lines.device)
                if nframes.value > out_frames.shape[0]:
                    out_frames = torch.empty((nframes.value, 56), dtype=torch.uint8, device=lines.device)
                continue
            break
        self._check(rc)
        return out_pairs[:npairs.value], out_frames[:nframes.value]

    def binarize_lines(self, luma, presets=None, frame_number: int = 1, first_line: int = 1, line_step: int = 1, doubled: bool = False, out_lines=None, stream=None):
        """Binarizer::processLine with an STC007Line output for every row of `luma` (torch.uint8 CUDA tensor (n_lines, width), rows contiguous) in one
        call.  `presets`: None or a torch.uint8 CUDA tensor (n_lines, 10) of sdv_bin_state - what the caller's Binarizer had been given before each line.
        Returns a torch.uint8 CUDA tensor (n_lines, 48) of sdv_line_rec, the lines as processLine leaves them (no VideoToDigital bookkeeping)."""
        import torch
        assert luma.is_cuda and luma.dtype == torch.uint8 and luma.dim() == 2 and luma.stride(1) == 1
        n, w = luma.shape
        if presets is not None:
            assert presets.is_cuda and presets.dtype == torch.uint8 and presets.shape == (n, 10) and presets.is_contiguous()
        if out_lines is None:
            out_lines = torch.empty((n, 48), dtype=torch.uint8, device=luma.device)
        _check_out(out_lines, 48, luma.device, "out_lines")
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(luma.device).cuda_stream)
        rc = self.lib.sdv_binarize_lines(self._h, C.c_void_p(luma.data_ptr()), luma.stride(0), w, n,
                                         None if presets is None else C.c_void_p(presets.data_ptr()), frame_number, first_line, line_step,
                                         FLAG_DOUBLED if doubled else 0, C.c_void_p(out_lines.data_ptr()), out_lines.shape[0], sptr)
        self._check(rc)
        return out_lines[:n]

    def pcm1_binarize_lines(self, luma, presets=None, frame_number: int = 1, first_line: int = 1, line_step: int = 1, doubled: bool = False,
                            coord_search: bool = True, out_lines=None, stream=None):
        """Binarizer::processLine with a PCM1Line output for every row of `luma` (torch.uint8 CUDA tensor (n_lines, width), rows
        contiguous) in one launch.  `presets`: None or a torch.uint8 CUDA tensor (n_lines, 10) of sdv_bin_state - what the caller's
        Binarizer had been given before each line.  Returns a torch.uint8 CUDA tensor (n_lines, 40) of sdv_pcm1_bin_rec."""
        import torch
        assert luma.is_cuda and luma.dtype == torch.uint8 and luma.dim() == 2 and luma.stride(1) == 1
        n, w = luma.shape
        if presets is not None:
            assert presets.is_cuda and presets.dtype == torch.uint8 and presets.shape == (n, 10) and presets.is_contiguous()
        if out_lines is None:
            out_lines = torch.empty((n, 40), dtype=torch.uint8, device=luma.device)
        _check_out(out_lines, 40, luma.device, "out_lines")
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(luma.device).cuda_stream)
        rc = self.lib.sdv_pcm1_binarize_lines(self._h, C.c_void_p(luma.data_ptr()), luma.stride(0), w, n,
                                              None if presets is None else C.c_void_p(presets.data_ptr()), frame_number, first_line, line_step,
                                              FLAG_DOUBLED if doubled else 0, 1 if coord_search else 0, C.c_void_p(out_lines.data_ptr()),
                                              out_lines.shape[0], sptr)
        self._check(rc)
        return out_lines[:n]

    def pcm16x0_binarize_lines(self, luma, presets=None, frame_number: int = 1, first_line: int = 1, line_step: int = 1, doubled: bool = False,
                               coord_search: bool = True, out_lines=None, with_scan_done: bool = False, stream=None):
        """Binarizer::processLine with a PCM16X0SubLine output, the three passes over every row of `luma` (torch.uint8 CUDA tensor
        (n_lines, width), rows contiguous) in one launch.  `presets`: None or a torch.uint8 CUDA tensor (3 * n_lines, 10) of sdv_bin_state -
        what the caller's Binarizer had been given before each pass.  Returns a torch.uint8 CUDA tensor (3 * n_lines, 36) of
        sdv_pcm16x0_bin_rec (and, with_scan_done, a uint8 tensor of VideoLine::scan_done behind each pass)."""
        import torch
        assert luma.is_cuda and luma.dtype == torch.uint8 and luma.dim() == 2 and luma.stride(1) == 1
        n, w = luma.shape
        if presets is not None:
            assert presets.is_cuda and presets.dtype == torch.uint8 and presets.shape == (3 * n, 10) and presets.is_contiguous()
        if out_lines is None:
            out_lines = torch.empty((3 * n, 36), dtype=torch.uint8, device=luma.device)
        _check_out(out_lines, 36, luma.device, "out_lines")
        scans = torch.zeros((3 * n,), dtype=torch.uint8, device=luma.device) if with_scan_done else None
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(luma.device).cuda_stream)
        rc = self.lib.sdv_pcm16x0_binarize_lines(self._h, C.c_void_p(luma.data_ptr()), luma.stride(0), w, n,
                                                 None if presets is None else C.c_void_p(presets.data_ptr()), frame_number, first_line, line_step,
                                                 FLAG_DOUBLED if doubled else 0, 1 if coord_search else 0, C.c_void_p(out_lines.data_ptr()),
                                                 out_lines.shape[0], None if scans is None else C.c_void_p(scans.data_ptr()), sptr)
        self._check(rc)
        return (out_lines[:3 * n], scans) if with_scan_done else out_lines[:3 * n]

    def pcm1_binarize_frames(self, luma, first_frame_no: int = 1, new_file: bool = False, doubled: bool = False,
                             out_lines=None, out_stats=None, stream=None, end_file: bool = False):
        """VideoToDigital::doBinarize with setPCMType(TYPE_PCM1) over whole frames: luma is a torch.uint8 CUDA tensor
        (n_frames, height, width).  Returns (lines, stats): torch.uint8 CUDA tensors (n_records, 40) of sdv_pcm1_bin_rec and
        (n_frames, 32) of sdv_frame_stats; record order and counts as for binarize_frames."""
        import torch
        assert luma.is_cuda and luma.dtype == torch.uint8 and luma.dim() == 3 and luma.stride(2) == 1
        n, h, w = luma.shape
        nrec = n * (h + 3) + (1 if new_file else 0) + (h + 4 if end_file else 0)
        nst = n + (1 if end_file else 0)
        if out_lines is None:
            out_lines = torch.empty((nrec, 40), dtype=torch.uint8, device=luma.device)
        if out_stats is None:
            out_stats = torch.empty((nst, 32), dtype=torch.uint8, device=luma.device)
        _check_out(out_lines, 40, luma.device, "out_lines")
        _check_out(out_stats, 32, luma.device, "out_stats")
        flags = (FLAG_NEW_FILE if new_file else 0) | (FLAG_DOUBLED if doubled else 0) | (FLAG_END_FILE if end_file else 0)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(luma.device).cuda_stream)
        rc = self.lib.sdv_pcm1_binarize_frames(self._h, C.c_void_p(luma.data_ptr()), luma.stride(1), luma.stride(0), w, h, n,
                                               first_frame_no, flags, C.c_void_p(out_lines.data_ptr()), out_lines.shape[0],
                                               C.c_void_p(out_stats.data_ptr()), out_stats.shape[0], sptr)
        self._check(rc)
        return out_lines[:nrec], out_stats[:nst]

    def pcm16x0_binarize_frames(self, luma, first_frame_no: int = 1, new_file: bool = False, doubled: bool = False,
                                out_lines=None, out_stats=None, stream=None, end_file: bool = False):
        """VideoToDigital::doBinarize with setPCMType(TYPE_PCM16X0) over whole frames: luma is a torch.uint8 CUDA tensor
        (n_frames, height, width).  Returns (lines, stats): torch.uint8 CUDA tensors (n_records, 36) of sdv_pcm16x0_bin_rec - three
        per video line, one per service line - and (n_frames, 32) of sdv_frame_stats."""
        import torch
        assert luma.is_cuda and luma.dtype == torch.uint8 and luma.dim() == 3 and luma.stride(2) == 1
        n, h, w = luma.shape
        flags = (FLAG_NEW_FILE if new_file else 0) | (FLAG_DOUBLED if doubled else 0) | (FLAG_END_FILE if end_file else 0)
        nrec = int(self.lib.sdv_pcm16x0_binarize_records(h, n, flags))
        nst = n + (1 if end_file else 0)
        if out_lines is None:
            out_lines = torch.empty((nrec, 36), dtype=torch.uint8, device=luma.device)
        if out_stats is None:
            out_stats = torch.empty((nst, 32), dtype=torch.uint8, device=luma.device)
        _check_out(out_lines, 36, luma.device, "out_lines")
        _check_out(out_stats, 32, luma.device, "out_stats")
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(luma.device).cuda_stream)
        rc = self.lib.sdv_pcm16x0_binarize_frames(self._h, C.c_void_p(luma.data_ptr()), luma.stride(1), luma.stride(0), w, h, n,
                                                  first_frame_no, flags, C.c_void_p(out_lines.data_ptr()), out_lines.shape[0],
                                                  C.c_void_p(out_stats.data_ptr()), out_stats.shape[0], sptr)
        self._check(rc)
        return out_lines[:nrec], out_stats[:nst]

    # ---- batch replacement of doBinarize ----
    def records_per_frame(self, height: int) -> int:
        return int(self.lib.sdv_records_per_frame(height))

    def binarize_frames(self, luma, first_frame_no: int = 1, new_file: bool = False, doubled: bool = False,
                        out_lines=None, out_stats=None, stream=None, end_file: bool = False):
        """luma: torch.uint8 CUDA tensor (n_frames, height, width), rows contiguous.
        Returns (lines, stats) as torch.uint8 CUDA tensors shaped (n_records, 48) and (n_frames, 32)."""
        import torch
        assert luma.is_cuda and luma.dtype == torch.uint8 and luma.dim() == 3 and luma.stride(2) == 1
        n, h, w = luma.shape
        nrec = n * (h + 3) + (1 if new_file else 0) + (h + 4 if end_file else 0)      # end_file: the filler frame that closes a source
        if out_lines is None:
            out_lines = torch.empty((nrec, 48), dtype=torch.uint8, device=luma.device)
        nst = n + (1 if end_file else 0)
        if out_stats is None:
            out_stats = torch.empty((nst, 32), dtype=torch.uint8, device=luma.device)
        _check_out(out_lines, 48, luma.device, "out_lines")
        _check_out(out_stats, 32, luma.device, "out_stats")
        flags = (FLAG_NEW_FILE if new_file else 0) | (FLAG_DOUBLED if doubled else 0) | (FLAG_END_FILE if end_file else 0)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(luma.device).cuda_stream)
        # the library checks the capacities against what the call will write (SDV_ERR_BAD_ARG when a buffer is too small)
        rc = self.lib.sdv_binarize_frames(self._h, C.c_void_p(luma.data_ptr()), luma.stride(1), luma.stride(0), w, h, n,
                                          first_frame_no, flags, C.c_void_p(out_lines.data_ptr()), out_lines.shape[0],
                                          C.c_void_p(out_stats.data_ptr()), out_stats.shape[0], sptr)
        self._check(rc)
        return out_lines[:nrec], out_stats[:nst]

    # ---- AudioProcessor / SamplesToWAV -----------------------------------------------------------
    def set_audio_masking(self, drop_mode: int):
        """AudioProcessor::setMasking (audioprocessor.cpp:1532-1574): SDV_DROP_* 0..6."""
        self._check(self.lib.sdv_set_audio_masking(self._h, int(drop_mode)))

    def reset_audio(self):
        """A freshly constructed AudioProcessor: nothing waits, the sample index is 0; the masking mode stays."""
        self._check(self.lib.sdv_reset_audio(self._h))

    def audio_pending(self) -> int:
        return int(self.lib.sdv_audio_pending(self._h))

    def audio_stalled(self) -> bool:
        """The worker takes no more input: its window is full and the first pairs can never leave (audioprocessor.cpp:108)."""
        return bool(self.lib.sdv_audio_stalled(self._h))

    def audio_next_index(self) -> int:
        return int(self.lib.sdv_audio_next_index(self._h))

    def audio_process(self, pairs, stop=False, out_pairs=None, out_purges=None, stream=None):
        """AudioProcessor::processAudio over one burst of the PCMSamplePair stream: `pairs` is a torch.uint8 CUDA tensor (n, 12) of
        sdv_sample_pair (what the stitch entry points return, tags included).  Returns (out, purges, masked): torch.uint8 CUDA tensors
        (n_out, 12) of sdv_sample_pair and (n_purges, 16) of sdv_audio_purge, and the sum of the guiAddMask reports."""
        import torch
        _check_out(pairs, 12, pairs.device, "pairs")
        n = pairs.shape[0]
        if out_pairs is None:
            out_pairs = torch.empty((n + 1024, 12), dtype=torch.uint8, device=pairs.device)
        if out_purges is None:
            out_purges = torch.empty((256, 16), dtype=torch.uint8, device=pairs.device)
        _check_out(out_pairs, 12, pairs.device, "out_pairs")
        _check_out(out_purges, 16, pairs.device, "out_purges")
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(pairs.device).cuda_stream)
        n_out, n_pur, masked = C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
        for attempt in range(2):
            rc = self.lib.sdv_audio_process(self._h, C.c_void_p(pairs.data_ptr()) if n else None, n, 1 if stop else 0, C.c_void_p(out_pairs.data_ptr()),
                                            out_pairs.shape[0], C.byref(n_out), C.c_void_p(out_purges.data_ptr()), out_purges.shape[0], C.byref(n_pur),
                                            C.byref(masked), sptr)
            if rc == -1 and attempt == 0 and (n_out.value > out_pairs.shape[0] or n_pur.value > out_purges.shape[0]):
                # a refused call takes nothing: come again with the sizes it reported
                if n_out.value > out_pairs.shape[0]:
                    out_pairs = torch.empty((n_out.value, 12), dtype=torch.uint8, device=pairs.device)
                if n_pur.value > out_purges.shape[0]:
                    out_purges = torch.empty((n_pur.value, 16), dtype=torch.uint8, device=pairs.device)
                continue
            break
        self._check(rc)
        return out_pairs[:n_out.value], out_purges[:n_pur.value], int(masked.value)

    def wav_pack(self, pairs, out=None, stream=None):
        """SamplesToWAV::saveAudio for every pair: (n, 12) sdv_sample_pair -> (n, 2) int16 on the device."""
        import torch
        _check_out(pairs, 12, pairs.device, "pairs")
        n = pairs.shape[0]
        if out is None:
            out = torch.empty((n, 2), dtype=torch.int16, device=pairs.device)
        assert out.is_cuda and out.dtype == torch.int16 and out.is_contiguous() and out.numel() >= 2 * n
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(pairs.device).cuda_stream)
        self._check(self.lib.sdv_wav_pack(self._h, C.c_void_p(pairs.data_ptr()) if n else None, n, C.c_void_p(out.data_ptr()), sptr))
        return out[:n]

    def wav_header(self, n_pairs: int, last_sample_rate: int) -> bytes:
        hdr = C.create_string_buffer(44)
        self.lib.sdv_wav_header(hdr, int(n_pairs), int(last_sample_rate))
        return hdr.raw

    def wav_files(self, out_pairs, purges):
        """The files SamplesToWAV leaves for an output stream of audio_process (one per NEW_FILE purge that is followed by at least one
        pair): {number of the NEW_FILE tag: bytes}.  `purges` as a numpy array of sdv_audio_purge records (or the uint8 tensor)."""
        import numpy as np
        if hasattr(purges, "cpu"):
            purges = purges.cpu().numpy().view(AUDIO_PURGE_DTYPE).reshape(-1)
        files, k, n = {}, 0, out_pairs.shape[0]
        for i, p in enumerate(purges):
            if p["kind"] != 1:
                continue
            a = int(p["first_pair"])
            b = int(purges[i + 1]["first_pair"]) if i + 1 < len(purges) else n
            if b > a:
                pcm = self.wav_pack(out_pairs[a:b])
                rate = int(out_pairs[b - 1].cpu().numpy().view(PAIR_DTYPE)["sample_rate"][0])
                files[k] = self.wav_header(b - a, rate) + pcm.cpu().numpy().tobytes()
            k += 1
        return files

    # ---- the workers back to back (sdv_decode_frames) ---------------------------------------------
    def decode_frames(self, pcm_type: int, luma, first_frame_no: int = 1, new_file: bool = False, doubled: bool = False, end_file: bool = False,
                      with_audio: bool = False, audio_stop: bool = False, stream=None, out_pairs=None, out_frames=None, out_stats=None):
        """Video frames -> PCMSamplePair in one call: the format's VideoToDigital worker, its data stitcher and - with_audio - the AudioProcessor;
        the line records (and the raw pair stream) stay inside the engine.  luma: torch.uint8 CUDA tensor (n_frames, height, width).
        out_pairs / out_frames / out_stats: the caller's buffers ((>= (n + 2) * 1800 + 8192, 12), (>= n + 16, 64 / 52 / 56), (>= n (+ 1), 32) uint8),
        allocated per call when not given.
        Returns (pairs (n, 12), frame descriptors (n, 64 / 52 / 56), frame stats (n, 32)) and, with_audio, also (purges (n, 16), masked)."""
        import torch
        assert luma.is_cuda and luma.dtype == torch.uint8 and luma.dim() == 3 and luma.stride(2) == 1
        n, h, w = luma.shape
        flags = (FLAG_NEW_FILE if new_file else 0) | (FLAG_DOUBLED if doubled else 0) | (FLAG_END_FILE if end_file else 0)
        fr_bytes = {PCM_STC007: 64, 0: 52, 1: 56}[pcm_type]
        nst = n + (1 if end_file else 0)
        if out_pairs is None:
            out_pairs = torch.empty(((n + 2) * 1800 + 8192, 12), dtype=torch.uint8, device=luma.device)
        if out_frames is None:
            out_frames = torch.empty((n + 16, fr_bytes), dtype=torch.uint8, device=luma.device)
        if out_stats is None:
            out_stats = torch.empty((nst, 32), dtype=torch.uint8, device=luma.device)
        for t, cols in ((out_pairs, 12), (out_frames, fr_bytes), (out_stats, 32)):
            assert t.is_cuda and t.dtype == torch.uint8 and t.dim() == 2 and t.shape[1] == cols and t.is_contiguous()
        assert out_stats.shape[0] >= nst
        out_purges = torch.empty((16, 16), dtype=torch.uint8, device=luma.device)
        sptr = C.c_void_p(stream.cuda_stream) if stream is not None else C.c_void_p(torch.cuda.current_stream(luma.device).cuda_stream)
        n_pairs, n_fr, n_pur, masked = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
        rc = self.lib.sdv_decode_frames(self._h, pcm_type, C.c_void_p(luma.data_ptr()), luma.stride(1), luma.stride(0), w, h, n, first_frame_no, flags,
                                        C.c_void_p(out_pairs.data_ptr()), out_pairs.shape[0], C.byref(n_pairs), C.c_void_p(out_frames.data_ptr()), out_frames.shape[0],
                                        C.byref(n_fr), C.c_void_p(out_stats.data_ptr()), out_stats.shape[0], 1 if with_audio else 0, 1 if audio_stop else 0,
                                        C.c_void_p(out_purges.data_ptr()), out_purges.shape[0], C.byref(n_pur), C.byref(masked), sptr)
        self._check(rc)
        if with_audio:
            return out_pairs[:n_pairs.value], out_frames[:n_fr.value], out_stats[:nst], out_purges[:n_pur.value], int(masked.value)
        return out_pairs[:n_pairs.value], out_frames[:n_fr.value], out_stats[:nst]
