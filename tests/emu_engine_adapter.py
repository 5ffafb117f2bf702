"""The emulator build (tests/emu/libsdvpcm_emu.so, host pointers) behind the method names of sdvpcmdecoder_amd.Engine, so that
host-side orchestration written for the product (sdvpcmdecoder_amd/sharded.py) can be exercised on CPU."""
import ctypes as C

import numpy as np

import engine_api as ea
import libs
import stitch_api as sa


class EmuEngine:
    def __init__(self, lib, mode=2):
        self.lib = ea.bind(lib)
        self.lib.sdv_stitch_state_size.restype = C.c_size_t
        self.lib.sdv_get_stitch_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.lib.sdv_set_stitch_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.lib.sdv_saturate_stitch_stats.argtypes = [C.c_void_p]
        self.h = C.c_void_p(self.lib.sdv_engine_create(0))
        self.lib.sdv_set_mode(self.h, mode)

    def close(self):
        self.lib.sdv_engine_destroy(self.h)

    def reset_stream(self):
        assert self.lib.sdv_reset_stream(self.h) == 0

    def reset_stitcher(self):
        assert self.lib.sdv_reset_stitcher(self.h) == 0

    def binarize_frames(self, luma, first_frame_no=1, new_file=False, end_file=False):
        rc, recs, stats = ea.emu_binarize(self.lib, self.h, np.ascontiguousarray(luma), first_frame_no=first_frame_no,
                                          flags=(1 if new_file else 0) | (4 if end_file else 0))
        assert rc == 0, self.lib.sdv_last_error(self.h)
        return recs, stats

    def stitch_frames(self, recs):
        rc, pairs, frames = ea.emu_stitch(self.lib, self.h, recs, None, pair_cap=len(recs) * 4 + 8192, frame_cap=len(recs) // 8 + 64)
        assert rc == 0, self.lib.sdv_last_error(self.h)
        return pairs.copy(), frames.copy()

    def decode_frames(self, pcm_type, luma, first_frame_no=1, new_file=False, end_file=False):
        """sdv_decode_frames on host memory: (pairs, frame descriptors, frame stats) like Engine.decode_frames."""
        import stitch_api as sa
        lib = self.lib
        lib.sdv_decode_frames.restype = C.c_int
        lib.sdv_decode_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t,
                                          C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_void_p]
        assert pcm_type == 2, "the adapter's fused call: STC-007"
        luma = np.ascontiguousarray(luma)
        n, hgt, w = luma.shape
        cap = (n + 2) * 1800 + 8192
        pairs = np.zeros(cap, dtype=sa.PAIR_DTYPE)
        frames = np.zeros(n + 16, dtype=sa.FRASM_DTYPE)
        stats = np.zeros(n + 1, dtype=ea.STATS_DTYPE)
        npairs, nfr, npur, nm = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_uint64(0)
        rc = lib.sdv_decode_frames(self.h, pcm_type, luma.ctypes.data, w, w * hgt, w, hgt, n, first_frame_no, (1 if new_file else 0) | (4 if end_file else 0),
                                   pairs.ctypes.data, cap, C.byref(npairs), frames.ctypes.data, len(frames), C.byref(nfr), stats.ctypes.data, len(stats), 0, 0,
                                   None, 0, C.byref(npur), C.byref(nm), None)
        assert rc == 0, lib.sdv_last_error(self.h)
        return pairs[:npairs.value].copy(), frames[:nfr.value].copy(), stats

    def set_stitch_settings(self, st):
        assert self.lib.sdv_set_stitch_settings(self.h, C.byref(st)) == 0

    def get_chain_state(self):
        buf = C.create_string_buffer(120)
        assert self.lib.sdv_get_chain_state(self.h, buf) == 0
        return buf.raw

    def set_chain_state(self, b):
        assert self.lib.sdv_set_chain_state(self.h, C.create_string_buffer(b, 120)) == 0

    def get_stitch_state(self):
        n = self.lib.sdv_stitch_state_size()
        buf = C.create_string_buffer(n)
        assert self.lib.sdv_get_stitch_state(self.h, buf, n) == 0
        return buf.raw

    def set_stitch_state(self, b):
        assert self.lib.sdv_set_stitch_state(self.h, C.create_string_buffer(b, len(b)), len(b)) == 0

    def saturate_stitch_stats(self):
        assert self.lib.sdv_saturate_stitch_stats(self.h) == 0

    # ---- PCM-1 / PCM-16x0 (ShardedPcmDecoder) ----------------------------------------------------------------------------------
    def _frames(self, pf, luma, first_frame_no, new_file, end_file):
        st = {}
        if new_file:
            st["new_file"] = True
        if end_file:
            st["end_file"] = True
        rc, recs, stats = pf.run_engine(self.lib, self.h, np.ascontiguousarray(luma), None, st, first_frame_no=first_frame_no, configure=False)
        assert rc == 0, self.lib.sdv_last_error(self.h)
        return recs, stats

    def pcm1_binarize_frames(self, luma, first_frame_no=1, new_file=False, end_file=False):
        import pcm1_frames_api as pf
        return self._frames(pf, luma, first_frame_no, new_file, end_file)

    def pcm16x0_binarize_frames(self, luma, first_frame_no=1, new_file=False, end_file=False):
        import pcm16_frames_api as pf
        return self._frames(pf, luma, first_frame_no, new_file, end_file)

    def pcm1_bin_to_line_recs(self, recs):
        import pcm1_api as p1
        recs = np.ascontiguousarray(recs)
        out = np.zeros(len(recs), dtype=p1.LINE1_DTYPE)
        self.lib.sdv_pcm1_bin_to_line_recs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        assert self.lib.sdv_pcm1_bin_to_line_recs(self.h, recs.ctypes.data, len(recs), out.ctypes.data, None) == 0
        return out

    def set_pcm1_stitch_settings(self, st):
        assert self.lib.sdv_set_pcm1_stitch_settings(self.h, C.byref(st)) == 0

    def pcm1_stitch_frames(self, recs):
        rc, pairs, frames = ea.emu_pcm1_stitch(self.lib, self.h, recs)
        assert rc == 0, self.lib.sdv_last_error(self.h)
        return pairs.copy(), frames.copy()

    def set_pcm16x0_stitch_settings(self, st):
        self.lib.sdv_set_pcm16x0_stitch_settings.argtypes = [C.c_void_p, C.c_void_p]
        assert self.lib.sdv_set_pcm16x0_stitch_settings(self.h, C.byref(st)) == 0

    def pcm16x0_stitch_frames(self, recs):
        rc, pairs, frames = ea.emu_pcm16_stitch(self.lib, self.h, recs)
        assert rc == 0, self.lib.sdv_last_error(self.h)
        return pairs.copy(), frames.copy()

    def _blob(self, size_fn, get_fn):
        size_fn.restype = C.c_size_t
        n = size_fn()
        buf = C.create_string_buffer(n)
        get_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        assert get_fn(self.h, buf, n) == 0
        return buf.raw

    def get_pcm16x0_chain_state(self):
        return self._blob(self.lib.sdv_pcm16x0_chain_state_size, self.lib.sdv_get_pcm16x0_chain_state)

    def set_pcm16x0_chain_state(self, b):
        self.lib.sdv_set_pcm16x0_chain_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        assert self.lib.sdv_set_pcm16x0_chain_state(self.h, C.create_string_buffer(b, len(b)), len(b)) == 0

    def get_pcm16x0_stitch_state(self):
        return self._blob(self.lib.sdv_pcm16x0_stitch_state_size, self.lib.sdv_get_pcm16x0_stitch_state)

    def set_pcm16x0_stitch_state(self, b):
        self.lib.sdv_set_pcm16x0_stitch_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        assert self.lib.sdv_set_pcm16x0_stitch_state(self.h, C.create_string_buffer(b, len(b)), len(b)) == 0

    def saturate_pcm16x0_stitch_stats(self):
        self.lib.sdv_saturate_pcm16x0_stitch_stats.argtypes = [C.c_void_p]
        assert self.lib.sdv_saturate_pcm16x0_stitch_stats(self.h) == 0
