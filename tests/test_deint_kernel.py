"""The HIP deinterleave + P/Q kernel: on the CPU through the SIMT emulator build, on the GPU through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

import deint_api as da
from sdvpcmdecoder_amd import synth

COMBOS = [(1, 1, 0), (1, 1, 1), (1, 0, 0), (1, 0, 1), (0, 0, 0)]


def damaged_streams(seed, n_blocks):
    rng = np.random.default_rng(seed)
    n = 113 + n_blocks - 1
    audio14 = rng.integers(0, 1 << 14, size=(n, 6), dtype=np.uint32)
    audio16 = rng.integers(0, 1 << 16, size=(n, 6), dtype=np.uint32)
    audio14[50:70] = 0
    out = []
    for w9 in (synth.interleave_stream(audio14), synth.interleave_stream_f1(audio16)):
        for (p_bad, p_sneak, p_cwd) in ((0.0, 0.0, 0.0), (0.05, 0.01, 0.02), (0.3, 0.02, 0.05)):
            out.append(da.make_lines(w9, rng=rng, p_bad=p_bad, p_corrupt_valid=p_sneak, p_cwd=p_cwd))
    return out


def all_settings():
    for res_mode in (0, 1, 2, 3):
        for (p, q, cwd) in COMBOS:
            for force in (1, 0):
                for ign in (0, 1):
                    yield da.settings(res_mode=res_mode, ignore_crc=ign, force=force, p=p, q=q, cwd=cwd)


def test_emu_deint_matches_oracle(emu_lib, oracle_lib):
    eng = C.c_void_p(emu_lib.sdv_engine_create(0))
    nb = 150
    for lines in damaged_streams(3, nb):
        for st in all_settings():
            rc_o, want = da.run_cpu(oracle_lib, "orc_", lines, st, nb)
            got = np.zeros(nb, dtype=da.BLOCK_DTYPE)
            rc = emu_lib.sdv_deinterleave_blocks(eng, lines.ctypes.data, len(lines), C.byref(st), got.ctypes.data, nb, None)
            assert rc == 0 and rc_o == 3
            bad = np.nonzero(got.view(np.uint8).reshape(nb, 72) != want.view(np.uint8).reshape(nb, 72))[0]
            assert len(bad) == 0, (st.res_mode, st.en_p_code, st.en_q_code, st.en_cwd, got[bad[0]], want[bad[0]])
    # DI_RET_NO_DATA / null pointers
    got = np.zeros(4, dtype=da.BLOCK_DTYPE)
    st = da.settings()
    assert emu_lib.sdv_deinterleave_blocks(eng, lines.ctypes.data, 115, C.byref(st), got.ctypes.data, 4, None) == 18
    assert emu_lib.sdv_deinterleave_blocks(eng, None, 200, C.byref(st), got.ctypes.data, 4, None) == 16
    assert emu_lib.sdv_deinterleave_blocks(eng, lines.ctypes.data, 200, C.byref(st), None, 4, None) == 17
    emu_lib.sdv_engine_destroy(eng)


@pytest.mark.gpu
def test_hip_deint_matches_oracle(oracle_lib):
    import torch
    from sdvpcmdecoder_amd import Engine, DeintSettings, BLOCK_DTYPE
    eng = Engine(0)
    nb = 4000
    for lines in damaged_streams(5, nb):
        d = torch.from_numpy(lines.view(np.uint8).reshape(len(lines), 24)).to("cuda:0")
        for st in all_settings():
            rc_o, want = da.run_cpu(oracle_lib, "orc_", lines, st, nb)
            est = DeintSettings(st.res_mode, st.ignore_crc, st.force_ecc_check, st.en_p_code, st.en_q_code, st.en_cwd)
            out = eng.deinterleave_blocks(d, est, nb)
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            bad = np.nonzero(got != want.view(np.uint8).reshape(nb, 72))[0]
            assert len(bad) == 0, (st.res_mode, st.en_p_code, st.en_q_code, st.en_cwd, got[bad[0]].view(BLOCK_DTYPE), want[bad[0]])
    eng.close()


@pytest.mark.gpu
def test_hip_deint_ecc_property_full_size():
    """Size-independent property on 1M blocks: a stream whose every line lost its CRC in one of two interleaved
    patterns (<= 2 bad words per block) must be restored to the generator's audio exactly; pcmtester.cpp:296-365."""
    import torch
    from sdvpcmdecoder_amd import Engine
    rng = np.random.default_rng(9)
    nb = 1_000_000
    n = nb + 112
    audio = rng.integers(0, 1 << 14, size=(n, 6), dtype=np.uint32)
    w9 = synth.interleave_stream(audio)
    lines = da.make_lines(w9, rng=rng)
    # kill the lines at 0, 32 and 40 modulo 256: a block reads lines s+16k (k = 0..7), so it can meet the pair
    # (0, 32) together (two damaged words -> Q correction) or a single one (P correction), never three
    m = np.arange(n) % 256
    kill = (m == 0) | (m == 32) | (m == 40)
    lines["word_crc_ok"][kill] = 0
    lines["words"][kill] ^= rng.integers(1, 1 << 14, size=(int(kill.sum()), 8), dtype=np.uint16)
    eng = Engine(0)
    st = eng.default_deint_settings()
    st.res_mode = 0
    d = torch.from_numpy(lines.view(np.uint8).reshape(n, 24)).to("cuda:0")
    out = eng.deinterleave_blocks(d, st, nb).cpu().numpy().view(da.BLOCK_DTYPE).reshape(-1)
    eng.close()
    # blocks start at line s; block s holds audio[s] once the stream is "full" (s >= 0 here by construction)
    assert ((out["word_valid"] & 0x3F) == 0x3F).all()
    assert (out["words"][:, :6] == audio[:nb].astype(np.uint16)).all()
    assert (out["audio_state"] != 3).all()
