"""The C-ABI library loads and exports every symbol include/sdvpcm.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sdvpcm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdv_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for s in ("sdv_engine_create", "sdv_engine_destroy", "sdv_binarize_frames", "sdv_set_mode", "sdv_set_bin_preset",
              "sdv_last_error"):
        assert s in syms


def test_product_library_exports_all_symbols():
    from sdvpcmdecoder_amd import build as b
    path = b.build_hip()
    lib = C.CDLL(path)
    for s in declared_symbols():
        assert hasattr(lib, s), f"libsdvpcm_hip.so does not export {s}"
    lib.sdv_abi_version.restype = C.c_int
    hdr = open(os.path.join(ROOT, "include", "sdvpcm.h")).read()
    assert lib.sdv_abi_version() == int(re.search(r"#define SDV_ABI_VERSION (\d+)", hdr).group(1))


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sdvpcmdecoder_amd import Engine
    with pytest.raises(RuntimeError, match="no HIP device|sdv_engine_create failed"):
        Engine(0)


def test_product_does_not_reference_oracle():
    """The product sources must not include or link anything under oracle/ or the emulator."""
    pkg = os.path.join(ROOT, "sdvpcmdecoder_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")) and f != "build.py":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liborc" not in text and "oracle/" not in text.replace("(see oracle/", ""), f
                assert "hip_emu.h" not in text or f == "stc007_device.h" or f == "engine.inc", f


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 and as C++ on its own, and the PODs must have the documented sizes."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include "sdvpcm.h"\n'
                   '#define CHECK(t, n) typedef char check_##t[(sizeof(t) == (n)) ? 1 : -1]\n'
                   'CHECK(sdv_line_rec, 48); CHECK(sdv_frame_stats, 32); CHECK(sdv_v2d_state, 120); CHECK(sdv_deint_line, 24);\n'
                   'CHECK(sdv_block_rec, 72); CHECK(sdv_sample_pair, 12); CHECK(sdv_frame_asm, 64); CHECK(sdv_stitch_settings, 16); CHECK(sdv_audio_purge, 16);\n'
                   'CHECK(sdv_pcm1_block_rec, 576); CHECK(sdv_pcm1_asm_line_rec, 16); CHECK(sdv_pcm16x0_block_rec, 32); CHECK(sdv_asm_line_rec, 32);\n'
                   'int main(void) { sdv_engine *e = sdv_engine_create(0); sdv_engine_destroy(e); return 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])


# ---- the boundary driven from plain C++ host code (examples/decode_tape.cpp): no Python, no torch in the process that decodes ----
def test_cpp_example_builds():
    from sdvpcmdecoder_amd import build as b
    assert os.path.exists(b.build_example())


@pytest.mark.gpu
def test_cpp_host_program_matches_reference_golden(tmp_path):
    """decode_tape stc007: luma file -> sdv_binarize_frames(NEW_FILE | END_FILE) -> sdv_stitch_frames, written by a C++ program that
    links libsdvpcm_hip.so; compared with the real reference's output for the same file (tests/golden/e2e_ntsc_file.npz)."""
    import subprocess
    import numpy as np
    from sdvpcmdecoder_amd import build as b
    import test_stitch_kernel as tsk
    exe = b.build_example()
    luma, z, want_p, want_f = tsk._e2e_fixture()
    n, h, w = luma.shape
    (tmp_path / "luma.raw").write_bytes(np.ascontiguousarray(luma).tobytes())
    out = subprocess.run([exe, "stc007", str(tmp_path / "luma.raw"), str(w), str(h), str(n), str(tmp_path / "pairs.out"), str(tmp_path / "frames.out")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert (tmp_path / "pairs.out").read_bytes() == want_p.tobytes()
    assert (tmp_path / "frames.out").read_bytes() == want_f.tobytes()


@pytest.mark.gpu
def test_cpp_host_program_pcm1_matches_reference_golden(tmp_path):
    import subprocess
    import numpy as np
    import pcm1_api as p1
    from sdvpcmdecoder_amd import build as b
    exe = b.build_example()
    z = np.load(os.path.join(ROOT, "tests", "golden", "pcm1_file_marks.npz"))
    recs, st = p1.make_input("file_marks")
    (tmp_path / "lines.raw").write_bytes(recs.tobytes())
    out = subprocess.run([exe, "pcm1", str(tmp_path / "lines.raw"), str(tmp_path / "pairs.out"), str(tmp_path / "frames.out")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert (tmp_path / "pairs.out").read_bytes() == np.ascontiguousarray(z["pairs"]).tobytes()
    assert (tmp_path / "frames.out").read_bytes() == np.ascontiguousarray(z["frames"]).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["si", "ei"])
def test_cpp_host_program_pcm16x0_matches_reference_golden(fmt, tmp_path):
    """decode_tape pcm16x0: luma file -> sdv_pcm16x0_binarize_frames(NEW_FILE | END_FILE) -> sdv_pcm16x0_stitch_frames from plain C++;
    compared with the real reference's two workers on the same file (tests/golden/e2e_pcm16x0_*.npz)."""
    import subprocess
    import numpy as np
    from sdvpcmdecoder_amd import build as b
    import test_pcm16 as t16
    exe = b.build_example()
    luma, audio, z, want_p, want_f = t16._e2e_fixture(fmt == "ei")
    n, h, w = luma.shape
    (tmp_path / "luma.raw").write_bytes(np.ascontiguousarray(luma).tobytes())
    out = subprocess.run([exe, "pcm16x0", str(tmp_path / "luma.raw"), str(w), str(h), str(n), fmt, str(tmp_path / "pairs.out"), str(tmp_path / "frames.out")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert (tmp_path / "pairs.out").read_bytes() == want_p.tobytes()
    assert (tmp_path / "frames.out").read_bytes() == want_f.tobytes()


@pytest.mark.gpu
def test_cpp_host_program_writes_the_reference_wav(tmp_path):
    """decode_tape wav: luma file -> binarize -> stitch -> sdv_audio_process -> sdv_wav_pack / sdv_wav_header from plain C++; the file equals the
    one the real reference's SamplesToWAV wrote at the end of its own chain (tests/golden/e2e_ntsc_file_audio.npz)."""
    import subprocess
    import numpy as np
    from sdvpcmdecoder_amd import build as b
    import test_stitch_kernel as tsk
    import test_audio as ta
    exe = b.build_example()
    luma, _, _, _ = tsk._e2e_fixture()
    z, _, wavs = ta._e2e_audio_fixture()
    n, h, w = luma.shape
    (tmp_path / "luma.raw").write_bytes(np.ascontiguousarray(luma).tobytes())
    out = subprocess.run([exe, "wav", str(tmp_path / "luma.raw"), str(w), str(h), str(n), str(tmp_path / "out.wav")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert (tmp_path / "out.wav").read_bytes() == wavs[0]
