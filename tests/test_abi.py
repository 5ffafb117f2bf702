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
    assert lib.sdv_abi_version() == 1


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
