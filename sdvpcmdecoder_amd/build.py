"""Build recipes. `build_all()` is what __graft_entry__.build() runs.

Product:  sdvpcmdecoder_amd/libsdvpcm_hip.so   hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
Test infrastructure (never loaded by the product):
          oracle/liborc.so                     gcc, plain C restatement of the reference path
          oracle/_ref/libsdvref.so             the real reference, only where /root/reference exists
          tests/emu/libsdvpcm_emu.so           the kernel source under the CPU SIMT emulator
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
HIP_LIB = os.path.join(PKG, "libsdvpcm_hip.so")


# the sources a kernel is compiled from and launched by (its device header, the engine that configures the launch, the C-ABI)
KERNEL_SOURCES = {
    "sdv_k_stc007_frames": ("stc007_device.h", "stc007_sweep_device.h", "engine.inc"),
    "sdv_k_stc007_sweep": ("stc007_device.h", "stc007_sweep_device.h", "engine.inc"),
    "sdv_k_hist_carry": ("stc007_device.h", "engine.inc"),
    "sdv_k_pcm1_lines": ("stc007_device.h", "pcm1_bin_device.h", "pcm1_engine.inc"),
    "sdv_k_pcm1_frames": ("pcm1_stitch_device.h", "pcm1_engine.inc"),
    "sdv_k_pcm1_frames_bin": ("stc007_device.h", "pcm1_bin_device.h", "pcm1_frames_device.h", "pcm1_frames_engine.inc"),
    "sdv_k_pcm1_frames_lean": ("stc007_device.h", "pcm1_bin_device.h", "pcm1_frames_device.h", "pcm1_frames_engine.inc"),
    "sdv_k_pcm1_prescan": ("stc007_device.h", "pcm1_bin_device.h", "pcm1_frames_device.h", "pcm1_frames_engine.inc"),
    "sdv_k_pcm16_frames_lean": ("stc007_device.h", "pcm1_bin_device.h", "pcm16_bin_device.h", "pcm16_frames_device.h", "pcm16_frames_engine.inc"),
    "sdv_k_pcm16_prescan": ("stc007_device.h", "pcm1_bin_device.h", "pcm16_bin_device.h", "pcm16_frames_device.h", "pcm16_frames_engine.inc"),
    "sdv_k_ap_": ("audio_device.h", "audio_engine.inc"),
    "sdv_k_pcm16_frames_bin": ("stc007_device.h", "pcm1_bin_device.h", "pcm16_bin_device.h", "pcm16_frames_device.h", "pcm16_frames_engine.inc"),
}


def source_hash(kernel=None):
    """Identity of the HIP library's sources (csrc/ + the C-ABI header), or of the sources one kernel depends on (KERNEL_SOURCES,
    longest matching prefix of its name): profiles/ records it next to the counters they hold, and bench.py only quotes a committed
    profile for the build that it was measured on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(PKG, "csrc")
    names = sorted(os.listdir(csrc))
    if kernel:
        keys = [k for k in KERNEL_SOURCES if kernel.startswith(k)]
        if keys:
            names = sorted(KERNEL_SOURCES[max(keys, key=len)])
    for f in names + [os.path.join(ROOT, "include", "sdvpcm.h")]:
        path = f if os.path.isabs(f) else os.path.join(csrc, f)
        if os.path.isfile(path) and path.endswith((".h", ".hip", ".inc")):
            h.update(os.path.basename(path).encode() + b"\0" + open(path, "rb").read())
    return h.hexdigest()[:16]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


HIP_DEV_LIB = os.path.join(PKG, "libsdvpcm_hip_dev.so")


def build_hip_dev(force=False):
    """TEST ONLY: the developer build of the same sources (-DSDV_DEV_AIDS: the scheduler's off-switches and traces can be set through the environment).
    The product never loads it; tests/test_decode_frames.py runs the fused entry's defensive ways through it on the GPU, in a process of its own."""
    csrc = os.path.join(PKG, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip", ".inc"))] + [os.path.join(ROOT, "include", "sdvpcm.h")]
    if not force and not _newer(HIP_DEV_LIB, srcs):
        return HIP_DEV_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "-DSDV_DEV_AIDS", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           "-o", HIP_DEV_LIB, os.path.join(csrc, "sdvpcm_hip.hip")], cwd=csrc)
    return HIP_DEV_LIB


def build_hip(force=False):
    csrc = os.path.join(PKG, "csrc")
    srcs = [os.path.join(csrc, f) for f in ("sdvpcm_hip.hip", "stc007_device.h", "stc007_sweep_device.h", "stc007_deint_device.h", "stc007_stitch_device.h", "engine.inc",
                                              "stitch_engine.inc", "pcm1_stitch_device.h", "pcm1_bin_device.h", "pcm1_engine.inc", "pcm1_frames_device.h", "pcm1_frames_engine.inc", "pcm16_bin_device.h", "pcm16_frames_device.h", "pcm16_frames_engine.inc", "pcm16_engine.inc", "pcm16_stitch_device.h", "audio_device.h", "audio_engine.inc", "vis_device.h", "vis_engine.inc")] + \
           [os.path.join(ROOT, "include", "sdvpcm.h")]
    if not force and not _newer(HIP_LIB, srcs):
        return HIP_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
           "-o", HIP_LIB, os.path.join(csrc, "sdvpcm_hip.hip")]
    if os.environ.get("SDVPCM_DEV_AIDS") == "1":        # a developer build: the scheduler trace etc. can be switched on through the environment
        cmd.insert(1, "-DSDV_DEV_AIDS")
        for d in os.environ.get("SDVPCM_DEV_DEFINES", "").split():       # experiments: extra -D switches of a developer build
            cmd.insert(1, "-D" + d)
    subprocess.check_call(cmd, cwd=csrc)
    return HIP_LIB


def build_example(force=False):
    """examples/decode_tape: the C-ABI driven from plain C++ host code (what a maintainer's binding looks like)."""
    src = os.path.join(ROOT, "examples", "decode_tape.cpp")
    out = os.path.join(ROOT, "examples", "decode_tape")
    if not force and not _newer(out, [src, os.path.join(ROOT, "include", "sdvpcm.h"), HIP_LIB]):
        return out
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                           "-I" + os.path.join(ROOT, "include"), src, "-L" + PKG, "-lsdvpcm_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                           "-Wl,-rpath,$ORIGIN/../sdvpcmdecoder_amd", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", out])
    return out


def build_example_sharded(force=False):
    """examples/decode_tape_sharded: one tape over several GPUs from plain C++ host code (RCCL all-gather of the hand-over states)."""
    src = os.path.join(ROOT, "examples", "decode_tape_sharded.cpp")
    out = os.path.join(ROOT, "examples", "decode_tape_sharded")
    if not force and not _newer(out, [src, os.path.join(ROOT, "include", "sdvpcm.h"), HIP_LIB]):
        return out
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"),
                           "-I" + os.path.join(ROOT, "include"), src, "-L" + PKG, "-lsdvpcm_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-lrccl", "-lpthread",
                           "-Wl,-rpath,$ORIGIN/../sdvpcmdecoder_amd", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", out])
    return out


def build_example_sharded_emu(force=False):
    """TEST ONLY: the same host program against the emulator build of the engine (host pointers, the file back end of the collective), so
    that the loop - warm-up, all-gather, verify, repair - can be run by two processes in the GPU-less container."""
    src = os.path.join(ROOT, "examples", "decode_tape_sharded.cpp")
    emu = build_emu()
    out = os.path.join(ROOT, "tests", "emu", "decode_tape_sharded_emu")
    if not force and not _newer(out, [src, os.path.join(ROOT, "include", "sdvpcm.h"), emu]):
        return out
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-DSDV_EXAMPLE_HOST_MEMORY", "-I" + os.path.join(ROOT, "include"), src,
                           "-L" + os.path.join(ROOT, "tests", "emu"), "-lsdvpcm_emu", "-lpthread", "-Wl,-rpath,$ORIGIN", "-o", out])
    return out


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_reference():
    """Only possible where the reference tree is mounted (the build container)."""
    if not os.path.isdir("/root/reference"):
        return False
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j4"])
    return True


def build_emu(force=False):
    d = os.path.join(ROOT, "tests", "emu")
    out = os.path.join(d, "libsdvpcm_emu.so")
    srcs = [os.path.join(d, "emu_engine.cpp"), os.path.join(d, "hip_emu.h"),
            os.path.join(PKG, "csrc", "stc007_device.h"), os.path.join(PKG, "csrc", "stc007_sweep_device.h"), os.path.join(PKG, "csrc", "stc007_deint_device.h"),
            os.path.join(PKG, "csrc", "engine.inc"), os.path.join(PKG, "csrc", "stc007_stitch_device.h"),
            os.path.join(PKG, "csrc", "stitch_engine.inc"),
            os.path.join(PKG, "csrc", "pcm1_stitch_device.h"), os.path.join(PKG, "csrc", "pcm1_bin_device.h"), os.path.join(PKG, "csrc", "pcm1_engine.inc"), os.path.join(PKG, "csrc", "pcm1_frames_device.h"), os.path.join(PKG, "csrc", "pcm1_frames_engine.inc"), os.path.join(PKG, "csrc", "pcm16_bin_device.h"), os.path.join(PKG, "csrc", "pcm16_frames_device.h"), os.path.join(PKG, "csrc", "pcm16_frames_engine.inc"), os.path.join(PKG, "csrc", "pcm16_engine.inc"), os.path.join(PKG, "csrc", "pcm16_stitch_device.h"), os.path.join(PKG, "csrc", "audio_device.h"), os.path.join(PKG, "csrc", "audio_engine.inc"),
            os.path.join(PKG, "csrc", "vis_device.h"), os.path.join(PKG, "csrc", "vis_engine.inc"), os.path.join(ROOT, "include", "sdvpcm.h")]
    if not force and not _newer(out, srcs):
        return out
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-o", out,
                           os.path.join(d, "emu_engine.cpp")])
    return out


def build_all():
    # the two builds of the HIP library side by side (each is one hipcc run of a minute and a half)
    import threading
    dev_err = []
    def _dev():
        try: build_hip_dev()
        except Exception as ex: dev_err.append(ex)
    th = threading.Thread(target=_dev); th.start()
    try:
        build_hip()
    finally:
        th.join()
    if dev_err: raise dev_err[0]
    build_example()
    build_example_sharded()
    build_oracle()
    build_reference()
    build_emu()
