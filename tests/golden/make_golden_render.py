"""Generates tests/golden/render_*.npz from the REAL RenderPCM (oracle/_ref/libsdvref.so, oracle/ref_render_driver.cpp):
the canvases of the binarized-lines visualiser for the scenarios of tests/render_api.py.  Run in the build container only (it needs
/root/reference compiled by oracle/Makefile.ref):

    python tests/golden/make_golden_render.py [names...]

A fixture holds the SHA-256 of the record stream (the tests regenerate it from the seeded generators), the SHA-256 of all canvases with the
pixels no frame has drawn yet zeroed (the real canvas holds uninitialised memory there), and the last frame's canvas masked the same way."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import render_api as ra  # noqa: E402

if __name__ == "__main__":
    only = sys.argv[1:]
    for name in ra.GOLDEN:
        if only and name not in only:
            continue
        kind, recs = ra.make_input(name)
        ref = ra.run_ref(kind, recs)
        mask = ra.written(kind, recs)
        path = os.path.join(HERE, "render_" + name + ".npz")
        np.savez_compressed(path, input_sha256=hashlib.sha256(recs.tobytes()).hexdigest(), canvases_sha256=ra.digest(ref, mask),
                            last_canvas=np.where(mask[-1], ref[-1], 0).astype(np.uint32))
        print(f"{name}: {len(recs)} records -> {len(ref)} canvases of {ref.shape[2]} x {ref.shape[1]}, {os.path.getsize(path)} bytes")
