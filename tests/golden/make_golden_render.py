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
    for name in ra.BLOCK_GOLDEN:          # the data blocks window: the real stitcher's blocks on the real RenderPCM
        if only and name not in only:
            continue
        import libs
        import oracle_run
        import stitch_api as sa
        import stitch_cases as sc
        kind, case = ra.BLOCK_CASES[name]
        recs, st = sc.make_input(case, lambda luma: oracle_run.oracle_binarize(luma, mode=2))
        pairs, frames, blocks = sa.run_cpu_blocks(libs.load_ref(), "ref_", recs, st)
        per = np.ascontiguousarray(frames["blocks_total"][frames["service_type"] == 0].astype(np.uint32))
        ref = ra.run_ref_blocks(kind, np.ascontiguousarray(blocks), per)
        mask = ra.written_blocks(kind, per)
        path = os.path.join(HERE, "render_" + name + ".npz")
        np.savez_compressed(path, blocks_sha256=hashlib.sha256(blocks.tobytes()).hexdigest(), canvases_sha256=ra.digest(ref, mask),
                            last_canvas=np.where(mask[-1], ref[-1], 0).astype(np.uint32))
        print(f"{name}: {len(blocks)} blocks -> {len(ref)} canvases of {ref.shape[2]} x {ref.shape[1]}, {os.path.getsize(path)} bytes")
    for name in ra.ASM_GOLDEN:            # the assembled-lines window: the real stitcher's lines on the real RenderPCM
        if only and name not in only:
            continue
        import libs
        import oracle_run
        import stitch_api as sa
        import stitch_cases as sc
        kind, case = ra.ASM_CASES[name]
        recs, st = sc.make_input(case, lambda luma: oracle_run.oracle_binarize(luma, mode=2))
        sa.run_cpu_blocks(libs.load_ref(), "ref_", recs, st)
        lines, per = sa.last_asm_lines(libs.load_ref(), "ref_")
        ref = ra.run_ref_asm(kind, np.ascontiguousarray(lines), np.ascontiguousarray(per))
        mask = ra.written_blocks(kind, per)
        path = os.path.join(HERE, "render_" + name + ".npz")
        np.savez_compressed(path, lines_sha256=hashlib.sha256(lines.tobytes()).hexdigest(), canvases_sha256=ra.digest(ref, mask),
                            last_canvas=np.where(mask[-1], ref[-1], 0).astype(np.uint32))
        print(f"{name}: {len(lines)} lines -> {len(ref)} canvases of {ref.shape[2]} x {ref.shape[1]}, {os.path.getsize(path)} bytes")
