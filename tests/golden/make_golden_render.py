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
    for name in ra.P1VIS_GOLDEN:          # the two windows of the PCM-1 stitcher: the real stitcher's blocks and sub-lines on the real RenderPCM
        if only and name not in only:
            continue
        import libs
        import pcm1_api as p1
        recs, st = p1.make_input(name)
        pairs, frames, blocks, lines_ref = p1.run_cpu_vis(libs.load_ref(), "ref_", recs, st)
        per = np.full(int((frames["service_type"] == 0).sum()), 16, dtype=np.uint32)
        # the places of the engine's line buffer (1470 per frame, the ones the stitcher does not hand over marked): the oracle's, whose handed-over ones equal the real stitcher's
        _, _, lines = ra.make_p1vis_input(name)
        assert lines[lines["flags"] != 0x80].tobytes() == lines_ref.tobytes()
        bref = ra.run_ref_blocks(ra.PCM1_BLOCKS, np.ascontiguousarray(blocks), per)
        lref = ra.run_ref_lines_into(ra.PCM1_ASM, lines, len(per))
        lmask = ra.written_p1_asm(lines)
        path = os.path.join(HERE, "render_p1vis_" + name + ".npz")
        np.savez_compressed(path, block_canvases_sha256=ra.digest(bref, ra.written_p1_blocks(per)), line_canvases_sha256=ra.digest(lref, lmask),
                            last_line_canvas=np.where(lmask[-1], lref[-1], 0).astype(np.uint32))
        print(f"{name}: {len(blocks)} blocks, {len(lines_ref)} sub-lines -> {len(per)} canvases each, {os.path.getsize(path)} bytes")
