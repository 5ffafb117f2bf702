#!/usr/bin/env python3
"""The generated parts of DESIGN.md, each between a pair of markers:
  numbers    the table of measured numbers, from profiles/<round>_bench_full.json and the PMC file of the lean frame kernel
  resources  registers / scratch / LDS / waves per SIMD of the kernels the text talks about, from hipcc's own remarks (tools/kernel_resources.py; the table of
             every kernel goes to profiles/<round>_kernel_resources.md)
  issue      the roofline of the kernels HBM does not bound (tools/issue_roofline.py over the PMC files under profiles/)
usage: design_numbers.py [r06] [--no-compile]     (--no-compile: keep the resources block as it is)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rnd = args[0] if args else "r06"
d = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_bench_full.json")))


def g(*p):
    x = d
    for k in p:
        x = x[k]
    return x


pal, dm = d["pal_stage"], d["damaged_tape"]
c3 = pal["lost_lines_and_flipped_cells"]
pmc = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_sdv_k_stc007_frames_lean.json")))
fetch = next(v["FETCH_SIZE"] for v in pmc.values() if isinstance(v, dict) and "FETCH_SIZE" in v)
write = next(v["WRITE_SIZE"] for v in pmc.values() if isinstance(v, dict) and "WRITE_SIZE" in v)
# the guide's corrections (MI355X_MICROARCH.md, HBM / rocprofv3): both counters in KiB, FETCH_SIZE x 2 on gfx950
traffic = (2 * fetch + write) * 1024
alg = g("roofline", "algorithmic_bytes_per_launch")
c4 = d.get("configs4_strong")
rows = [
    ("**headline**: `sdv_binarize_frames`, configs[1]",
     f"{d['value'] / 1e6:.2f} M frames/s, {d['ms_per_step']:.3f} ms per step; lean kernel {g('roofline', 'avg_launch_ms'):.3f} ms per launch = {g('roofline', 'achieved') / 1000:.2f} TB/s algorithmic = **{g('roofline', 'frac'):.3f} of the 8 TB/s peak**",
     f"PMC traffic {traffic / 1e9:.2f} GB per launch = {traffic / alg:.2f} × algorithmic (read {2 * fetch * 1024 / 1e9:.2f} GB, written {write * 1024 / 1e6:.0f} MB for {10000 * 489 * 48 / 1e6:.0f} MB of records; `profiles/{rnd}_pmc_sdv_k_stc007_frames_lean.json`); same-box comparisons with the libraries of the rounds before: `profiles/r05_tuning_notes.md` §1, `profiles/r06_tuning_notes.md` §0"),
    ("CPU baseline (real reference, `oracle/_ref`)", f"{g('cpu_baseline', 'value'):.0f} frames/s on one core; {g('cpu_baseline_all_cores', 'value'):.0f} frames/s with a worker on each of the {g('cpu_baseline_all_cores', 'cores')} cores the container is granted", "bit-exact on the overlap"),
    ("frames → pairs, fused (`sdv_decode_frames`)", f"{g('end_to_end', 'fused_entry_ms_per_step'):.3f} ms = {10000 / g('end_to_end', 'fused_entry_ms_per_step') / 1e3:.2f} M frames/s, {g('end_to_end', 'roofline', 'frac'):.2f} of peak on 377 232 B per frame; the two entry points one after the other: {g('end_to_end', 'two_calls_ms_per_step'):.3f} ms", "what the fused entry saves: §6, stitch stage"),
]
if c4:
    rows.append(("**configs[4]**: one 100 000-frame stream, NEW_FILE … END_FILE, frames → pairs (`ShardedDecoder`, strong scaling)", f"{c4['ms']:.2f} ms = {c4['frames_per_s'] / 1e6:.2f} M frames/s at {c4['ranks']} rank, {c4['roofline']['frac']:.2f} of peak on 377 232 B per frame", "the same leg runs at every N (`bench.py --gpus N`): the tape is cut into N frame ranges, one all-gather of the hand-over states"))
rows += [
    ("stitch stage alone", f"{g('stitch_stage', 'stitch_ms_per_step'):.3f} ms ({g('stitch_stage', 'stitch_device_ms_per_step'):.3f} ms on the device)", f"CPU: {g('stitch_stage', 'cpu_baseline', 'value'):.0f} frames/s"),
    ("PAL (configs[2]) clean, binarize + stitch", f"{pal['clean']['ms_per_step']:.2f} ms = {pal['clean']['frames_per_s'] / 1e6:.2f} M frames/s for the two entry points one after the other" + (f"; fused entry {pal['clean']['fused_entry_ms_per_step']:.2f} ms = {pal['clean']['frames_per_step'] / pal['clean']['fused_entry_ms_per_step'] / 1e3:.2f} M frames/s" if pal['clean'].get('fused_entry_ms_per_step') else ""), f"{pal['clean']['frames_per_step']} frames of 720 x 576 per step"),
    ("PAL C3 tape (2000 frames; every 97th line lost, a cell inverted on one line in 53)",
     f"binarize {c3['binarize_ms_per_step']:.1f} ms ({2000 / c3['binarize_ms_per_step']:.0f} k frames/s), {c3['binarize_rounds_per_step']:.0f} rounds, {c3['reference_level_sweeps_per_step']:.0f} sweeps; + stitch {c3['stitch_ms_per_step']:.1f} ms → **{c3['frames_per_s'] / 1e3:.1f} k frames/s**",
     f"CPU (real reference): {c3['cpu_baseline']['value']:.0f} frames/s, bit-exact on the overlap; round 5: 90 k frames/s, 13 + 6 rounds (this round: small rounds that settle their sweeps themselves, sweep comparisons on bit planes - `profiles/r06_tuning_notes.md` §9, §10; what was tried besides: §2, §5, §7)"),
    ("16 lost lines / 16 window jumps per 10 000 frames", f"{dm['lost_lines']['ms_per_step']:.2f} ms ({dm['lost_lines']['rounds_per_step']:.0f} rounds) / {dm['window_jumps']['ms_per_step']:.1f} ms ({dm['window_jumps']['rounds_per_step']:.0f} rounds, {dm['window_jumps']['frames_by_full_kernel_per_step']:.0f} frames through the general kernel)", "kernel time of the rounds: " + f"{dm['lost_lines']['kernel_ms_per_step']:.2f} / {dm['window_jumps']['kernel_ms_per_step']:.2f} ms"),
    ("the whole tape two pixels beside its coordinates / every 97th row of every frame lost", f"{dm['beside_coordinates']['ms_per_step']:.2f} ms ({dm['beside_coordinates']['rounds_per_step']:.0f} rounds) / {dm['lost_lines_in_every_frame']['ms_per_step']:.2f} ms ({dm['lost_lines_in_every_frame']['rounds_per_step']:.0f} rounds, {dm['lost_lines_in_every_frame']['frames_by_full_kernel_per_step']:.0f} frame decodes by the general kernel)", "round 5 (same boxes, other tools): 2.7 ms per 10 000 frames beside their coordinates; `profiles/r06_tuning_notes.md` §2, §3") if 'beside_coordinates' in dm else ("damaged tapes, more", "-", "-"),
    ("PCM-1: line kernel / frame driver / stitch", f"{g('pcm1_front_stage', 'ms_per_step'):.2f} ms per 980 000 lines / {g('pcm1_frames_stage', 'ms_per_step'):.2f} ms / {g('pcm1_stage', 'ms_per_step'):.2f} ms", f"CPU: {g('pcm1_frames_stage', 'cpu_baseline', 'value'):.0f} frames/s (frame driver); round 4: 6.28 ms"),
    ("PCM-16x0: frame driver / stitch SI / EI", f"{g('pcm16x0_frames_stage', 'ms_per_step'):.2f} ms / {g('pcm16x0_stage', 'si', 'ms_per_step'):.2f} ms / {g('pcm16x0_stage', 'ei', 'ms_per_step'):.2f} ms", f"CPU: {g('pcm16x0_frames_stage', 'cpu_baseline', 'value'):.0f} frames/s (frame driver); round 4: 10.76 ms"),
    ("AudioProcessor: clean / dropout every 25 frames / invalid word in every window", f"{g('audio_stage', 'clean', 'ms_per_step'):.2f} / {g('audio_stage', 'dropout_every_25_frames', 'ms_per_step'):.2f} / {g('audio_stage', 'invalid_word_in_every_window', 'ms_per_step'):.2f} ms", "CPU: 1.3–2.8 k frames/s"),
    ("host-fed (the boundary handed host frames)", f"{g('host_fed', 'h2d_gb_per_s'):.0f} GB/s pinned host → HBM: at most {g('host_fed', 'frames_per_s_bound_by_pcie') / 1e3:.0f} k frames/s", "never `value`"),
]
table = f"| what (10 000 NTSC frames per step unless said) | measured (`profiles/{rnd}_bench_full.json`, one box) | note |\n|---|---|---|\n" + "".join(f"| {a} | {b} | {c} |\n" for a, b, c in rows)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()


def put(text, name, block):
    a, b = f"<!-- {name}: begin (tools/design_numbers.py) -->\n", f"<!-- {name}: end -->\n"
    if a not in text:
        raise SystemExit(f"DESIGN.md has no `{name}` markers")
    return text[:text.index(a) + len(a)] + block + text[text.index(b):]


s = put(s, "numbers", table)
if "--no-compile" not in sys.argv:
    import kernel_resources as kr
    res = kr.parse(kr.remarks())
    open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_resources.md"), "w").write(
        "Registers, scratch, LDS and waves per SIMD of every kernel of `libsdvpcm_hip.so`, from hipcc's remarks (`tools/kernel_resources.py --md`).\n\n" + kr.table(res, md=True))
    main = ["stc007_frames", "stc007_sweep", "stitch_analyze", "stitch_step", "pcm1_frames", "pcm1_prescanN", "pcm16_frames_lean", "pcm16_frames_binN", "pcm16_prescanN", "pcm1_linesN",
            "pcm16_linesN", "pcm16_analyse", "pcm16_choose", "ap_plan", "ap_prepare", "stc007_deint"]
    s = put(s, "resources", kr.table(res, md=True, only=main))
issue = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "issue_roofline.py"), "--md"], stdout=subprocess.PIPE, text=True, check=True).stdout
s = put(s, "issue", issue)
open(path, "w").write(s)
print(table)
