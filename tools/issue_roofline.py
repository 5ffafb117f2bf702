#!/usr/bin/env python3
"""A roofline for the kernels that HBM does not bound: achieved VALU issue against what the SIMDs can issue, from the PMC passes kept under profiles/
(`<round>_pmc_<kernel>.json`, the newest round that has the kernel), plus where the waves' cycles went - the limiter the counters name.

  kernel cycles   = SQ_BUSY_CYCLES / 32            (the counter sums the 32 shader engines' busy cycles; checked against the rocprofv3 launch durations)
  issue peak      = 1 VALU wave-instruction per 2 cycles per SIMD (a wave64 instruction takes two passes through a SIMD-32: MI355X_MICROARCH.md,
                    "Wave scheduling", `v_fma_f32 (wave64) 2 cyc`), 4 SIMDs x 256 CUs
  issue fraction  = SQ_INSTS_VALU x 2 / (kernel cycles x 1024)        [x SIMDs in use / 1024 for a kernel of a few waves]
  wave cycles     = SQ_WAVE_CYCLES (quad-cycles), split by SQ_ACTIVE_INST_ANY (issuing), SQ_WAIT_INST_ANY (ready but not issued: the issue port, a busy
                    pipe, a dependency) and the rest (parked at s_waitcnt / a barrier: memory and LDS latency)
usage: issue_roofline.py [--md]"""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["sdv_k_stc007_sweep_levels", "sdv_k_stc007_sweep_pick", "sdv_k_stc007_frames", "sdv_k_stc007_frames_plain", "sdv_k_stc007_frames_fat", "sdv_k_stitch_analyze", "sdv_k_stitch_step", "sdv_k_pcm1_prescan", "sdv_k_pcm16_prescan",
           "sdv_k_pcm1_frames_lean", "sdv_k_pcm16_frames_lean", "sdv_k_pcm16_analyse_si", "sdv_k_pcm16_analyse_ei", "sdv_k_ap_plan"]


def newest(kernel):
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % kernel)))
    fs = [f for f in fs if re.search(r"r\d+_pmc_%s\.json$" % re.escape(kernel), f)]
    return fs[-1] if fs else None


def counters(path):
    d = json.load(open(path)); c = {}
    for v in d.values():
        if isinstance(v, dict):
            c.update({k: x for k, x in v.items() if k.isupper() or k in ("grid_size", "vgpr", "lds", "scratch")})
    return c


def rows():
    out = []
    for k in KERNELS:
        f = newest(k)
        if not f:
            continue
        c = counters(f)
        if "SQ_INSTS_VALU" not in c or "SQ_BUSY_CYCLES" not in c:
            continue
        waves = c.get("SQ_WAVES", 0) or 1
        cyc = c["SQ_BUSY_CYCLES"] / 32.0
        simds = min(1024.0, waves)                      # a kernel of a few waves cannot use more SIMDs than it has waves
        if waves < 64:
            cyc = c["SQ_WAVE_CYCLES"] * 4.0 / waves      # a handful of waves: their own residence time is the kernel's length
        frac = c["SQ_INSTS_VALU"] * 2.0 / (cyc * simds)
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        act = c.get("SQ_ACTIVE_INST_ANY", 0) / wc
        stall = c.get("SQ_WAIT_INST_ANY", 0) / wc
        parked = max(0.0, 1.0 - act - stall)
        per_wave = c["SQ_INSTS_VALU"] / waves
        salu = c.get("SQ_INSTS_SALU", 0) / waves
        lds = c.get("SQ_INSTS_LDS", 0) / waves
        resident = wc * 4.0 / cyc / simds               # mean waves resident per SIMD in use
        if frac >= 0.45:
            lim = "VALU issue"
        elif parked >= 0.5:
            lim = "latency (waves parked at s_waitcnt / barriers %.0f %% of their cycles)" % (100 * parked)
        elif stall >= 0.4:
            lim = "issue stalls (ready waves not issued %.0f %% of their cycles: dependencies, SALU : VALU = %.1f)" % (100 * stall, salu / max(per_wave, 1))
        else:
            lim = "occupancy x latency (%.1f waves per SIMD in flight)" % resident
        out.append((k, os.path.basename(f), waves, per_wave, salu, lds, cyc, frac, act, stall, parked, resident, lim))
    return out


if __name__ == "__main__":
    r = rows()
    if "--md" in sys.argv:
        print("| kernel | counters | waves | VALU / SALU / LDS instructions per wave | kernel cycles | VALU issue: achieved / peak | waves' cycles: issuing / stalled / parked | waves per SIMD in flight | limiter |")
        print("|---|---|---|---|---|---|---|---|---|")
        for k, f, waves, pw, salu, lds, cyc, frac, act, stall, parked, res, lim in r:
            print("| `%s` | `%s` | %d | %.0f / %.0f / %.0f | %.2f M | **%.2f** | %.0f %% / %.0f %% / %.0f %% | %.1f | %s |" % (k, f, waves, pw, salu, lds, cyc / 1e6, frac, 100 * act, 100 * stall, 100 * parked, res, lim))
    else:
        for k, f, waves, pw, salu, lds, cyc, frac, act, stall, parked, res, lim in r:
            print("%-28s %-36s waves %7d  VALU/wave %8.0f  cycles %8.2f M  issue frac %.3f  act %.2f stall %.2f parked %.2f  res %.1f  %s" % (k, f, waves, pw, cyc / 1e6, frac, act, stall, parked, res, lim))
