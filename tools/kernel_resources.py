"""Register / scratch / LDS / occupancy figures of every kernel, from hipcc's own remarks (-Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py [remarks.txt]   (without a file: compiles csrc/sdvpcm_hip.hip for gfx950 and reads the remarks of that build)
Prints a table; with --md a markdown table (DESIGN.md section 6 is generated from it by tools/design_numbers.py)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def remarks(path=None, defines=()):
    if path:
        return open(path).read()
    csrc = os.path.join(ROOT, "sdvpcmdecoder_amd", "csrc")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage",
           "-o", "/dev/null", "sdvpcm_hip.hip"] + ["-D" + d for d in defines]
    return subprocess.run(cmd, cwd=csrc, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, check=True).stderr


def parse(text):
    out, cur = {}, None
    keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "SGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
            "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds", "TotalSGPRs": "sgpr", "Dynamic Stack": "dyn_stack"}
    for line in text.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {}); continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*?): (\S+) \[-Rpass", line)
        if m and cur is not None and m.group(1).strip() in keys:
            v = m.group(2)
            cur[keys[m.group(1).strip()]] = int(v) if v.isdigit() else v
    return out


def plain_name(mangled):
    m = re.match(r"_Z(\d+)", mangled)
    return mangled[m.end():m.end() + int(m.group(1))] if m else mangled


def table(res, md=False, only=None):
    rows = []
    for k in sorted(res, key=plain_name):
        if only and not any(s in k for s in only):
            continue
        r = res[k]
        if md and "sdv_k_" not in k:
            continue
        rows.append((plain_name(k) if md else k, r.get("vgpr", 0), r.get("agpr", 0), r.get("sgpr", 0), r.get("vgpr_spill", 0), r.get("sgpr_spill", 0), r.get("scratch", 0), r.get("lds", 0), r.get("occupancy", 0)))
    hdr = ("kernel", "VGPR", "AGPR", "SGPR", "VGPR spills", "SGPR spills", "scratch B/lane", "LDS B", "waves/SIMD")
    if md:
        s = "| " + " | ".join(hdr) + " |\n|" + "---|" * len(hdr) + "\n"
        for r in rows:
            s += "| `" + r[0] + "` | " + " | ".join(str(x) for x in r[1:]) + " |\n"
        return s
    s = "%-40s %5s %5s %5s %6s %6s %8s %7s %5s\n" % hdr
    for r in rows:
        s += "%-40s %5d %5d %5d %6d %6d %8d %7d %5s\n" % r
    return s


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    print(table(parse(remarks(args[0] if args else None)), md="--md" in sys.argv), end="")
