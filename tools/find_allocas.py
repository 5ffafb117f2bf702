"""Which objects of the kernels live in scratch memory, and why: the allocas left in the optimised LLVM IR of the device code and, for each, the
accesses through a computed address (a getelementptr with a variable index, a select or phi of addresses, a call that takes the address) - the
usual reason SROA could not split the object into registers.  usage: find_allocas.py [kernel substring]   (compiles csrc/sdvpcm_hip.hip)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ll = "/tmp/sdv_all.ll"
if "--reuse" not in sys.argv:
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-emit-llvm", "-S", "-gline-tables-only", "-o", ll, "sdvpcm_hip.hip"],
                   cwd=os.path.join(ROOT, "sdvpcmdecoder_amd", "csrc"), stderr=subprocess.DEVNULL, check=True)
want = [a for a in sys.argv[1:] if not a.startswith("--")]
text = open(ll).read()
dbg = dict(re.findall(r"^(!\d+) = !DILocation\(line: (\d+), column: \d+, scope: (!\d+)", text, re.M) and [(m[0], m) for m in re.findall(r"^(!\d+) = !DILocation\(line: (\d+), column: \d+, scope: (!\d+)(?:, inlinedAt: (!\d+))?", text, re.M)])
files = {m[0]: m[1] for m in re.findall(r"^(!\d+) = (?:distinct )?!DI(?:Subprogram|LexicalBlock|LexicalBlockFile)\(.*?file: (!\d+)", text, re.M)}
fnames = dict(re.findall(r'^(!\d+) = !DIFile\(filename: "([^"]+)"', text, re.M))
scope_file = {m[0]: fnames.get(m[1], "?") for m in re.findall(r"^(!\d+) = (?:distinct )?!DI(?:Subprogram|LexicalBlock\w*)\([^\n]*?file: (!\d+)", text, re.M)}
def where(ref):
    out = []
    while ref in dbg and len(out) < 4:
        _, line, scope, inl = dbg[ref]
        out.append("%s:%s" % (os.path.basename(scope_file.get(scope, "?")), line))
        ref = inl
    return " <- ".join(out)
for m in re.finditer(r"^define [^\n]*@(\w+)\([^\n]*\n(.*?)^}", text, re.M | re.S):
    name, body = m.group(1), m.group(2)
    if want and not any(w in name for w in want):
        continue
    allocas = re.findall(r"^\s+(%[\w.]+) = alloca ([^\n]+?), align", body, re.M)
    if not allocas:
        continue
    print(re.sub(r"^_Z\d+", "", name))
    for var, ty in allocas:
        print("   ", var, ty)
        for line in body.splitlines():
            if re.search(r"(?<![\w.])" + re.escape(var) + r"(?![\w.])", line) and "getelementptr" in line:
                idx = re.split(r"(?<![\w.])" + re.escape(var) + r"(?![\w.])", line, 1)[1]
                if re.search(r"%[\w.]+", idx.split("!dbg")[0]):
                    d = re.search(r"!dbg (!\d+)", line)
                    print("        computed address:", line.strip()[:110], "|", where(d.group(1)) if d else "")
            elif re.search(r"(?<![\w.])" + re.escape(var) + r"(?![\w.])", line) and re.search(r"\b(select|phi|call)\b", line) and "lifetime" not in line:
                d = re.search(r"!dbg (!\d+)", line)
                print("        address escapes:", line.strip()[:110], "|", where(d.group(1)) if d else "")
