#!/bin/bash
# instructions, scratch instructions and calls of every kernel AND every out-of-line device function of the HIP library (the compiler's resource remarks
# fold the callees into the kernels):  tools/func_stats.sh [name filter regex] [-D...]
# What it is for: a device function that kernels share is compiled once, for the registers of one of its callers; a function with two call sites is not inlined
# and what it takes by reference goes to scratch memory.  Both show up here as scratch_ counts of the callee (profiles/r06_tuning_notes.md, section 5).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
filter=${1:-.}; shift || true
cd $R/sdvpcmdecoder_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value --cuda-device-only -S "$@" -o /tmp/func_stats_$$.s sdvpcm_hip.hip 2>/dev/null
FILTER="$filter" python3 - /tmp/func_stats_$$.s <<'PY'
import os, re, sys
out = {}; cur = None
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m: cur = m.group(1); out[cur] = [0, 0, 0]; continue
    if cur and re.match(r'^\s+[vsdgb]\w*_', line):
        out[cur][0] += 1
        if 'scratch_' in line: out[cur][1] += 1
        if 's_swappc' in line: out[cur][2] += 1
    if line.startswith('.Lfunc_end'): cur = None
pat = re.compile(os.environ.get('FILTER', '.'))
print(f"{'function':90s} {'instr':>7s} {'scratch':>8s} {'calls':>6s}")
for k, v in sorted(out.items(), key=lambda kv: -kv[1][1]):
    if pat.search(k) and v[0] > 0: print(f"{k[:90]:90s} {v[0]:7d} {v[1]:8d} {v[2]:6d}")
PY
rm -f /tmp/func_stats_$$.s
