#!/bin/bash
# ISA of one kernel: tools/isa_kernel.sh kernel_substring out.s [-D...]   (compiles csrc/sdvpcm_hip.hip for the device only)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
k=$1; out=$2; shift 2
cd $R/sdvpcmdecoder_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value --cuda-device-only -S "$@" -o /tmp/all_$$.s sdvpcm_hip.hip 2>/dev/null
L=$(grep -n "^_Z[0-9]*$k.*:" /tmp/all_$$.s | head -1 | cut -d: -f1)
awk -v s=$L 'NR>=s' /tmp/all_$$.s | awk '/\.end_amdhsa_kernel|^\.Lfunc_end/{exit} {print}' > $out
rm -f /tmp/all_$$.s
wc -l $out
