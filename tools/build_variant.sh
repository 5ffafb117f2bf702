#!/bin/bash
# build a variant of the HIP library into build/variants/<name>.so:  tools/build_variant.sh name [-DX=Y ...]; prints the lean kernel's resources
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=$1; shift
mkdir -p $R/build/variants
cd $R/sdvpcmdecoder_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Rpass-analysis=kernel-resource-usage "$@" -o $R/build/variants/$name.so sdvpcm_hip.hip 2> /tmp/res_$name.txt
python $R/tools/kernel_resources.py /tmp/res_$name.txt | grep -E "kernel|${GREP:-frames_lean|stc007_framesN}"
