# a longer soak with other seeds than the validation's:  gpurun -- 'bash tools/gpu_soak.sh [cases] [first seed]'
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/soak.py ${1:-40} ${2:-5000} 2>&1 | tail -4
