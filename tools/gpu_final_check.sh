# the driver's round-end checks on the tree as it stands:  gpurun -- 'bash tools/gpu_final_check.sh'
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 | cut -c1-700
