# round 6: the GPU suite, the soak and the bench line on the tree as it stands
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/soak.py 8 2000 > gpurun_out/soak_r06.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/soak_r06.log
python bench.py 2>/dev/null | tail -1 | cut -c1-2600
