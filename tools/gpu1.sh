set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ls /opt/conda/lib/libQt5Core.so.5 2>&1 | head -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
