# round 5, experiment 1: the lean kernel's record stores.  gpurun -- 'bash tools/gpu_k1_exp1.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.get_device_name(0))"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_decode_frames.py tests/test_dropped_frames.py tests/test_sharded.py tests/test_stitch_kernel.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/k1_ab.py 6 build/ab/r05_start.so build/variants/*.so 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
for v in $R/build/ab/r05_start.so $R/build/variants/coal.so $R/build/variants/coal_plain.so $R/build/variants/nocoal.so; do
  b=$(basename $v .so)
  rm -rf $R/gpurun_out/x1_${b}_w $R/gpurun_out/x1_${b}_f
  timeout 300 rocprofv3 --kernel-include-regex 'sdv_k_stc007_frames_lean' --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/x1_${b}_w -- python3 $R/tools/k1_once.py $v > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-include-regex 'sdv_k_stc007_frames_lean' --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/x1_${b}_f -- python3 $R/tools/k1_once.py $v > /dev/null 2>&1
  python3 $R/tools/pmc_sizes.py sdv_k_stc007_frames_lean $R/gpurun_out/x1_${b}_w $R/gpurun_out/x1_${b}_f
done
