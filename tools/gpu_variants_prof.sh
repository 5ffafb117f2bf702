# kernel stats of the frame kernels under alternative builds of the library (build/variants/*.so):  gpurun -- 'bash tools/gpu_variants_prof.sh <script> <kernel regex> [args]'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
SCRIPT=$1; KRE=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for lib in product $R/build/variants/*.so; do
  rm -rf $R/gpurun_out/prof_var
  if [ "$lib" = product ]; then unset SDVPCM_LIB; else export SDVPCM_LIB=$lib; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_var -- python3 $R/tools/$SCRIPT "$@" > $R/gpurun_out/prof_var.log 2>&1
  f=$(ls -t $R/gpurun_out/prof_var/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $(basename $lib): $(grep -c 'frames/s' $R/gpurun_out/prof_var.log) result lines"
  if [ -n "$f" ]; then grep -E "$KRE" "$f" | cut -d, -f1-7; fi
done
