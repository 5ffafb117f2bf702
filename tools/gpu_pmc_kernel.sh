# Four rocprofv3 --pmc passes (instruction mix, waits, FETCH_SIZE, WRITE_SIZE: each its own run, no tracing) of ONE kernel:
#   bash tools/gpu_pmc_kernel.sh <kernel regex> <dir prefix under gpurun_out/> <python script> [args...]
# The program goes directly after `--` (python3 script args): no env / bash -c hop (the profiler's preloaded library has initialised the GPU).
# tools/pmc_to_json.py <kernel> <out.json> <prefix> "<workload>" turns gpurun_out/<prefix>1..4 into the JSON kept under profiles/.
R=$GRAFT_REPO_ROOT
K=$1; P=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${P}1 $R/gpurun_out/${P}2 $R/gpurun_out/${P}3 $R/gpurun_out/${P}4
timeout 600 rocprofv3 --kernel-include-regex "$K" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${P}1 -- python3 "$@" > /dev/null 2> $R/gpurun_out/${P}1.err; echo "${P}1 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex "$K" --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/${P}2 -- python3 "$@" > /dev/null 2> $R/gpurun_out/${P}2.err; echo "${P}2 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex "$K" --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}3 -- python3 "$@" > /dev/null 2> $R/gpurun_out/${P}3.err; echo "${P}3 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex "$K" --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}4 -- python3 "$@" > /dev/null 2> $R/gpurun_out/${P}4.err; echo "${P}4 rc=$?"
