# HBM traffic counters (separate passes, MI355X_MICROARCH.md 'HBM'): bash tools/pmc_hbm.sh <tag> conv <conv_one args> | gemm <gemm_one args>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; KIND=$2; shift; shift
PROG=$R/tools/conv_one.py; [ "$KIND" = "gemm" ] && PROG=$R/tools/gemm_one.py
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/hbm_${TAG}_f -o p --output-format csv -- python3 $PROG "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/hbm_${TAG}_w -o p --output-format csv -- python3 $PROG "$@" > /dev/null 2>&1
rm -f $R/gpurun_out/hbm_${TAG}_*/p_kernel_trace.csv
