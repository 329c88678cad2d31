R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
B="python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20"
run() { echo "$1"; env $1 $B 2>&1 | tail -1 | cut -c140-170; }
run "X=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "HSA_ENABLE_INTERRUPT=0"
run "X=0"
run "HIP_FORCE_DEV_KERNARG=1"
run "HSA_ENABLE_INTERRUPT=0"
