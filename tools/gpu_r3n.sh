R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
B="python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20"
run() { echo "$1"; env $1 $B 2>&1 | tail -1 | cut -c140-170; }
run "ICL_LOSS_LANES=0"
run "ICL_LOSS_LANES=1"
run "ICL_LOSS_LANES=0"
run "ICL_LOSS_LANES=1"
