R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
B="python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20"
run() { echo "$1"; env $1 $B 2>&1 | tail -1 | cut -c140-170; }
run "X=0"
run "ICL_STREAM_DEPTH=6"
run "ICL_STREAM_MAXLEN=1024"
run "ICL_STREAM_MAXLEN=2048"
run "X=0"
run "ICL_STREAM_DEPTH=6"
