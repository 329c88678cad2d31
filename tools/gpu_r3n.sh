R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for V in 0 1 0 1; do echo SGD_DEFER=$V; ICL_SGD_DEFER=$V python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20 2>&1 | tail -1 | cut -c140-170; done
