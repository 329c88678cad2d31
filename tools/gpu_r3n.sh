R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for V in 0 64 32 16 0 32; do echo swin SPLIT_MAX=$V; ICL_GEMM_SPLIT_MAX=$V python3 bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare --no-kernel-timer --steps 10 2>&1 | tail -1 | cut -c140-170; done
for V in 0 32; do echo unet SPLIT_MAX=$V; ICL_GEMM_SPLIT_MAX=$V python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20 2>&1 | tail -1 | cut -c140-170; done
