cd $GRAFT_REPO_ROOT
O=gpurun_out/r6x; mkdir -p $O
bash tools/gpu_run.sh ab ICL_CONV_SPLIT_KSPLIT 0 -1 --model swinunetr_icl 2>&1 | tee $O/swin_ksplit_ab.txt
bash tools/gpu_run.sh ab ICL_CONV_SPLIT_KSPLIT 0 -1 --num-classes 16 2>&1 | tee $O/nc16_ksplit_ab.txt
