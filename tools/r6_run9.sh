cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
bash tools/gpu_run.sh ab ICL_UPDATE_EARLY 0 1 2>&1 | tee $O/early_ab.txt
ICL_UPDATE_WGS=160 bash tools/gpu_run.sh ab ICL_UPDATE_EARLY 0 1 2>&1 | tee -a $O/early_ab.txt
ICL_UPDATE_WGS=96 bash tools/gpu_run.sh ab ICL_UPDATE_EARLY 0 1 2>&1 | tee -a $O/early_ab.txt
TAIL=30 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path.txt
python -m pytest tests -m gpu -q -k "compat_root or update_inside_backward or bit_reproducible or three_trainer or graph_replay or one_line_swap" 2>&1 | tail -4 | tee $O/tests.txt
