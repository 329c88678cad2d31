cd $GRAFT_REPO_ROOT
O=gpurun_out/r6d; mkdir -p $O
for w in 64 96 128 160; do
  ICL_UPDATE_WGS=$w bash tools/gpu_run.sh ab ICL_UPDATE_PLACEMENT tail deep 2>&1 | tee -a $O/deep_ab.txt
done
ICL_UPDATE_PLACEMENT=deep ICL_UPDATE_WGS=96 TAIL=30 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path_deep96.txt
ICL_UPDATE_PLACEMENT=deep python -m pytest tests -m gpu -q -x -k "three_trainer_steps or bit_reproducible or graph_replay_equals or update_inside_backward" 2>&1 | tail -5 | tee $O/suite_deep.txt
