cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_ten_trainer_steps_match_reference_golden 2>&1 | tail -8 | tee $O/suite.txt
bash tools/gpu_run.sh ab ICL_TOKEN_LANE 0 1 2>&1 | tee $O/token_lane_ab.txt
CP_ALIGNER_DETAIL=1 TAIL=60 bash tools/gpu_run.sh critical-path 2>&1 | tee $O/critical_path_detail.txt
python tests/diag/ten_steps.py > $O/ten_steps_noise.txt 2>&1; tail -70 $O/ten_steps_noise.txt
python tests/diag/drift_200.py > $O/drift.txt 2>&1; tail -30 $O/drift.txt
