cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
bash tools/gpu_run.sh ab ICL_UPDATE_SMALL_AT_GATE 0 1 2>&1 | tee $O/small_gate_ab.txt
bash tools/gpu_run.sh ab ICL_PACK_STREAM 0 1 2>&1 | tee $O/pack_stream_ab.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -3 | tee $O/suite.txt
