cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
bash tools/gpu_run.sh ab ICL_USCL_LANE_ROT 0 1 2>&1 | tee $O/rot_ab.txt
bash tools/gpu_run.sh ab ICL_USCL_LANE_ROT 0 2 2>&1 | tee -a $O/rot_ab.txt
bash tools/gpu_run.sh ab ICL_QCHAIN 0 1 --num-classes 16 2>&1 | tee $O/qchain_ab_nc16.txt
python tests/diag/dense_wgrad_errors.py > $O/dense_errors.txt 2>&1; cat $O/dense_errors.txt
python -m pytest tests -m gpu -q -k "ten_trainer_steps or 200_steps or compat_root" 2>&1 | tail -5 | tee $O/new_tests.txt
python bench.py --steps 10 --no-cpu-baseline --no-exact-compare > $O/bench_other.json 2> $O/bench_other.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6g/bench_other.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], json.dumps(d['config'].get('other_workloads'), indent=1)[:1500])
PY
