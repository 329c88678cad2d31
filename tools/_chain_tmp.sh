cd /root/repo
bash tools/gpu_run.sh bench --steps 30 -- bench --steps 30 -- critical-path -- suite
python tests/diag/momentum_sample.py 2>&1 | tail -1
