cd $GRAFT_REPO_ROOT; O=gpurun_out
for v in 16 8 4 2; do ICL_STREAM_SMALL_WGS=$v python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('small_wgs=$v', d['ms_per_step'])"; done
for v in 16 8 4; do ICL_STREAM_SMALL_WGS=$v python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --model swinunetr_icl 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swin small_wgs=$v', d['ms_per_step'])"; done
