#!/bin/bash
# Rehearsal of the EIGHT-rank data-parallel step on ONE GPU: eight processes share device 0 over gloo
# (ICL_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device).  It exercises the code path the 8-GPU bench will take — self-launch,
# rendezvous, calibrate(), the three / four hipGraphs per rank, eager collectives between the replays, row-sharded updates with their
# overlapped all-gathers (nc = 16, forced by ICL_DDP_SHARD_ROWS since gloo's measured rates would never choose them), MAX-over-ranks
# timing, the plan on the JSON line.  Its TIMINGS MEAN NOTHING (eight ranks time-share one GPU, collectives stage through the host).
# Then the one-rank RCCL-group step of each workload with its projection to 8 ranks (bench.py --project-world 8).
#   tools/gpu_run.sh is not needed: run from the repository root on the GPU box; outputs under gpurun_out/
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
export ICL_BENCH_SHARE_GPU=1 ICL_BENCH_BACKEND=gloo
timeout 1500 python bench.py --gpus 8 --steps 3 --warmup 2 --no-cpu-baseline --no-exact-compare --no-kernel-timer > $O/r5_ddp_rehearsal_eight_ranks_nc2.json 2> $O/r5_ddp_rehearsal_eight_ranks_nc2.err
echo "nc=2 rc=$?"; tail -c 600 $O/r5_ddp_rehearsal_eight_ranks_nc2.json
ICL_DDP_SHARD_ROWS=768 timeout 2400 python bench.py --gpus 8 --steps 3 --warmup 2 --num-classes 16 --no-cpu-baseline --no-exact-compare --no-kernel-timer > $O/r5_ddp_rehearsal_eight_ranks_nc16.json 2> $O/r5_ddp_rehearsal_eight_ranks_nc16.err
echo "nc=16 rc=$?"; tail -c 600 $O/r5_ddp_rehearsal_eight_ranks_nc16.json
unset ICL_BENCH_SHARE_GPU ICL_BENCH_BACKEND
for nc in 2 16; do
  timeout 900 python bench.py --gpus 1 --force-ddp --project-world 8 --num-classes $nc --steps 20 --no-cpu-baseline --no-exact-compare --no-kernel-timer > $O/r5_ddp_one_rank_projection_nc$nc.json 2> $O/r5_ddp_one_rank_projection_nc$nc.err
  python - $O/r5_ddp_one_rank_projection_nc$nc.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
p = d['config']['ddp_plan'].get('projection')
print('one-rank group', d['ms_per_step'], 'ms;', {k: p[k] for k in p if k not in ('matrices', 'note', 'rates_gbps')} if p else None)
PY
done
