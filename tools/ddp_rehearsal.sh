#!/bin/bash
# Rehearsal of the 2- / 4- / 8-rank data-parallel step on ONE GPU: N processes share device 0 over gloo (ICL_BENCH_SHARE_GPU=1; RCCL
# refuses two ranks on one device).  It exercises the code path the N-GPU bench will take — self-launch, rendezvous, calibrate(), the three
# / four hipGraphs per rank, eager collectives between the replays, row-sharded updates with their overlapped all-gathers (nc = 16,
# forced by ICL_DDP_SHARD_ROWS since gloo's measured rates would never choose them), MAX-over-ranks timing, the plan and (round 6)
# config.ddp_phases on the JSON line.  Its TIMINGS MEAN NOTHING (the ranks time-share one GPU, collectives stage through the host).
# Then the one-rank RCCL-group step of each workload with its projection to 8 ranks (bench.py --project-world 8; round 6: with the
# break-even link rate).  Run from the repository root on the GPU box; outputs under gpurun_out/ (ROUND=r6 by default).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; R=${ROUND:-r6}; mkdir -p $O
export ICL_BENCH_SHARE_GPU=1 ICL_BENCH_BACKEND=gloo
for n in 2 4 8; do
  timeout 1500 python bench.py --gpus $n --steps 3 --warmup 2 --no-cpu-baseline --no-exact-compare --no-kernel-timer --no-other-workloads > $O/${R}_ddp_rehearsal_${n}_ranks_nc2.json 2> $O/${R}_ddp_rehearsal_${n}_ranks_nc2.err
  echo "ranks=$n nc=2 rc=$?"; tail -c 400 $O/${R}_ddp_rehearsal_${n}_ranks_nc2.json
done
ICL_DDP_SHARD_ROWS=768 timeout 2400 python bench.py --gpus 8 --steps 3 --warmup 2 --num-classes 16 --no-cpu-baseline --no-exact-compare --no-kernel-timer --no-other-workloads > $O/${R}_ddp_rehearsal_8_ranks_nc16.json 2> $O/${R}_ddp_rehearsal_8_ranks_nc16.err
echo "ranks=8 nc=16 rc=$?"; tail -c 400 $O/${R}_ddp_rehearsal_8_ranks_nc16.json
unset ICL_BENCH_SHARE_GPU ICL_BENCH_BACKEND
for nc in 2 16; do
  timeout 900 python bench.py --gpus 1 --force-ddp --project-world 8 --num-classes $nc --steps 20 --no-cpu-baseline --no-exact-compare --no-kernel-timer --no-other-workloads > $O/${R}_ddp_one_rank_projection_nc$nc.json 2> $O/${R}_ddp_one_rank_projection_nc$nc.err
  python - $O/${R}_ddp_one_rank_projection_nc$nc.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
p = d['config']['ddp_plan'].get('projection')
print('one-rank group', d['ms_per_step'], 'ms;', {k: p[k] for k in p if k not in ('matrices', 'note', 'rates_gbps')} if p else None)
PY
done
