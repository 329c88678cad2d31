cd $GRAFT_REPO_ROOT
O=gpurun_out/r6z; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -k "as_accurate" 2>&1 | tail -2 | tee $O/acc.txt
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'], d['config'].get('launch'))" || echo "$* failed"; }
for rep in 1 2; do
  run X=0
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run DEBUG_HIP_GRAPH_BATCH_SIZE=16
  run DEBUG_HIP_GRAPH_BATCH_SIZE=256
  run DEBUG_HIP_DYNAMIC_QUEUES=0
  run DEBUG_HIP_DYNAMIC_QUEUES=1
  run GPU_STREAMOPS_CP_WAIT=0
  run GPU_STREAMOPS_CP_WAIT=1
  run AMD_DIRECT_DISPATCH=0
  run HIP_FORCE_DEV_KERNARG=0
  run HIP_FORCE_DEV_KERNARG=1
  run DEBUG_HIP_KERNARG_COPY_OPT=0
  run ROC_ACTIVE_WAIT_TIMEOUT=0
done 2>&1 | tee $O/runtime_knobs.txt
