#!/bin/bash
# The GPU-box tasks of this repository, one script (run through gpurun from the repository root, output under gpurun_out/):
#   tools/gpu_run.sh suite                          pytest -m gpu
#   tools/gpu_run.sh bench [bench.py arguments]     one bench line (ms per step, value)
#   tools/gpu_run.sh ab VAR v1 v2 [bench args]      bench A/B over an environment variable, two interleaved runs each
#   tools/gpu_run.sh stats NAME [bench args]        rocprofv3 --kernel-trace --stats of a graph-replayed bench -> gpurun_out/NAME_kernel_stats.csv,
#                                                   launches and kernel ms per step, the ATen / runtime kernels that are left
#   tools/gpu_run.sh critical-path                  tools/critical_path.py (stamp kernels inside the replayed graph)
#   tools/gpu_run.sh timeline                       kernel trace of the replayed step -> per-queue timeline (tools/timeline.py) -> gpurun_out/timeline.txt
#   tools/gpu_run.sh pmc                            PMC counters (two SQ passes, tools/pmc_conv.sh) and HBM traffic (tools/pmc_hbm.sh) of the convolution
#                                                   kernels on the layer shapes of the step, batch 2 -> gpurun_out/pmc_conv_raw.txt
#   tools/gpu_run.sh probe SRC.hip ARGS...          build a stand-alone probe of tools/probe/ and run it
# Several tasks in one call: separate them with "--".
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); O=$R/gpurun_out; mkdir -p "$O"
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['ms_per_step'], 'ms/step', d['value'], d['unit'])"; }
task() {
  local t=$1; shift
  case $t in
    suite) python -m pytest tests -m gpu -q -x 2>&1 | tail -${TAIL:-6} ;;
    bench) timeout 600 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads "$@" 2>/dev/null | line "bench $*" ;;
    ab) local var=$1 a=$2 b=$3; shift 3
        for rep in 1 2; do for v in "$a" "$b"; do
          # (a runtime knob that makes the replay hang must not hold the box until gpurun's own limit)
          env "$var=$v" timeout 300 python bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --steps 30 "$@" 2>/dev/null | line "$var=$v" || echo "$var=$v: no result (timeout or error)"
        done; done ;;
    stats) local name=$1; shift
        ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/${name}_prof" -o b --output-format csv -- python3 "$R/bench.py" \
            --no-cpu-baseline --no-exact-compare --no-other-workloads --no-kernel-timer --launch graph --steps 24 --warmup 3 "$@" > /dev/null 2>&1 )
        f=$(find "$O/${name}_prof" -name "*kernel_stats.csv" | head -1)
        cp "$f" "$O/${name}_kernel_stats.csv"
        python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
import collections
cnt = collections.Counter(int(r['Calls']) for r in rows if int(r['Calls']) >= 10)
steps = min(k for k, v in cnt.items() if v >= 3)      # the call count of the once-per-step kernels: the smallest count several kernels share
print('launches/step', round(sum(int(r['Calls']) for r in rows) / steps, 1), 'kernel ms/step', round(sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6, 3))
left = [(int(r['Calls']) / steps, r['Name']) for r in rows if 'at::' in r['Name'] or 'rocclr' in r['Name']]
print('ATen / runtime launches per step:', round(sum(n for n, _ in left), 1))
for n, name in sorted(left, reverse=True)[:12]:
    print(f'  {n:6.1f}  {name[:120]}')
PY
        ;;
    critical-path) python tools/critical_path.py 2>&1 | tail -${TAIL:-24} ;;
    timeline)
        ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/tl_prof" -o b --output-format csv -- python3 "$R/bench.py" \
            --no-cpu-baseline --no-exact-compare --no-other-workloads --launch graph --steps 8 --warmup 2 > /dev/null 2>&1 )
        T=$(find "$O/tl_prof" -name "*kernel_trace.csv" | head -1)
        python3 tools/timeline.py "$T" 3 0.25 > "$O/timeline.txt" 2>&1
        python3 tools/queue_tail.py "$T" 70 > "$O/queue_tails.txt" 2>&1      # the tails of the backward chains, per hardware queue
        python3 tools/step_trace.py "$T" 2 > "$O/step_trace.txt" 2>&1                   # every kernel of one replayed step
        rm -f "$T"; head -60 "$O/timeline.txt" ;;
    pmc)
        for spec in "f16 16 16 96 fwd" "f48 48 16 96 fwd" "f32 32 32 48 fwd" "f4848 48 48 96 fwd" "w16 16 16 96 wgrad" "w32 32 32 48 wgrad" \
                    "f24a 32 64 24 fwd" "f24b 192 64 24 fwd" "f12a 128 128 12 fwd" "f12b 384 128 12 fwd" "f6 256 256 6 fwd"; do
          set -- $spec; tag=$1; shift
          bash tools/pmc_conv.sh "$tag" -- "$1" "$2" "$3" "$4" 5 2
          bash tools/pmc_hbm.sh "$tag" conv "$1" "$2" "$3" "$4" 5 2
        done
        python3 tools/pmc_summary.py "$O"/pmc_*_1 "$O"/pmc_*_2 "$O"/hbm_*_f "$O"/hbm_*_w > "$O/pmc_conv_raw.txt" 2>&1
        cut -c1-400 "$O/pmc_conv_raw.txt" ;;
    probe) local src=$1; shift
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icl_amd/csrc -I tools/probe "tools/probe/$src" -o "/tmp/${src%.hip}" && "/tmp/${src%.hip}" "$@" ;;
    *) echo "unknown task $t"; return 1 ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then task "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && task "${args[@]}"
