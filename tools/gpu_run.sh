#!/bin/bash
# The GPU-box tasks of this repository, one script (run through gpurun from the repository root, output under gpurun_out/):
#   tools/gpu_run.sh suite                          pytest -m gpu
#   tools/gpu_run.sh bench [bench.py arguments]     one bench line (ms per step, value)
#   tools/gpu_run.sh ab VAR v1 v2 [bench args]      bench A/B over an environment variable, two interleaved runs each
#   tools/gpu_run.sh stats NAME [bench args]        rocprofv3 --kernel-trace --stats of a graph-replayed bench -> gpurun_out/NAME_kernel_stats.csv,
#                                                   launches and kernel ms per step, the ATen / runtime kernels that are left
#   tools/gpu_run.sh critical-path                  tools/critical_path.py (stamp kernels inside the replayed graph)
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
    bench) python bench.py --no-cpu-baseline --no-exact-compare "$@" 2>/dev/null | line "bench $*" ;;
    ab) local var=$1 a=$2 b=$3; shift 3
        for rep in 1 2; do for v in "$a" "$b"; do
          env "$var=$v" python bench.py --no-cpu-baseline --no-exact-compare --steps 30 "$@" 2>/dev/null | line "$var=$v"
        done; done ;;
    stats) local name=$1; shift
        ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$O/${name}_prof" -o b --output-format csv -- python3 "$R/bench.py" \
            --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 24 --warmup 3 "$@" > /dev/null 2>&1 )
        f=$(find "$O/${name}_prof" -name "*kernel_stats.csv" | head -1)
        cp "$f" "$O/${name}_kernel_stats.csv"
        python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
import collections
steps = collections.Counter(int(r['Calls']) for r in rows if int(r['Calls']) >= 10).most_common(1)[0][0]      # once-per-step kernels: the most common call count
print('launches/step', round(sum(int(r['Calls']) for r in rows) / steps, 1), 'kernel ms/step', round(sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6, 3))
left = [(int(r['Calls']) / steps, r['Name']) for r in rows if 'at::' in r['Name'] or 'rocclr' in r['Name']]
print('ATen / runtime launches per step:', round(sum(n for n, _ in left), 1))
for n, name in sorted(left, reverse=True)[:12]:
    print(f'  {n:6.1f}  {name[:120]}')
PY
        ;;
    critical-path) python tools/critical_path.py 2>&1 | tail -24 ;;
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
