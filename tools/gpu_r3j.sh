cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 -m pytest tests -m gpu -q -x > $O/r3j_tests.log 2>&1; echo "tests rc=$?" >> $O/r3j_tests.log
grep -E "passed|failed|^FAILED|rc=" $O/r3j_tests.log | tail -6
python3 bench.py --no-cpu-baseline --no-exact-compare > $O/r3j_bench.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/r3j_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['avg_launch_us'], r['all_conv'])"
python3 bench.py --model swinunetr_icl --no-cpu-baseline --no-exact-compare > $O/r3j_swin.json 2>/dev/null; cut -c1-200 $O/r3j_swin.json
