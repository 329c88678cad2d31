cd $GRAFT_REPO_ROOT; O=gpurun_out
python3 -m pytest tests -m gpu -q > $O/r3f_tests.log 2>&1; echo "tests rc=$?" >> $O/r3f_tests.log
grep -E "passed|failed|^FAILED|rc=" $O/r3f_tests.log | tail -12
python3 bench.py --no-cpu-baseline --no-exact-compare --feed > $O/r3f_bench_feed.json 2> $O/r3f_bench_feed.err; python3 -c "
import json; d=json.loads(open('$O/r3f_bench_feed.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('feed'))"
bash tools/pmc_conv.sh f16 -- 16 16 96 fwd 5 2
bash tools/pmc_conv.sh w16 -- 16 16 96 wgrad 5 2
bash tools/pmc_conv.sh w32 -- 32 32 48 wgrad 5 2
bash tools/pmc_conv.sh w96 -- 96 32 48 wgrad 5 2
bash tools/pmc_hbm.sh f16 conv 16 16 96 fwd 3 2
bash tools/pmc_hbm.sh w16 conv 16 16 96 wgrad 3 2
bash tools/pmc_hbm.sh w32 conv 32 32 48 wgrad 3 2
bash tools/pmc_hbm.sh w48 conv 48 16 96 wgrad 3 2
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $O/pmc_f16_1 $O/pmc_f16_2 $O/pmc_w16_1 $O/pmc_w16_2 $O/pmc_w32_1 $O/pmc_w32_2 $O/pmc_w96_1 $O/pmc_w96_2 $O/hbm_f16_f $O/hbm_f16_w $O/hbm_w16_f $O/hbm_w16_w $O/hbm_w32_f $O/hbm_w32_w $O/hbm_w48_f $O/hbm_w48_w > $O/r3f_pmc_summary.txt 2>&1
cat $O/r3f_pmc_summary.txt | cut -c1-400
