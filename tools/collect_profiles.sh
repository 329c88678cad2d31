# Round-end profile collection on the GPU box (ROUND=r6 by default): bench lines + rocprofv3 kernel stats of the same commands -> gpurun_out/${ROUND}_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
python3 -m pytest $R/tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/${ROUND:-r6}_gpu_tests_summary.txt
python3 $R/bench.py > $O/${ROUND:-r6}_bench.json 2> $O/${ROUND:-r6}_bench.err
python3 $R/bench.py --num-classes 16 --no-cpu-baseline > $O/${ROUND:-r6}_nc16_bench.json 2>/dev/null
python3 $R/bench.py --model swinunetr_icl --no-cpu-baseline > $O/${ROUND:-r6}_swin_bench.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --no-other-workloads --force-ddp 2>/dev/null | tail -1 > $O/${ROUND:-r6}_ddp_one_rank_bench.json
python3 $R/bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --feed 2>/dev/null | tail -1 > $O/${ROUND:-r6}_feed_bench.json
rocprofv3 --kernel-trace --stats -d $O/${ROUND:-r6}_prof_unet -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --launch graph > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/${ROUND:-r6}_prof_nc16 -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --launch graph --num-classes 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/${ROUND:-r6}_prof_swin -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --no-other-workloads --launch graph --model swinunetr_icl > /dev/null 2>&1
rm -f $O/${ROUND:-r6}_prof_unet/b_kernel_trace.csv $O/${ROUND:-r6}_prof_nc16/b_kernel_trace.csv $O/${ROUND:-r6}_prof_swin/b_kernel_trace.csv
cat $O/${ROUND:-r6}_gpu_tests_summary.txt; cut -c1-300 $O/${ROUND:-r6}_bench.json; cut -c1-200 $O/${ROUND:-r6}_nc16_bench.json; cut -c1-200 $O/${ROUND:-r6}_swin_bench.json; cut -c1-200 $O/${ROUND:-r6}_ddp_one_rank_bench.json; cut -c1-200 $O/${ROUND:-r6}_feed_bench.json
