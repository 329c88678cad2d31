# Round-end profile collection on the GPU box: bench lines + rocprofv3 kernel stats of the same commands -> gpurun_out/r3_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/bench.py > $O/r3_bench.json 2> $O/r3_bench.err
python3 $R/bench.py --num-classes 16 --no-cpu-baseline > $O/r3_nc16_bench.json 2>/dev/null
python3 $R/bench.py --model swinunetr_icl > $O/r3_swin_bench.json 2>/dev/null
python3 $R/bench.py --no-cpu-baseline --force-ddp 2>/dev/null | tail -1 > $O/r3_ddp_one_rank_bench.json
rocprofv3 --kernel-trace --stats -d $O/r3_prof_unet -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --launch graph > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/r3_prof_swin -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --launch graph --model swinunetr_icl > /dev/null 2>&1
rm -f $O/r3_prof_unet/b_kernel_trace.csv $O/r3_prof_swin/b_kernel_trace.csv
cut -c1-300 $O/r3_bench.json; cut -c1-200 $O/r3_nc16_bench.json; cut -c1-200 $O/r3_swin_bench.json; cut -c1-200 $O/r3_ddp_one_rank_bench.json
