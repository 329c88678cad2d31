# usage (GPU box): bash tools/kernel_coverage.sh   -> gpurun_out/kernel_coverage.txt
# Which kernels of libicl_hip.so does the GPU suite + the three bench configurations launch?  (rocprofv3 kernel trace, names only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cov
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/suite -o c --output-format csv -- python3 -m pytest tests -m gpu -q -p no:cacheprovider > $O/suite.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/unet -o c --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/unet16 -o c --output-format csv -- python3 bench.py --num-classes 16 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/swin -o c --output-format csv -- python3 bench.py --model swinunetr_icl --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/refloop -o c --output-format csv -- python3 bench.py --loop reference --steps 3 --warmup 2 > /dev/null 2>&1
python3 tools/kernel_coverage.py $O > $R/gpurun_out/kernel_coverage.txt 2>&1
tail -2 $O/suite.log
find $O -name '*kernel_trace.csv' -delete      # tens of MB each; the stats CSVs are what the summary reads
