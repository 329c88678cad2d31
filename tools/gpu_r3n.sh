cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace -d $O/r3n_trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 6 > /dev/null 2>&1
python3 $R/tools/timeline.py $O/r3n_trace/t_kernel_trace.csv 3 100 0 1.6 15 | grep '^   '
python3 $R/tools/timeline.py $O/r3n_trace/t_kernel_trace.csv 3 100 12.3 16 15 | grep '^   '
rm -f $O/r3n_trace/t_kernel_trace.csv
