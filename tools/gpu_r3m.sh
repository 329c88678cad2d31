cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace -d $O/r3m_trace -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 6 > /dev/null 2>&1
ls -la $O/r3m_trace
python3 $R/tools/timeline.py $O/r3m_trace/t_kernel_trace.csv 3
python3 $R/tools/timeline.py $O/r3m_trace/t_kernel_trace.csv 2
head -2 $O/r3m_trace/t_kernel_trace.csv | cut -c1-400
rm -f $O/r3m_trace/t_kernel_trace.csv
