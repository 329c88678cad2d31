# usage: bash tools/pmc_winattn.sh <stage>   (two SQ counter passes, kernel-trace only, on tools/winattn_probe.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ST=$1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD --kernel-trace -d $R/gpurun_out/pmc_winattn${ST}_1 -o p --output-format csv -- python3 $R/tools/winattn_probe.py $ST > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/pmc_winattn${ST}_2 -o p --output-format csv -- python3 $R/tools/winattn_probe.py $ST > /dev/null 2>&1
rm -f $R/gpurun_out/pmc_winattn${ST}_*/p_kernel_trace.csv
