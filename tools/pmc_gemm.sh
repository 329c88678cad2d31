cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD --kernel-trace -d $R/gpurun_out/pmc_gemm1 -o p --output-format csv -- python3 $R/tools/gemm_one.py 128 13824 13824 fwd 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/pmc_gemm2 -o p --output-format csv -- python3 $R/tools/gemm_one.py 128 13824 13824 fwd 2 > /dev/null 2>&1
ls $R/gpurun_out/pmc_gemm1 $R/gpurun_out/pmc_gemm2
