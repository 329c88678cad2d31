#!/bin/bash
# round 4, run p: which convolution path does the drop-in test's ICL_CONV_SPLIT=0 case run?  + the rest of the GPU suite
mkdir -p gpurun_out
python tests/diag/mlp2_grad_sensitivity.py > gpurun_out/r4p_sens.txt 2>&1
python -m pytest tests/test_gpu_dropin.py -q -x -k "unet3d_icl_through_compat_root and 0-0.003" 2>&1 | tail -5 >> gpurun_out/r4p_sens.txt
python -m pytest tests -m gpu -q -s --deselect "tests/test_gpu_dropin.py::test_reference_loop_body_unet3d_icl_through_compat_root" 2>&1 | tail -30 > gpurun_out/r4p_gpu_tests.txt
cat gpurun_out/r4p_sens.txt; tail -12 gpurun_out/r4p_gpu_tests.txt
