#!/bin/bash
# round 4: PMC counters (two SQ passes) and HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of the convolution kernels, batch 2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for spec in "f16 16 16 96 fwd" "f48 48 16 96 fwd" "f32 32 32 48 fwd" "f4848 48 48 96 fwd" "w16 16 16 96 wgrad" "w32 32 32 48 wgrad"; do
  set -- $spec; tag=$1; shift
  bash $R/tools/pmc_conv.sh r4_$tag -- $1 $2 $3 $4 5 2
  bash $R/tools/pmc_hbm.sh r4_$tag conv $1 $2 $3 $4 5 2
done
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_r4_*_1 gpurun_out/pmc_r4_*_2 gpurun_out/hbm_r4_*_f gpurun_out/hbm_r4_*_w > gpurun_out/r4_pmc_conv_raw.txt 2>&1
cat gpurun_out/r4_pmc_conv_raw.txt | cut -c1-400
