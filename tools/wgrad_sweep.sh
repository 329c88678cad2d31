# A/B of the weight-gradient kernels on the layer shapes of the 3-D U-Net (batch 2): ICL_WGRAD_ROWS = 0 (conv_mfma_static.h) / a4 a2 b4 c2
for shape in "32 32 48" "96 32 48" "16 32 48" "64 64 24" "192 64 24"; do
  for mode in 0 a4 a2 c4 c2; do
    ICL_WGRAD_ROWS=$mode python tools/conv_one.py $shape wgrad 5 2 2>&1 | grep TFLOP | sed "s/^/rows=$mode  /" | cut -c1-150
  done
done
