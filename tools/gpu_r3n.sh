R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for cfg in "ICL_ALIGNER_LANES=3" "ICL_ALIGNER_LANES=2" "ICL_ALIGNER_LANES=1" "ICL_ALIGNER_LANES=3 ICL_ALIGNER_LANE_MASK=1" "ICL_ALIGNER_LANES=3 ICL_ALIGNER_LANE_MASK=2" "ICL_ALIGNER_LANES=2 ICL_ALIGNER_LANE_MASK=2"; do
  echo "== $cfg"
  env $cfg python3 tools/aligner_probe.py 2>&1 | tail -1 | cut -c45-140
  env $cfg python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20 2>&1 | tail -1 | cut -c140-170
done
