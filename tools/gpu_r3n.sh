R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for cfg in "ICL_ALIGNER_LANES=3" "ICL_ALIGNER_LANES=4 ICL_ALIGNER_GUIDED_MAP=1,2,4" "ICL_ALIGNER_LANES=4 ICL_ALIGNER_GUIDED_MAP=1,1,4 ICL_ALIGNER_OWN_MAP=2,2,3" "ICL_ALIGNER_LANES=3 ICL_ALIGNER_GUIDED_MAP=1,1,2 ICL_ALIGNER_OWN_MAP=1,1,3" "ICL_ALIGNER_LANES=3 ICL_ALIGNER_GUIDED_MAP=2,1,3 ICL_ALIGNER_OWN_MAP=1,2,3" "ICL_ALIGNER_LANES=3"; do
  echo "== $cfg"
  env $cfg python3 bench.py --no-cpu-baseline --no-exact-compare --no-kernel-timer --launch graph --steps 20 2>&1 | tail -1 | cut -c140-170
done
