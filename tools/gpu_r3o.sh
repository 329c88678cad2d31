R=$GRAFT_REPO_ROOT
cd $R
for V in 60; do
ICL_HIP_LIB=$R/gpurun_in/libicl_dbg16.so ICL_CONV_SPLIT_V=$V python3 tools/bf3_stamps.py 16 16 96 | grep -v 'item [34]'
ICL_HIP_LIB=$R/gpurun_in/libicl_dbg16.so ICL_CONV_SPLIT_V=$V python3 tools/bf3_stamps.py 32 32 48 | grep -v 'item [34]'
done
python3 tools/conv_ab.py --shapes "16,16,96,fwd;48,16,96,fwd;16,48,96,fwd;32,32,48,fwd;96,32,48,fwd;48,48,96,fwd" --var ICL_CONV_SPLIT_V=24 --var ICL_CONV_SPLIT_V=56 --var ICL_CONV_SPLIT_V=60
