R=$GRAFT_REPO_ROOT
cd $R
for shape in "16 16 96" "48 16 96" "32 32 48" "48 48 96"; do python3 tools/planes_probe.py $shape 2>&1 | tail -1; done
for shape in "16 16 96" "48 48 96"; do python3 tools/planes_probe.py $shape 2>&1 | tail -1; done
