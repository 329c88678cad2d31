# PMC + HBM traffic of the round-3 final forward kernel variants (V = 60)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
bash $R/tools/pmc_conv.sh f16v60 -- 16 16 96 fwd 5 2
bash $R/tools/pmc_conv.sh f48v60 -- 48 48 96 fwd 5 2
bash $R/tools/pmc_conv.sh f32v60 -- 32 32 48 fwd 5 2
bash $R/tools/pmc_hbm.sh f16v60 conv 16 16 96 fwd 3 2
cd $R
python3 tools/pmc_summary.py $O/pmc_f16v60_1 $O/pmc_f16v60_2 $O/pmc_f48v60_1 $O/pmc_f48v60_2 $O/pmc_f32v60_1 $O/pmc_f32v60_2 $O/hbm_f16v60_f $O/hbm_f16v60_w > $O/r3_pmc_v60_raw.txt 2>&1
cat $O/r3_pmc_v60_raw.txt | cut -c1-400
