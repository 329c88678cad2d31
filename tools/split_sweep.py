import os, sys, subprocess
for ns in (2,3,4,5,6,7,8,9,10,12,14):
    env=dict(os.environ, ICL_GEMM_SPLIT=str(ns))
    out=subprocess.run([sys.executable,"tools/gemm_probe.py","one"],env=env,capture_output=True,text=True).stdout
    print("ns",ns,out.strip().splitlines()[-1] if out.strip() else "")
