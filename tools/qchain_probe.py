#!/usr/bin/env python3
"""Per-launch time of the fused query-chain stages (csrc/kernels/qchain.h) at the shapes of the 3-D U-Net ICL's three levels, alone on the
chip (HIP events around 20 back-to-back launches, eager).   python tools/qchain_probe.py [nc] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import _lib, ops  # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
R = B * nc
dp = ((0, 0, 1.0), (0, 0, 1.0), None)
print("# library", _lib.lib_path(), "rows", R)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for C in (256, 128, 64):
    H = 4 * C
    r = lambda *s: torch.randn(*s, device=dev)      # noqa: E731
    x, xh, rs, g, b = r(R, C), r(R, C), r(R).abs() + 0.5, r(C), r(C)
    wcc, wch, whc, wq = r(C, C) * 0.05, r(C, H) * 0.05, r(H, C) * 0.05, r(C // 2, C) * 0.05
    u, yC, yH, yq, so0, so1, so2 = r(R, H), r(R, C), r(R, H), r(R, C // 2), r(R, C), r(R), r(R, C)
    soH = r(R, H)
    rows = [
        ("F1 LN + fc_q          (fwd C->C)", lambda: ops._qc_stage(x, dp, y=yC, R=R, K=C, N=C, nc=nc, x=x, w=wcc, bias=b, pro=ops.QC_PRO_LN, pa=g, pb=b, so0=so0, so1=so1, so2=so2)),
        ("F2 proj, x(1+f)        (fwd C->C)", lambda: ops._qc_stage(x, dp, y=yC, R=R, K=C, N=C, nc=nc, x=x, w=wcc, bias=b, epi=ops.QC_EPI_DP1)),
        ("F3 LN + fc1           (fwd C->4C)", lambda: ops._qc_stage(x, dp, y=yH, R=R, K=C, N=H, nc=nc, x=x, w=whc, bias=r(H), pro=ops.QC_PRO_LN, pa=g, pb=b, so0=so0, so1=so1, so2=so2)),
        ("F4 GELU + fc2 + res   (fwd 4C->C)", lambda: ops._qc_stage(x, dp, y=yC, R=R, K=H, N=C, nc=nc, x=u, w=wch, bias=b, pro=ops.QC_PRO_GELU, so2=soH, epi=ops.QC_EPI_RES_DP, ea=x)),
        ("F5 query_convs        (fwd C->C/2)", lambda: ops._qc_stage(x, dp, y=yq, R=R, K=C, N=C // 2, nc=nc, x=x, w=wq, bias=r(C // 2))),
        ("B1 dq2 = dnxt Wqc     (bwd C/2->C)", lambda: ops._qc_stage(x, dp, y=yC, R=R, K=C // 2, N=C, trans=1, nc=nc, x=yq, w=wq, epi=ops.QC_EPI_ADDROWS, ea=x, ea_rows=(0, nc))),
        ("B2 du = g2 W2 gelu'   (bwd C->4C)", lambda: ops._qc_stage(x, dp, y=yH, R=R, K=C, N=H, trans=1, nc=nc, x=x, w=wch, pro=ops.QC_PRO_SCALE, so2=so2, epi=ops.QC_EPI_GELUBWD, ea=u)),
        ("B3 dn2 = du W1        (bwd 4C->C)", lambda: ops._qc_stage(x, dp, y=yC, R=R, K=H, N=C, trans=1, nc=nc, x=u, w=whc)),
        ("B4 LN' + proj dgrad   (bwd C->C)", lambda: ops._qc_stage(x, dp, y=yC, R=R, K=C, N=C, trans=1, nc=nc, x=x, w=wcc, pro=ops.QC_PRO_LNBWD, pa=xh, pb=rs, pc=g, pd=x, pro_dp=1, so2=so2)),
        ("B6 LN' only           (rows)", lambda: ops._qc_stage(x, dp, y=yC, R=R, K=C, trans=2, nc=nc, x=x, pro=ops.QC_PRO_LNBWD, pa=xh, pb=rs, pc=g)),
    ]
    gw = [r(C, C), r(C), r(C), r(C), r(H, C), r(H), r(C, H), r(C), r(C // 2, C), r(C // 2), r(C), r(C), r(C, C), r(C)]
    jobs = [(x, x, gw[0], gw[1], R, C, C, 0), (x, xh, gw[2], gw[3], R, C, C, 1), (u, x, gw[4], gw[5], R, H, C, 0), (x, u, gw[6], gw[7], R, C, H, 0),
            (yq, x, gw[8], gw[9], R, C // 2, C, 0), (x, xh, gw[10], gw[11], R, C, C, 1), (x, x, gw[12], gw[13], R, C, C, 0)]
    rows.append(("W  all parameter gradients (7 jobs)", lambda: ops._qc_wgrad(x, jobs)))
    print(f"C = {C}")
    for name, fn in rows:
        print(f"   {name:40s} {timed(fn):7.1f} us")
