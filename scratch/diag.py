import sys, os, ctypes as C; sys.path.insert(0,'.')
import numpy as np, torch
from pointcloudregistration_amd import capi, synth
capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libpcr_diag.so")
L = capi.lib()
n = 1000000
occ = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
halo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L.pcr_tune(2, halo)
src,tgt,Rt,tt = synth.registration_pair(n,n,20250001)
R0,T0 = synth.coarse_initial_transform(Rt,tt,n)
ctx = capi.Context(0)
six = capi.Index(ctx, capi.Cloud(ctx,src), kind=capi.INDEX_GRID, target_occupancy=occ)
tix = capi.Index(ctx, capi.Cloud(ctx,tgt), kind=capi.INDEX_GRID, target_occupancy=occ)
U = tix.unique_sq_dist()
pt = torch.zeros(17, dtype=torch.float64, device='cuda')
buf = (C.c_ulonglong*16)()
names = ["tiles","unfit_lanes","rows","x3","candidates","cyc_rows","cyc_stage","cyc_scan","cyc_total","undone_lanes","cyc_global","searches"]
for rep in range(2):
    L.pcr_diag_read_icp(buf, 1)
    capi.icp_step_d(ctx, six, tix, R0, T0, 4, U*36, U*4, pt, sample_begin=0, m=n); ctx.synchronize()
    L.pcr_diag_read_icp(buf, 1)
d = dict(zip(names, list(buf)))
w = max(1, d["searches"])
print("occ", six.info()["mean_occupancy"], "halo", halo, d)
print("per wave-search: tiles %.2f rows %.1f cand %.1f | cycles rows %.0f stage %.0f scan %.0f global %.0f total %.0f | undone lanes %.4f unfit lanes %.3f" % (
    d["tiles"]/w, d["rows"]/w, d["candidates"]/w, d["cyc_rows"]/w, d["cyc_stage"]/w, d["cyc_scan"]/w, d["cyc_global"]/w, d["cyc_total"]/w, d["undone_lanes"]/w, d["unfit_lanes"]/w))
