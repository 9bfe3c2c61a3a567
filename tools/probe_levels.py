"""One ICP step and batched K-NN on a LiDAR-like pair (C5), nested-level index vs a single uniform grid.
Runs on the GPU box: python tools/probe_levels.py [points] [outlier fractions, comma separated]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudregistration_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
fracs = [float(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0.3]
ctx = capi.Context(0)
for of in fracs:
    src,tgt,Rm,t = synth.lidar_pair(n, 516, outlier_frac=of)
    cs, ct = capi.Cloud(ctx,src), capi.Cloud(ctx,tgt)
    R,T = synth.colmajor(Rm), t
    pt = torch.zeros(17, dtype=torch.float64, device='cuda')
    nq = min(n, 1000000)
    q = torch.from_numpy(src[:nq].copy()).cuda(); oi = torch.empty((nq,4), dtype=torch.int32, device='cuda')
    res = {}
    for ml in (0, 1):
        t0=time.time(); six = capi.Index(ctx, cs, kind=capi.INDEX_GRID, max_levels=ml); tix = capi.Index(ctx, ct, kind=capi.INDEX_GRID, max_levels=ml); ctx.synchronize(); tb=time.time()-t0
        print("outliers", of, "max_levels", ml, "build %.3f s" % tb, flush=True)
        i1, i2 = six.info(), tix.info()
        print("  src levels %d h0 %.3g maxocc %d | tgt levels %d h0 %.3g maxocc %d" % (i1['levels'], i1['cell_size'], i1['max_occupancy'], i2['levels'], i2['cell_size'], i2['max_occupancy']), flush=True)
        U = tix.unique_sq_dist()
        for rep in range(3):
            t0=time.time(); capi.icp_step_d(ctx, six, tix, R, T, 4, U*36, U*4, pt); ctx.synchronize(); dt=time.time()-t0
        print("  step %.4f s pairs %d  (gate radius %.3g)" % (dt, pt[0].item(), 6*np.sqrt(U)), flush=True)
        res[ml] = pt.cpu().numpy().copy()
        for rep in range(2):
            t0=time.time(); tix.knn_d(q, 4, oi); ctx.synchronize(); d1=time.time()-t0
        res[('knn', ml)] = oi.cpu().numpy().copy()
        for rep in range(2):
            t0=time.time(); tix.knn_d(q, 4, oi, max_d2=U*36); ctx.synchronize(); d2=time.time()-t0
        for rep in range(2):
            t0=time.time(); six.knn_d(q, 1, oi[:, :1].contiguous()); ctx.synchronize(); d3=time.time()-t0
        print("  knn %d: k4 unbounded %.4f s | k4 gated %.4f s | k1 self (src index) %.4f s" % (nq, d1, d2, d3), flush=True)
        six.close(); tix.close()
    print("  pairs equal", res[0][0] == res[1][0], "knn equal", (res[('knn',0)] == res[('knn',1)]).all(), flush=True)
