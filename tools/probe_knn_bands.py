"""K-NN time by range band of a LiDAR-like scan, nested-level index vs a single uniform grid (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudregistration_amd import capi, synth
n = 1_000_000
ctx = capi.Context(0)
src,tgt,Rm,t = synth.lidar_pair(n, 516, outlier_frac=0.0)
ct = capi.Cloud(ctx,tgt)
ixs = {ml: capi.Index(ctx, ct, kind=capi.INDEX_GRID, max_levels=ml) for ml in (0,1)}
print(ixs[0].info(), flush=True)
U = ixs[0].unique_sq_dist()
r = np.hypot(src[:,0], src[:,1])
def run(name, qn):
    if len(qn) < 64: return
    q = torch.from_numpy(np.ascontiguousarray(qn)).cuda(); oi = torch.empty((len(qn),4), dtype=torch.int32, device='cuda')
    out = []
    for ml in (0,1):
        for md in (-1.0, U*36):
            for rep in range(2):
                ctx.synchronize(); t0=time.time(); ixs[ml].knn_d(q, 4, oi, max_d2=md); ctx.synchronize(); dt=time.time()-t0
            out.append(dt*1e3)
    print("%-28s n=%7d nested: unb %8.3f gated %8.3f ms | flat: unb %8.3f gated %8.3f ms" % (name, len(qn), *out), flush=True)
for lo,hi in ((0,2),(2,4),(4,8),(8,16),(16,32),(32,64),(64,200)):
    m = (r>=lo)&(r<hi)
    run("src r[%d,%d) ground" % (lo,hi), src[m & (src[:,2] < 0.3)])
    run("src r[%d,%d) walls" % (lo,hi), src[m & (src[:,2] >= 0.3)])
run("tgt itself", tgt[:200000])
run("tgt lifted 0.5", tgt[:200000] + [0,0,0.5])
run("all src", src)
