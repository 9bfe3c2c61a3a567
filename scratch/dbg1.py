import sys; sys.path.insert(0,'.')
import numpy as np, torch, time
from pointcloudregistration_amd import capi, synth
from oracle import oracle as O
for n in (100000, 1000000):
    src,tgt,Rt,tt=synth.registration_pair(n,n,20250001)
    R0,T0=synth.coarse_initial_transform(Rt,tt,n)
    ctx=capi.Context(0)
    six=capi.Index(ctx,capi.Cloud(ctx,src),kind=capi.INDEX_GRID); tix=capi.Index(ctx,capi.Cloud(ctx,tgt),kind=capi.INDEX_GRID)
    print(n, six.info(), tix.info())
    U=tix.unique_sq_dist(); print("U",U, O.uniq_sq_dist(tgt), synth.mean_spacing(n)**2)
    t=time.time(); out=capi.icp_step(ctx,six,tix,R0,T0,4,U*36,U*4,exact=False); print("step",time.time()-t, out['partials'][:4], out['keep'].sum())
    # device path
    pt=torch.zeros(17,dtype=torch.float64,device='cuda')
    capi.icp_step_d(ctx,six,tix,R0,T0,4,U*36,U*4,pt,sample_begin=0,m=n); ctx.synchronize(); print("d", pt[:2])
    # wrapped
    st=torch.from_numpy(src).cuda(); tt_=torch.from_numpy(tgt).cuda()
    ctx2=capi.Context(0,stream=torch.cuda.current_stream())
    six2=capi.Index(ctx2,capi.Cloud(ctx2,st),kind=capi.INDEX_GRID); tix2=capi.Index(ctx2,capi.Cloud(ctx2,tt_),kind=capi.INDEX_GRID)
    print(six2.info()['mean_occupancy'], tix2.unique_sq_dist())
    capi.icp_step_d(ctx2,six2,tix2,R0,T0,4,U*36,U*4,pt,sample_begin=0,m=n); torch.cuda.synchronize(); print("d2", pt[:2])
