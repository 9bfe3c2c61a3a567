#!/usr/bin/env python3
"""What would regrouping a wave's samples by their phase-2 trip counts buy the filtered step kernel?
Needs the diagnostic build (make -C pointcloudregistration_amd/csrc diag; PCR_LIB=.../libpcr_diag.so), which records
every sample's forward and reverse trips of the last launch (pcr_diag_read_trips).  Prints, for iterations 5, 6 and 12 of
the 1 M x 1 M fit: the mean over waves of the slowest lane's trips as the kernel runs now, with the samples of 4 / 16 / 64
neighbouring tiles sorted by their CURRENT counts (the ideal), sorted by the counts of an EARLIER iteration (what a
kernel could know), and with 2 / 4 / 8 neighbouring lanes sharing their rows evenly.
"""
import sys, ctypes as C, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudregistration_amd import capi, synth
N = 1_000_000
src, tgt, Rm, t = synth.registration_pair(N, N, 20250001)
ctx = capi.Context(0)
six = capi.Index(ctx, capi.Cloud(ctx, src), kind=capi.INDEX_GRID)
tix = capi.Index(ctx, capi.Cloud(ctx, tgt), kind=capi.INDEX_GRID)
U = tix.unique_sq_dist()
R0, T0 = synth.coarse_initial_transform(Rm, t, N)
fit = capi.Fit(ctx, six, tix, 4, U * 36.0, U * 4.0, conv=None, unique_sq_dist=U, traj_cap=40)
fit.reset(R0, T0)
L = capi.lib()
buf = np.zeros((2, 1 << 20), np.uint8)
prev = None
for it in range(1, 13):
    fit.iterate(); ctx.synchronize()
    if it in (5, 6, 12):
        L.pcr_diag_read_trips(buf.ctypes.data_as(C.c_void_p))
        tr = buf[:, :N].astype(np.int32)
        for name, x in (("fwd", tr[0]), ("rev", tr[1]), ("fwd+rev", tr[0] + tr[1])):
            n64 = (N // 64) * 64
            w = x[:n64].reshape(-1, 64)
            base = w.max(1).sum()
            line = f"it {it} {name}: lane mean {x.mean():.2f}, wave-max mean {w.max(1).mean():.2f}"
            for g in (4, 16, 64):
                ng = (n64 // (64 * g)) * 64 * g
                grp = np.sort(x[:ng].reshape(-1, 64 * g), axis=1).reshape(-1, 64)
                line += f"; sorted within {g} tiles: {grp.max(1).mean():.2f}"
            print(line, flush=True)
        # lanes SHARING their work: 2 / 4 / 8 neighbouring lanes split the sum of their trips evenly (rounded up)
        n64 = (N // 64) * 64
        for share in (2, 4, 8):
            f = tr[0][:n64].reshape(-1, 64 // share, share).sum(2)
            r = tr[1][:n64].reshape(-1, 64 // share, share).sum(2)
            fm = -(-f // share); rm = -(-r // share)
            print(f"   it {it}: {share} lanes sharing their rows: max(fwd)+max(rev) per wave {(fm.max(1) + rm.max(1)).mean():.2f} "
                  f"(now {(tr[0][:n64].reshape(-1, 64).max(1) + tr[1][:n64].reshape(-1, 64).max(1)).mean():.2f})", flush=True)
        if prev is not None:
            # how well does last look's count predict this one's?  regroup by the PREVIOUS counts, evaluate with the current
            for g in (4, 16):
                ng = (N // (64 * g)) * 64 * g
                x = (tr[0] + tr[1])[:ng].reshape(-1, 64 * g); pv = prev[:ng].reshape(-1, 64 * g)
                order = np.argsort(pv, axis=1, kind="stable")
                re = np.take_along_axis(x, order, axis=1).reshape(-1, 64)
                print(f"   fwd+rev regrouped within {g} tiles by the counts of the earlier look: wave-max mean {re.max(1).mean():.2f}")
            # separate maxima (the two searches run one after the other: the wave pays max(fwd) + max(rev))
            for g in (4, 16):
                ng = (N // (64 * g)) * 64 * g
                pv = prev[:ng].reshape(-1, 64 * g); order = np.argsort(pv, axis=1, kind="stable")
                f = np.take_along_axis(tr[0][:ng].reshape(-1, 64 * g), order, axis=1).reshape(-1, 64)
                r = np.take_along_axis(tr[1][:ng].reshape(-1, 64 * g), order, axis=1).reshape(-1, 64)
                print(f"   ... max(fwd)+max(rev) per wave: {(f.max(1) + r.max(1)).mean():.2f} (now: {(tr[0][:ng].reshape(-1,64).max(1) + tr[1][:ng].reshape(-1,64).max(1)).mean():.2f})")
        prev = (tr[0] + tr[1]).copy()
