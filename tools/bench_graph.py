#!/usr/bin/env python3
"""The device-resident fit of bench.py (C3: 1 M x 1 M, K=4, criteria off) enqueued three ways -- one launch per
iteration from the host, the same with the step kernel bracketed by HIP events (what bench.py's timed region does), and
all iterations as ONE HIP graph launch (pcr_fit_enqueue_iterations) -- wall time per iteration each.
    python3 tools/bench_graph.py [--points 1000000] [--steps 30] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from pointcloudregistration_amd import capi, synth
    n = a.points
    src, tgt, Rt, tt = synth.registration_pair(n, n, 20250001)
    ctx = capi.Context(0)   # its own stream (the legacy default stream cannot be captured)
    six = capi.Index(ctx, capi.Cloud(ctx, torch.from_numpy(src).cuda()), kind=capi.INDEX_GRID)
    tix = capi.Index(ctx, capi.Cloud(ctx, torch.from_numpy(tgt).cuda()), kind=capi.INDEX_GRID)
    U = tix.unique_sq_dist()
    R0, T0 = synth.coarse_initial_transform(Rt, tt, n)
    fit = capi.Fit(ctx, six, tix, 4, U * 36.0, U * 4.0, conv=None, unique_sq_dist=U, traj_cap=a.steps)
    out = {}

    def run(name, enqueue, profiling):
        ts, res = [], None
        for rep in range(a.reps + 2):
            fit.reset(R0, T0)
            ctx.set_profiling(profiling, classes=(1,) if profiling else None)
            ctx.synchronize()
            t0 = time.perf_counter()
            enqueue()
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) / a.steps)
            if profiling:
                ctx.read_kernel_times()
            res = fit.read()
            assert res["iterations"] == a.steps
        ctx.set_profiling(False)
        out[name] = {"us_per_iteration": float(np.median(ts[2:])) * 1e6, "iterations_per_s": 1.0 / float(np.median(ts[2:]))}
        return res

    def plain():
        for _ in range(a.steps):
            fit.iterate()
    r1 = run("one launch per iteration", plain, False)
    r2 = run("one launch per iteration, step kernel bracketed by HIP events", plain, True)
    r3 = run("one HIP graph launch for all iterations", lambda: fit.iterate_many(a.steps), False)
    for r in (r2, r3):
        assert (r["R"] == r1["R"]).all() and (r["T"] == r1["T"]).all() and (r["traj"] == r1["traj"]).all()
    out["workload"] = f"C3 {n}x{n}, K=4, {a.steps} iterations per fit, median of {a.reps} fits; results bit-identical across the three"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
