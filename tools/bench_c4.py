#!/usr/bin/env python3
"""C4 (BASELINE.json: a scan set registered pair by pair -- many small ICP problems): fits per second of three ways to
run the same 14 pairwise fits of 15 synthetic partial scans of 50-80 k points (synth.scan_sequence; the Happy Buddha
scans themselves are not in the reference tree), 30 iterations each with the reference's convergence criteria:
  sequential    one device-resident fit after the other (one launch per iteration each)
  streams       one context (stream) per pair, iterations enqueued round-robin (round 1's scheme)
  batch         pcr_fit_batch: ONE launch per iteration for all pairs
Index builds are outside the timed regions (reported separately).  Every scheme must give identical results.
    python tools/bench_c4.py            (on the GPU box, from the repo root)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudregistration_amd import capi as pcr, synth  # noqa: E402


def main():
    rng = np.random.default_rng(7)
    sizes = rng.integers(50000, 80001, 15)
    scans = []
    for i, n in enumerate(sizes):
        s, _ = synth.scan_sequence(i + 1, int(n), 9000)  # scan i of the sequence (deterministic per index)
        scans.append(s[i])
    pairs = [(scans[i + 1], scans[i]) for i in range(len(scans) - 1)]
    conv = pcr.ConvergenceParams.regcfg(30)
    eye, zero = np.eye(3).reshape(9), np.zeros(3)

    def make(ctx, src, tgt):
        six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src)), pcr.Index(ctx, pcr.Cloud(ctx, tgt))
        U = tix.unique_sq_dist()
        return pcr.Fit(ctx, six, tix, 4, U * 36.0, U * 4.0, conv=conv, unique_sq_dist=U, traj_cap=32)

    ctx = pcr.Context(0)
    t0 = time.perf_counter()
    fits = [make(ctx, s, t) for s, t in pairs]
    ctx.synchronize()
    build_s = time.perf_counter() - t0
    res = {}

    def timed(name, fn, reps=5):
        best = None
        for _ in range(reps):
            for f in all_fits[name]:
                f.reset(eye, zero)
            for c in ctxs_of[name]:
                c.synchronize()
            t0 = time.perf_counter()
            fn()
            for c in ctxs_of[name]:
                c.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[name] = best
        return [f.read() for f in all_fits[name]]

    all_fits, ctxs_of = {"sequential": fits, "batch": fits}, {"sequential": [ctx], "batch": [ctx]}

    def run_sequential():
        for f in fits:
            for _ in range(30):
                f.iterate()
    out_seq = timed("sequential", run_sequential)

    batch = pcr.FitBatch(ctx, fits)

    def run_batch():
        for _ in range(30):
            batch.iterate()
    out_bat = timed("batch", run_batch)

    ctxs = [pcr.Context(0) for _ in pairs]
    sfits = [make(c, s, t) for c, (s, t) in zip(ctxs, pairs)]
    all_fits["streams"], ctxs_of["streams"] = sfits, ctxs

    def run_streams():
        for _ in range(30):
            for f in sfits:
                f.iterate()
    out_str = timed("streams", run_streams)

    for a, b, c in zip(out_seq, out_bat, out_str):
        assert a["iterations"] == b["iterations"] == c["iterations"] and a["state"] == b["state"] == c["state"]
        assert (a["R"] == b["R"]).all() and (a["R"] == c["R"]).all() and (a["T"] == b["T"]).all() and (a["T"] == c["T"]).all()
    n = len(pairs)
    print(json.dumps({"workload": f"{n} pairwise fits over 15 synthetic scans of {int(sizes.min())}-{int(sizes.max())} points, "
                                  "K=4, RegCfg criteria, up to 30 iterations each",
                      "iterations_per_fit": [r["iterations"] for r in out_seq],
                      "index_build_s_all_pairs": build_s,
                      "fits_per_s": {k: n / v for k, v in res.items()},
                      "ms_total": {k: v * 1e3 for k, v in res.items()}}))


if __name__ == "__main__":
    main()
