#!/usr/bin/env python3
"""Config C5 as a command the profiler can wrap: LiDAR-like 16 M x 16 M pair, 30 % outliers, nested
index levels; runs `--steps` production-mode ICRP iterations' worth of step launches over all samples and prints one
JSON line (kernel time per launch by HIP events on the context's stream).
    python3 tools/c5_step.py [--n 16000000] [--steps 5] [--outliers 0.3]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--outliers", type=float, default=0.3)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--occupancy", type=float, default=0.0, help="index target occupancy (0: library default)")
    ap.add_argument("--max-cells", type=int, default=0, help="cap on a level's cell table (0: library default)")
    ap.add_argument("--tables", type=int, default=0, help="pcr_index_params.compact_tables: 0 = dense cell tables (default), "
                    "1 = compact tables for the levels under 1/8 occupancy, 2 = for every level")
    ap.add_argument("--fit", type=int, default=0, help="instead: a device-resident fit of this many iterations from a "
                    "perturbed start; prints every iteration's kernel time (what the temporal bounds buy shows from the second)")
    a = ap.parse_args()
    from pointcloudregistration_amd import capi as pcr, synth
    src, tgt, Rm, t = synth.lidar_pair(a.n, 516, outlier_frac=a.outliers)
    ctx = pcr.Context(0)
    import time
    t_build = time.perf_counter()
    six = pcr.Index(ctx, pcr.Cloud(ctx, src), target_occupancy=a.occupancy, max_cells=a.max_cells, compact_tables=a.tables)
    tix = pcr.Index(ctx, pcr.Cloud(ctx, tgt), target_occupancy=a.occupancy, max_cells=a.max_cells, compact_tables=a.tables)
    ctx.synchronize()
    t_build = time.perf_counter() - t_build
    U = tix.unique_sq_dist()
    R, T = synth.colmajor(Rm), t
    if a.fit > 0:
        R0, T0 = synth.coarse_initial_transform(Rm, t, a.n)
        fit = pcr.Fit(ctx, six, tix, a.k, U * 36.0, U * 4.0, conv=None, unique_sq_dist=U, traj_cap=a.fit)
        fit.reset(R0, T0)
        ctx.set_profiling(True, classes=[1])
        per = []
        for _ in range(a.fit):
            fit.iterate()
            ctx.synchronize()
            per.append(round(ctx.read_kernel_times()["step"][0], 3))
        res = fit.read()
        print(json.dumps({"workload": f"C5 lidar {a.n}x{a.n}, {a.outliers:.0%} outliers, K={a.k}, device-resident fit",
                          "iteration_ms": per, "kept_pairs_last": float(res["pairs"]), "mse_last": float(res["mse"]),
                          "kernel": pcr.last_step_kernel()}))
        return
    pcr.icp_step(ctx, six, tix, R, T, a.k, U * 36.0, U * 4.0, exact=False, want_outputs=False)   # warm-up
    ctx.set_profiling(True, classes=[1])
    for _ in range(a.steps):
        out = pcr.icp_step(ctx, six, tix, R, T, a.k, U * 36.0, U * 4.0, exact=False, want_outputs=False)
    ms, cnt = ctx.read_kernel_times()["step"]
    si, ti = six.info(), tix.info()
    print(json.dumps({"workload": f"C5 lidar {a.n}x{a.n}, {a.outliers:.0%} outliers, K={a.k}", "step_ms": ms / max(cnt, 1),
                      "launches": cnt, "samples_per_s": a.n / (ms / max(cnt, 1) * 1e-3), "kept_pairs": float(out["partials"][0]),
                      "kernel": pcr.last_step_kernel(), "tables": a.tables, "build_s_both": round(t_build, 3),
                      "src_index": {k: si[k] for k in ("levels", "cells", "occupied_cells", "device_bytes")},
                      "tgt_index": {k: ti[k] for k in ("levels", "cells", "occupied_cells", "device_bytes")}}))


if __name__ == "__main__":
    main()
