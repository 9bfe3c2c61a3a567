#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box)
into the small files committed under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats, verbatim
  profiles/<tag>_bench_trace.json   the bench.py line printed under that trace
  profiles/<tag>_pmc.json           per-kernel per-launch counter means, plus
                                    the corrected HBM traffic of the step kernel

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]":
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B read
requests as 64 B, so reads are doubled; WRITE_SIZE is taken as is.  The x2 was
re-checked on this library's own streaming kernel (transform_kernel: 32 MB read
per launch, FETCH_SIZE reports 15.8 MB).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    return n.replace("void ", "").replace("pcr::", "").strip()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", "prof_" + tag)
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)

    stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        stats.sort(key=os.path.getmtime)
        shutil.copy(stats[-1], os.path.join(dst, tag + "_kernel_stats.csv"))
    bt = os.path.join(src, "bench_trace.json")
    if os.path.exists(bt):
        shutil.copy(bt, os.path.join(dst, tag + "_bench_trace.json"))

    per = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> [per-dispatch sums]
    meta = {}
    newest = {}                                      # one (the latest) file per counter pass
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        d = os.path.relpath(f, src).split(os.sep)[0]
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in sorted(newest.values()):
        disp = defaultdict(float)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                disp[(k, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
                meta[k] = {"grid": int(row["Grid_Size"]), "workgroup": int(row["Workgroup_Size"]),
                           "lds_bytes": int(row["LDS_Block_Size"]), "scratch_bytes": int(row["Scratch_Size"]),
                           "vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]),
                           "sgpr": int(row["SGPR_Count"])}
        for (k, _d, c), v in disp.items():
            per[k][c].append(v)

    # the fingerprint profile.sh took on the GPU box when the counters were collected -- never the sources' state at the time
    # this summary is written (a re-summarised old collection must not pass for a fresh one)
    stamp_file = os.path.join(src, "kernel_sources_stamp.txt")
    stamp = open(stamp_file).read().strip() if os.path.exists(stamp_file) else None
    out = {"tag": tag, "note": "per-launch means over the dispatches of one `bench.py --steps 30 --warmup 3` run per counter pass",
           "kernel_sources_stamp": stamp,  # bench.py quotes these counters only while this matches
           "kernels": {}}
    for k, cs in sorted(per.items()):
        e = dict(meta[k])
        e["launches"] = max(len(v) for v in cs.values())
        e["counters"] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
        cc = e["counters"]
        if "FETCH_SIZE" in cc and "WRITE_SIZE" in cc:
            e["hbm_read_bytes"] = cc["FETCH_SIZE"] * 1024.0 * 2.0      # gfx950 correction
            e["hbm_write_bytes"] = cc["WRITE_SIZE"] * 1024.0
            e["hbm_traffic_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        out["kernels"][k] = e
    with open(os.path.join(dst, tag + "_pmc.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    step = [k for k in out["kernels"] if k.startswith("icp_step_")]
    for k in step:
        print(k, json.dumps(out["kernels"][k], indent=1))


if __name__ == "__main__":
    main()
