#!/usr/bin/env python3
"""bench.py -- ICP iterations/s (and KNN queries/s) of the MI355X hot path on BASELINE.json's headline
workload: synthetic 1M x 1M-point clouds, known rigid transform + Gaussian noise, K = NN = 4,
SDFT = 36, NDFT = 4, every source point a sample.

One "step" = ONE ICRP iteration: transform the source, forward K-NN, reverse 1-NN, both gates, the
17-sum reduction (+ all-reduce over ranks), 3x3 SVD + composition (a one-thread kernel: the loop is
device-resident, the host only enqueues).  Inputs and indices are resident in HBM before the timed
region; index builds are reported separately.

  python bench.py --gpus 1 --steps 30 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

For N > 1 the samples are sharded over the ranks at FIXED problem size (strong scaling), clouds and
indices replicated, partials combined by an RCCL all-reduce (torch.distributed backend "nccl").
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20250001
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_sample(k):
    """read s_i (24) + K forward winners (24K) + reverse winner (24) + two u32 out (8)."""
    return 56 + 24 * k


def committed_traffic(kernel_prefix, points, k):
    """(HBM bytes, source file, VALU wave instructions) per launch of the step kernel from the PMC passes committed
    under profiles/ (bytes = FETCH_SIZE x 2 [gfx950 correction] + WRITE_SIZE, see tools/summarize_profile.py).  Counters
    cannot be collected from inside this process; the figure is from `rocprofv3 --pmc` runs of this same command
    (tools/profile.sh) and is quoted ONLY for the headline workload it was collected on AND while the profile's
    fingerprint of the device sources equals the current one -- otherwise (None, reason, None): a counter replayed
    after the kernel changed says nothing."""
    if points != 1_000_000 or k != 4:
        return None, "not the profiled workload", None
    import glob
    from pointcloudregistration_amd.stamp import kernel_sources_stamp
    now = kernel_sources_stamp()
    best = (None, "no profile under profiles/ matches the current device sources (" + now + ")", None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if d.get("kernel_sources_stamp") != now:
            continue
        for name, e in d.get("kernels", {}).items():
            if name.startswith(kernel_prefix) and "hbm_traffic_bytes" in e:
                best = (e["hbm_traffic_bytes"], os.path.relpath(f, ROOT), e.get("counters", {}).get("SQ_INSTS_VALU"))
    return best


def live_pmc(kernel_prefix, steps):
    """HBM traffic and VALU instruction count of the step kernel measured NOW, on this box: three `rocprofv3 --pmc` passes
    (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU -- one counter block per pass, as MI355X_MICROARCH.md prescribes) of this same
    command in its short form, run as CHILD processes before this process touches the GPU.  Per-launch means over every
    launch of the kernel in the child (pre-warm, warm-up, timed and instrumented iterations alike); bytes = FETCH_SIZE x 2
    [gfx950: 128-byte requests counted as 64] x 1 KiB + WRITE_SIZE x 1 KiB, as tools/summarize_profile.py computes them for
    the committed profile.  None when rocprofv3 is missing or a pass fails: the committed profile is quoted instead."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None
    got, launches = {}, 0
    base = "/tmp" if os.path.isdir("/tmp") else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            od = os.path.join(d, ctr)
            cmd = ["timeout", "-s", "KILL", "60", exe, "--pmc", ctr, "--output-format", "csv", "-d", od, "--",
                   sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "3", "--prewarm-steps", "200",
                   "--no-cpu-baseline", "--no-knn", "--no-extra", "--no-live-pmc"]
            try:
                r = subprocess.run(cmd, cwd=base, env=dict(os.environ, TMPDIR=base or ""), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.DEVNULL, timeout=90)
            except (OSError, subprocess.SubprocessError):
                return None
            if r.returncode != 0:
                return None
            per = {}
            for f in glob.glob(os.path.join(od, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        name = row["Kernel_Name"].replace("void ", "").replace("pcr::", "")
                        if name.startswith(kernel_prefix) and row["Counter_Name"] == ctr:
                            per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None
            got[ctr] = sum(per.values()) / len(per)
            launches = len(per)
    return {"traffic": got["FETCH_SIZE"] * 2048.0 + got["WRITE_SIZE"] * 1024.0, "read": got["FETCH_SIZE"] * 2048.0,
            "written": got["WRITE_SIZE"] * 1024.0, "valu": got["SQ_INSTS_VALU"], "launches": launches}


def cpu_baseline(src, tgt, R0, T0, k, sdft, ndft, budget_s=25.0):
    """The oracle (CPU restatement of the reference loop) timed on this box's host cores -- the CHECKER
    timed as a baseline, never the thing measured above.  Per iteration it does what the reference does:
    copy + transform the source, rebuild the source KD-tree, serial forward K-NN + reverse 1-NN per
    sample, two-pass centroid/covariance + SVD (the target tree is built once, outside, like the
    reference's __setMethodPointCloud)."""
    from oracle import oracle as O
    U = O.uniq_sq_dist(tgt)
    t0 = time.perf_counter()
    ttree = O.KdOld(tgt, k)
    build_s = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)

    def run(threads, iters):
        R, T = np.array(R0, np.float64), np.array(T0, np.float64)
        t0 = time.perf_counter()
        for _ in range(iters):
            c = O.icrp_correspondences(src, tgt, ttree, R, T, k, U * sdft, U * ndft, threads=threads)
            R, T, _, _ = O.svd2(c["pair_p"], c["pair_q"], R, T)
        return (time.perf_counter() - t0) / iters

    s_multi = run(cores, 2)
    iters_single = 2 if s_multi * cores * 0.6 * 2 < budget_s else 1
    s_single = run(1, iters_single)
    return {"value": 1.0 / s_multi, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"2 full ICRP iterations at {len(src)}x{len(tgt)} points, K={k}, OpenMP over samples on "
                      f"{cores} threads (source KD-tree rebuilt serially every iteration, as the reference does); "
                      f"target-tree build {build_s:.2f} s excluded",
            "single_thread_value": 1.0 / s_single, "single_thread_iterations": iters_single,
            "seconds_per_iteration": s_multi, "single_thread_seconds_per_iteration": s_single}


def extra_c5(capi, synth, device, n=16_000_000, k=4, launches=5):
    """BASELINE config C5 in the driver's line: LiDAR-like n x n pair (density ~ 1/r^2 from the sensor, 30 % of the source
    replaced by uniform outliers), nested-level grid indices, ONE ICRP step over all n samples per launch (production
    mode: gate early-out, no index outputs).  Kernel time per launch by HIP events on the context's stream; inputs and
    indices resident.  Same algorithmic bytes per sample as the headline (56 + 24 K)."""
    t0 = time.perf_counter()
    src, tgt, Rm, t = synth.lidar_pair(n, 516, outlier_frac=0.3)
    gen_s = time.perf_counter() - t0
    ctx = capi.Context(device)
    t0 = time.perf_counter()
    six, tix = capi.Index(ctx, capi.Cloud(ctx, src)), capi.Index(ctx, capi.Cloud(ctx, tgt))
    ctx.synchronize()
    build_s = time.perf_counter() - t0
    U = tix.unique_sq_dist()
    R, T = synth.colmajor(Rm), t
    capi.icp_step(ctx, six, tix, R, T, k, U * 36.0, U * 4.0, exact=False, want_outputs=False)  # warm-up
    ctx.set_profiling(True, classes=[1])
    ctx.read_kernel_times()
    for _ in range(launches):
        out = capi.icp_step(ctx, six, tix, R, T, k, U * 36.0, U * 4.0, exact=False, want_outputs=False)
    ms, cnt = ctx.read_kernel_times()["step"]
    ctx.set_profiling(False)
    step_ms = ms / max(cnt, 1)
    si, ti = six.info(), tix.info()
    by = algorithmic_bytes_per_sample(k) * n
    res = {"workload": f"C5 synthetic LiDAR-like {n}x{n} points, 30% outliers in the source, K={k}, one ICRP step over all "
                       f"samples per launch (seed 516)",
           "step_ms": step_ms, "launches": cnt, "samples_per_s": n / (step_ms * 1e-3), "kernel": capi.last_step_kernel(),
           "kept_pairs": float(out["partials"][0]),
           "roofline": {"bound": "hbm", "achieved": by / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": by / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": by},
           "index_build_s_both_clouds": build_s, "synthesis_s": gen_s,
           "src_index": {kk: si[kk] for kk in ("levels", "cells", "occupied_cells", "device_bytes")},
           "tgt_index": {kk: ti[kk] for kk in ("levels", "cells", "occupied_cells", "device_bytes")}}
    six.close()
    tix.close()
    ctx.close()
    return res


def extra_c4(capi, synth, device, reps=3):
    """BASELINE config C4 in the driver's line: a scan set registered pair by pair (14 fits over 15 synthetic partial
    scans of 50-80 k points; the Happy Buddha scans are not in the reference tree), RegCfg criteria, up to 30 iterations
    each -- fits per second one after the other against ONE launch per iteration for all of them (pcr_fit_batch); both
    must give identical transforms.  Index builds outside the timed regions."""
    rng = np.random.default_rng(7)
    sizes = rng.integers(50000, 80001, 15)
    scans = [synth.scan_sequence(i + 1, int(nn), 9000)[0][i] for i, nn in enumerate(sizes)]
    pairs = [(scans[i + 1], scans[i]) for i in range(len(scans) - 1)]
    conv = capi.ConvergenceParams.regcfg(30)
    eye, zero = np.eye(3).reshape(9), np.zeros(3)
    ctx = capi.Context(device)
    t0 = time.perf_counter()
    fits, keep = [], []
    for sp, tp in pairs:
        six, tix = capi.Index(ctx, capi.Cloud(ctx, sp)), capi.Index(ctx, capi.Cloud(ctx, tp))
        U = tix.unique_sq_dist()
        keep.append((six, tix))
        fits.append(capi.Fit(ctx, six, tix, 4, U * 36.0, U * 4.0, conv=conv, unique_sq_dist=U, traj_cap=32))
    ctx.synchronize()
    build_s = time.perf_counter() - t0
    batch = capi.FitBatch(ctx, fits)

    def timed(fn):
        best = None
        for _ in range(reps):
            for f in fits:
                f.reset(eye, zero)
            ctx.synchronize()
            t0 = time.perf_counter()
            fn()
            ctx.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, [f.read() for f in fits]

    def sequential():
        for f in fits:
            for _ in range(30):
                f.iterate()

    def one_launch():
        for _ in range(30):
            batch.iterate()
    t_seq, r_seq = timed(sequential)
    t_bat, r_bat = timed(one_launch)
    same = all(a["iterations"] == b["iterations"] and a["state"] == b["state"] and (a["R"] == b["R"]).all() and
               (a["T"] == b["T"]).all() for a, b in zip(r_seq, r_bat))
    res = {"workload": f"C4 stand-in: {len(pairs)} pairwise fits over 15 synthetic partial scans of {int(sizes.min())}-"
                       f"{int(sizes.max())} points, K=4, RegCfg criteria, up to 30 iterations each",
           "fits_per_s": {"sequential": len(pairs) / t_seq, "one_launch_per_iteration": len(pairs) / t_bat},
           "ms_total": {"sequential": t_seq * 1e3, "one_launch_per_iteration": t_bat * 1e3},
           "iterations_per_fit": [r["iterations"] for r in r_seq], "batch_equals_sequential_bit_for_bit": bool(same),
           "index_build_s_all_pairs": build_s}
    batch.close()
    for f in fits:
        f.close()
    for a, b in keep:
        a.close()
        b.close()
    ctx.close()
    return res


def spawn_ranks(n):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--points", type=int, default=1_000_000, help="points per cloud (headline: 1M)")
    ap.add_argument("--k", type=int, default=4, help="CorrespondenceEstimationClosestPoints.NN")
    ap.add_argument("--occupancy", type=float, default=0.0, help="grid target occupancy (0 = library default)")
    ap.add_argument("--tune", type=str, default="", help="debug: comma list key=value for pcr_tune")
    ap.add_argument("--prewarm-steps", type=int, default=600,
                    help="untimed iterations run before the W warm-up steps so that the GPU is at its sustained clocks "
                         "(a power-managed device needs ~0.1 s of load: 6 550 it/s after 10 iterations, 6 800 after 300)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C5 (16 M LiDAR step) and C4 (batch of small fits) legs")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not collect the step kernel's counters with rocprofv3 child "
                    "runs before the measurement (the committed profile is quoted instead, while it matches the sources)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: be the launcher.  N fresh child processes, one per GPU, started
    # BEFORE anything here touches the GPU; rank 0's single JSON line is passed through.  (Never a silent one-GPU run.)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # The step kernel's counters, measured on THIS box by three short rocprofv3 child runs of this command -- before this
    # process initialises the GPU (a profiled child must be started from a process that has not).  One GPU, headline
    # workload only; anything going wrong leaves `live` None and the committed profile is quoted.
    live = None
    if (args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_live_pmc and args.points == 1_000_000 and args.k == 4
            and not args.tune and os.environ.get("PCR_BENCH_EMULATE_SHARDS", "0") in ("", "0") and not os.environ.get("PCR_LIB")
            and os.environ.get("PCR_BENCH_FORCE_DIST") != "1"
            # (not when this process is itself being profiled: a profiler inside a profiler)
            and "rocprof" not in os.environ.get("LD_PRELOAD", "").lower()
            and not any(kk.startswith(("ROCPROF", "ROCP_")) for kk in os.environ)):
        try:
            live = live_pmc("icp_step_fast_kernel", args.steps)
        except Exception as e:  # never let the counters cost the measurement
            print(f"bench: live counter passes failed ({type(e).__name__}: {e})", file=sys.stderr)
            live = None

    # Libraries underneath (RCCL prints a version banner) write to stdout; the contract is ONE JSON line there.
    # Everything but that line goes to stderr: fd 1 is pointed at fd 2 until the result is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from pointcloudregistration_amd import capi, synth
    from pointcloudregistration_amd.dist import shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PCR_BENCH_FORCE_DIST=1: run the collective path on a single rank too (rehearsal of the N > 1 code on one GPU)
    force_dist = os.environ.get("PCR_BENCH_FORCE_DIST") == "1"
    # PCR_BENCH_REHEARSE_ONE_GPU=1: every rank on cuda:0 and gloo instead of RCCL (which refuses two ranks on one
    # device) -- a rehearsal of the N > 1 code path on a one-GPU box; its numbers mean nothing
    rehearse = os.environ.get("PCR_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("PCR_BENCH_DUMP_AFTER", "240")), exit=True)  # a rehearsal that stalls says where
        if os.environ.get("PCR_BENCH_THREAD_DUMP_AFTER"):  # ... and what every OS thread was blocked in (tools/thread_dump.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import thread_dump
            thread_dump.arm(float(os.environ["PCR_BENCH_THREAD_DUMP_AFTER"]))
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node = --gpus"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n, k, sdft, ndft = args.points, args.k, 36.0, 4.0
    src, tgt, R_true, t_true = synth.registration_pair(n, n, SEED)

    if os.environ.get("PCR_LIB"):
        capi.LIB_PATH = os.environ["PCR_LIB"]
    for kv in filter(None, args.tune.split(",")):
        kk, vv = kv.split("=")
        capi.lib().pcr_tune(int(kk), int(vv))
    # a stream of this process's own, current for torch too: the legacy default stream orders itself against every other
    # blocking stream of the process on each submission
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = capi.Context(local_rank, stream=stream)
    src_t = torch.from_numpy(src).to(dev)
    tgt_t = torch.from_numpy(tgt).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scloud, tcloud = capi.Cloud(ctx, src_t), capi.Cloud(ctx, tgt_t)
    six = capi.Index(ctx, scloud, kind=capi.INDEX_GRID, target_occupancy=args.occupancy)
    tix = capi.Index(ctx, tcloud, kind=capi.INDEX_GRID, target_occupancy=args.occupancy)
    ctx.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    U = tix.unique_sq_dist()
    thr_f, thr_r = U * sdft, U * ndft

    begin, m = shard_range(n, rank, world)
    emulate = int(os.environ.get("PCR_BENCH_EMULATE_SHARDS", "0"))  # debug: one rank does the share of 1/N (no peers)
    if emulate > 1:
        begin, m = shard_range(n, 0, emulate)
    R0, T0 = synth.coarse_initial_transform(R_true, t_true, n)

    # The device-resident loop (pcr_fit_*): per iteration the host only ENQUEUES transform + fused step + fold,
    # the RCCL all-reduce of the 17 partials (N > 1), and the one-thread solve kernel; (R,T) stays on the device,
    # so nothing waits for the host between iterations.  Convergence tests are off for the timed loop (SURVEY 8d:
    # a fixed number of iterations); the result is read back once, after the timed region's barrier.
    fit = capi.Fit(ctx, six, tix, k, thr_f, thr_r, conv=None, unique_sq_dist=U, sample_begin=begin, m=m,
                   traj_cap=max(args.steps, args.warmup, 1))
    part_t = fit.partials_tensor()

    # The exchange: the library's own in-stream RCCL all-reduce (pcr_allreduce_partials, on the stream the kernels run
    # on: no side stream, no event hops) -- or, if that cannot be set up on every rank, torch.distributed's communicator.
    # PCR_BENCH_COMM=torch forces the latter.
    comm, exchange = None, "none"
    if world > 1 or force_dist:
        exchange = "torch.distributed all_reduce (gloo)" if rehearse else "torch.distributed all_reduce (RCCL)"
        if not rehearse and os.environ.get("PCR_BENCH_COMM", "lib") == "lib":
            uid = [None]
            if rank == 0:
                try:
                    uid = [capi.Comm.unique_id()]
                except Exception as e:  # RCCL not loadable: every rank learns it from the broadcast below
                    print(f"bench: pcr_comm_unique_id failed ({e}); using torch.distributed", file=sys.stderr)
            dist.broadcast_object_list(uid, src=0)
            ok = 0
            if uid[0] is not None:
                try:
                    comm = capi.Comm(ctx, rank, world, uid[0])
                    ok = 1
                except Exception as e:
                    print(f"bench: rank {rank}: pcr_comm_create failed ({e})", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                exchange = "pcr_allreduce_partials (RCCL, in-stream)"
            else:
                comm = None
    part_ptr = part_t.data_ptr()

    def enqueue(steps):
        for _ in range(steps):
            if not (world > 1 or force_dist):
                fit.iterate()  # one launch: the step kernel's last block folds and solves
                continue
            fit.step()
            if world > 1 or force_dist:
                if rehearse:
                    torch.cuda.synchronize()  # gloo stages through the host: tell a stalled kernel from a stalled exchange
                if comm is not None:
                    comm.allreduce_partials(part_ptr)
                elif rehearse and os.environ.get("PCR_BENCH_GLOO_DEVICE") == "1":  # (diagnostic: the route that stalled)
                    dist.all_reduce(part_t)
                elif rehearse:
                    # gloo on a HOST tensor.  Its device-tensor path stalls now and then when two processes share one GPU:
                    # ProcessGroupGloo's worker thread spins in hipStreamSynchronize on its pool stream behind the staging
                    # copy while this library's stream is idle -- reproduced with nothing of this repository loaded
                    # (tools/repro_gloo_device_stall.py; stacks in profiles/r3_gloo_device_stall.txt)
                    host = part_t.cpu()
                    dist.all_reduce(host)
                    part_t.copy_(host)
                else:
                    dist.all_reduce(part_t)
            fit.solve()

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # device pre-warm (not one of the W warm-up steps, not timed): the same loop, long enough for the clocks to settle
    if args.prewarm_steps > 0 and (not rehearse or os.environ.get("PCR_BENCH_PREWARM_IN_REHEARSAL") == "1"):  # (a rehearsal's numbers mean nothing, and gloo makes every step slow)
        fit.reset(R0, T0)
        enqueue(args.prewarm_steps)
        barrier()
    # warm-up: every kernel class bracketed by HIP events (the figures of the minor kernels come from here)
    fit.reset(R0, T0)
    ctx.set_profiling(True)
    ctx.read_kernel_times()  # reset
    enqueue(args.warmup)
    barrier()
    kt_warm = ctx.read_kernel_times()
    # The timed fit restarts from the coarse initial transform: steps 1..K of one registration.  Nothing is bracketed per
    # launch in here -- a HIP event pair per launch serialises the dispatches and costs 7-9 us of a ~146 us iteration
    # (tools/bench_graph.py: 145.7 vs 153.0 us) -- ONE event pair on the kernels' stream brackets the whole region.
    fit.reset(R0, T0)
    ctx.set_profiling(False)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    enqueue(args.steps)
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)
    res = fit.read()
    assert res["iterations"] == args.steps, res
    R, T, mse, kept = res["R"], res["T"], res["mse"], res["pairs"]
    # the same K steps once more, untimed, every launch of the dominant kernel bracketed by its own HIP event pair: the
    # per-launch durations rocprofv3's kernel trace is compared with (both serialise the dispatches)
    fit.reset(R0, T0)
    ctx.set_profiling(True, classes=(1,))
    ctx.read_kernel_times()
    barrier()
    enqueue(args.steps)
    barrier()
    per_launch = ctx.read_kernel_launches(1)
    kt = ctx.read_kernel_times()
    ctx.set_profiling(False)
    again = fit.read()
    assert (again["R"] == R).all() and again["pairs"] == kept  # the instrumented pass did the same work
    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)  # (gloo: host tensors, see above)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    elapsed = max_over_ranks(elapsed)

    # registration quality of the timed fit (not part of the metric; shows the work was real)
    R_est = synth.rowcol(R)
    rot_err = float(np.abs(R_est - R_true).max())
    trans_err = float(np.abs(T - t_true).max())

    knn = None
    if not args.no_knn:
        # the query batch is sharded over the ranks like the samples are (contiguous shares, target index replicated, no
        # exchange); the figure is the whole batch over the slowest rank's time
        qb, qm = shard_range(n, rank, world)
        q_t = torch.from_numpy(np.ascontiguousarray((src @ R_true.T + t_true)[qb:qb + qm])).to(dev)
        knn = {}
        for kk in (1, 4):  # 3 warm-ups, median of 20, HIP events on the stream the kernels run on
            oi = torch.empty((qm, kk), dtype=torch.int32, device=dev)
            ts = []
            for rep in range(23):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                tix.knn_d(q_t, kk, oi)
                b.record(stream)
                b.synchronize()
                ts.append(a.elapsed_time(b) * 1e-3)
            med = float(np.median(ts[3:]))
            med = max_over_ranks(med)
            knn[f"k{kk}"] = n / med
        # 24 + 28 K algorithmic bytes per query; peak = every rank's HBM
        knn["roofline"] = {f"k{kk}": {"bound": "hbm", "achieved": knn[f"k{kk}"] * (24 + 28 * kk) / 1e9, "peak": HBM_PEAK_GBS * world,
                                      "unit": "GB/s", "frac": knn[f"k{kk}"] * (24 + 28 * kk) / 1e9 / (HBM_PEAK_GBS * world),
                                      "algorithmic_bytes_per_query": 24 + 28 * kk} for kk in (1, 4)}
        knn["includes"] = "grouping the query batch by grid brick (two small kernels) + the search kernel; device-resident in/out"
        knn["queries"] = n
        knn["sharded_over_ranks"] = world
    reject = None
    if rank == 0 and not args.no_knn:
        # CRejectionCentroidDist on the kept pairs of the last iteration's transform: two K = MSN = 25 searches per pair
        st = capi.icp_step(ctx, six, tix, R, T, k, thr_f, thr_r, exact=True)
        kp = st["keep"].astype(bool)
        pp = (src @ synth.rowcol(R).T + T)[kp]
        pq = tgt[st["fwd"][kp]]
        capi.reject_centroid_dist(ctx, six, tix, pp, pq, msn=25, msd=8.0, cdf=1.0)  # warm-up (as the K-NN figures have theirs)
        ctx.set_profiling(True, classes=(3,))
        ctx.read_kernel_times()
        rj = capi.reject_centroid_dist(ctx, six, tix, pp, pq, msn=25, msd=8.0, cdf=1.0)
        ms = ctx.read_kernel_times()["knn"][0]
        ctx.set_profiling(False)
        by = (24 + 28 * 25) * 2 * len(pp)
        reject = {"pairs": int(len(pp)), "kept": rj["kept"], "msn": 25, "stat_kernel_ms": ms, "warmup_calls": 1,
                  "roofline": {"bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0,
                               "algorithmic_bytes_per_pair": (24 + 28 * 25) * 2}}
    if rank == 0:
        # The dominant kernel's average launch duration.  One GPU: an iteration IS one launch of it (search, sums, fold,
        # solve), so the event pair around the timed region, divided by K, bounds it from above (it includes the gaps
        # between launches) -- measured on the un-instrumented run the headline value comes from.  N > 1: other work sits
        # between the launches, so the per-launch event pairs of the instrumented pass are what there is.
        per_launch_ms = kt["step"][0] / max(1, kt["step"][1])
        single = not (world > 1 or force_dist)
        step_ms = region_ms / args.steps if single else per_launch_ms
        algo_bytes = algorithmic_bytes_per_sample(k) * m
        achieved = algo_bytes / (step_ms * 1e-3) / 1e9 if step_ms > 0 else 0.0
        info = tix.info()
        if live is not None and capi.last_step_kernel().startswith("icp_step_fast_kernel"):
            traffic, valu_insts = live["traffic"], live["valu"]
            traffic_src = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, one pass each, of "
                           f"`bench.py --steps {args.steps} --warmup 3 --prewarm-steps 200 --no-knn --no-extra` as child processes "
                           f"before the timed run; per-launch means over {live['launches']} launches; read {live['read']:.4g} B "
                           f"(FETCH_SIZE x 2 KiB, gfx950 correction) + written {live['written']:.4g} B")
        else:
            traffic, traffic_src, valu_insts = (committed_traffic(capi.last_step_kernel(), n, k) if world == 1
                                                else (None, "multi-GPU run", None))
        # VALU utilisation (north star): wave instructions x 4 issue cycles over the SIMD-cycles of the launch
        # (256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock); instruction count from the committed PMC pass
        valu_util = (valu_insts * 4.0 / (1024 * step_ms * 1e-3 * 2.4e9)) if valu_insts and step_ms > 0 else None
        out = {
            "metric": "icp_iterations_per_s", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C3 synthetic {n}x{n}-point three-sphere surface, known rigid transform + "
                                   f"Gaussian noise (seed {SEED}), K={k}, SDFT={sdft:g}, NDFT={ndft:g}, every "
                                   f"source point a sample, {args.steps} ICRP iterations from a coarse alignment "
                                   f"(known transform perturbed by ~1.5 point spacings)",
                       "points_per_cloud": n, "k": k, "parallelism": f"sample-sharded x{world}, clouds+indices "
                       "replicated, 17-double all-reduce per iteration", "exchange": exchange},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "valu_insts_per_launch": valu_insts, "valu_utilisation": valu_util,
                         "kernel": capi.last_step_kernel(),
                         "kernel_ms_avg": step_ms, "launches": args.steps,
                         "kernel_ms_avg_source": ("one HIP event pair on the kernels' stream around the K back-to-back launches of "
                                                  "the timed region, / K: an upper bound (gaps between launches included)"
                                                  if single else "a HIP event pair around every launch (instrumented pass)"),
                         "kernel_ms_avg_per_launch_events": per_launch_ms,
                         "per_launch_events_note": "the same K steps run once more, untimed, each launch bracketed by its own "
                                                   "event pair: dispatches are then serialised, as under rocprofv3 --kernel-trace "
                                                   "(profiles/), which is why this figure -- not kernel_ms_avg -- matches the "
                                                   "profile's average; tools/bench_graph.py measures the difference",
                         "kernel_ms_first_iteration": float(per_launch[0]) if len(per_launch) else None,
                         "kernel_ms_per_iteration": [round(float(x), 5) for x in per_launch],
                         "prewarm_steps": args.prewarm_steps,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "other_kernels_ms_avg": {"transform": kt_warm["transform"][0] / max(1, kt_warm["transform"][1]),
                                                  "fold": kt_warm["fold"][0] / max(1, kt_warm["fold"][1]),
                                                  "measured_over": "warm-up iterations"}},
            "knn_queries_per_s": knn,
            "reject_centroid_dist": reject,
            "index": {"build_ms_both_clouds": build_ms, "cell_size": info["cell_size"], "dims": info["dims"],
                      "mean_occupancy": info["mean_occupancy"], "cells": info["cells"], "levels": info["levels"]},
            "fit": {"kept_pairs_last": float(kept), "mse_last": float(mse), "max_abs_rot_err": rot_err,
                    "max_abs_trans_err": trans_err},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(src, tgt, R0, T0, k, sdft, ndft)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if world == 1 and not args.no_extra and not force_dist and emulate <= 1:
            # the other BASELINE configurations that fit one GPU, after the headline (extra keys; `value` is C3's)
            for key, fn in (("c5", extra_c5), ("c4", extra_c4)):
                try:
                    out[key] = fn(capi, synth, local_rank)
                except Exception as e:  # an extra leg must never cost the headline line
                    out[key] = {"error": f"{type(e).__name__}: {e}"}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
