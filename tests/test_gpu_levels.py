"""GPU parity of the nested-level index (clouds of very uneven density: the build hands the dense region down to a
finer grid of its own).  The same queries must give the same answers -- bit for bit -- on the nested index, on a
single uniform grid over the same cloud, and from the oracle's KD-tree."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(pcr, ctx):
    from pointcloudregistration_amd import synth
    src, tgt, Rm, t = synth.lidar_pair(300_000, 77)
    cs, ct = pcr.Cloud(ctx, src), pcr.Cloud(ctx, tgt)
    nested = (pcr.Index(ctx, cs, kind=pcr.INDEX_GRID), pcr.Index(ctx, ct, kind=pcr.INDEX_GRID))
    flat = (pcr.Index(ctx, cs, kind=pcr.INDEX_GRID, max_levels=1), pcr.Index(ctx, ct, kind=pcr.INDEX_GRID, max_levels=1))
    return src, tgt, Rm, t, nested, flat


def test_build_splits_a_lidar_sweep_and_leaves_uniform_clouds_alone(pcr, ctx, scene):
    from pointcloudregistration_amd import synth
    src, tgt, _, _, nested, flat = scene
    for ix in nested:
        info = ix.info()
        assert 2 <= info["levels"] <= 6 and info["n"] == 300_000
    for a, b in zip(nested, flat):
        assert b.info()["levels"] == 1
        assert a.info()["max_occupancy"] < b.info()["max_occupancy"] / 4   # what the levels are for
    u, _, _, _ = synth.registration_pair(200_000, 1000, 3)
    assert pcr.Index(ctx, pcr.Cloud(ctx, u), kind=pcr.INDEX_GRID).info()["levels"] == 1


@pytest.mark.parametrize("k", [1, 4, 9, 32])
def test_knn_nested_equals_flat_equals_oracle(pcr, ctx, oracle, scene, k):
    src, tgt, _, _, nested, flat = scene
    rng = np.random.default_rng(k)
    q = np.concatenate([src[rng.choice(len(src), 60000, replace=False)],       # near the data, and 30 % outliers
                        tgt[:2000],                                            # the indexed points themselves
                        (rng.random((2000, 3)) - 0.5) * 2000.0,                # far outside every level
                        rng.standard_normal((2000, 3)) * [3.0, 3.0, 0.05]])    # the dense core
    ia, da = nested[1].knn(q, k, want_d2=True)
    ib, db = flat[1].knn(q, k, want_d2=True)
    assert (ia == ib).all() and (da == db).all()
    pick = rng.choice(len(q), 8000, replace=False)
    wi, wd = oracle.KdOld(tgt, k).query(q[pick], want_d2=True, threads=16)
    assert (wi == ia[pick]).all() and (wd == da[pick]).all()
    # a radius cap prunes whole levels; the answers must still be the capped prefix of the uncapped ones
    cap = float(np.median(da[:, -1]))
    ic, dc = nested[1].knn(q, k, max_d2=cap, want_d2=True)
    inside = da < cap
    assert (ic[inside] == ia[inside]).all() and (ic[~inside] == pcr.NO_INDEX).all()


def test_icp_step_nested_equals_flat_equals_oracle(pcr, ctx, oracle, scene):
    from pointcloudregistration_amd import synth
    src, tgt, Rm, t, nested, flat = scene
    U = nested[1].unique_sq_dist()
    assert U == flat[1].unique_sq_dist() == oracle.uniq_sq_dist(tgt)
    R, T = synth.colmajor(synth.euler_zyx(0.3, -0.2, 1.2)), np.array([0.3, -0.15, 0.04])
    a = pcr.icp_step(ctx, nested[0], nested[1], R, T, 4, U * 36.0, U * 4.0, exact=True)
    b = pcr.icp_step(ctx, flat[0], flat[1], R, T, 4, U * 36.0, U * 4.0, exact=True)
    c = pcr.icp_step(ctx, nested[0], flat[1], R, T, 4, U * 36.0, U * 4.0, exact=True)   # mixed pair of indices
    want = oracle.icrp_correspondences(src, tgt, oracle.KdOld(tgt, 4), R, T, 4, U * 36.0, U * 4.0, threads=16)
    for got in (a, b, c):
        assert (got["fwd"] == want["fwd"]).all() and (got["rev"] == want["rev"]).all() and (got["keep"] == want["keep"]).all()
        assert got["partials"][0] == want["P"] > 1000
    np.testing.assert_allclose(a["partials"], b["partials"], rtol=1e-11, atol=1e-9)
    # production mode (gated searches) agrees with the exact mode on everything that is kept
    p = pcr.icp_step(ctx, nested[0], nested[1], R, T, 4, U * 36.0, U * 4.0, exact=False)
    assert (p["keep"] == want["keep"]).all()
    assert (p["fwd"][want["keep"] == 1] == want["fwd"][want["keep"] == 1]).all()


def test_fit_on_nested_indices_matches_the_oracle_loop(pcr, ctx, oracle, scene):
    from pointcloudregistration_amd import synth
    src, tgt, Rm, t, nested, flat = scene
    U = nested[1].unique_sq_dist()
    R0, T0 = synth.colmajor(synth.euler_zyx(0.3, -0.2, 1.2)), np.array([0.3, -0.15, 0.04])
    conv = pcr.ConvergenceParams.regcfg(6)
    out = []
    for six, tix in (nested, flat):
        fit = pcr.Fit(ctx, six, tix, 4, U * 36.0, U * 4.0, conv, U)
        out.append(fit.run(R0, T0, 6))
        fit.close()
        ctx.synchronize()
    assert out[0]["iterations"] >= 2
    assert out[0]["iterations"] == out[1]["iterations"] and out[0]["state"] == out[1]["state"]
    np.testing.assert_allclose(out[0]["R"], out[1]["R"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out[0]["T"], out[1]["T"], rtol=0, atol=1e-10)


def test_grid_sampler_counts_every_level(pcr, ctx, scene):
    src, tgt, _, _, nested, flat = scene
    smp = pcr.sample_grid(ctx, nested[0])
    assert len(smp) == nested[0].info()["occupied_cells"]
    assert (np.diff(smp.astype(np.int64)) > 0).all() and smp.max() < len(src)


def test_compact_cell_tables_answer_like_dense_ones(pcr, ctx, scene):
    """A sparse level of a nested index keeps occupancy bits, ranks and the starts of its occupied cells instead of a
    start per cell (pcr_index_params.compact_tables, GridView::cbits).  Every level compact (2) and the sparse levels
    compact (1) against every level dense (0, the default): K-NN indices and distances, the ICP step's indices, flags and
    sums, the rejection statistic and the grid sampler's picks must not change by a bit; the index must shrink."""
    from pointcloudregistration_amd import synth
    src, tgt, Rm, t, dense, _ = scene
    cs, ct = dense[0].cloud, dense[1].cloud
    compact = (pcr.Index(ctx, cs, kind=pcr.INDEX_GRID, compact_tables=2), pcr.Index(ctx, ct, kind=pcr.INDEX_GRID, compact_tables=2))
    sparse = (pcr.Index(ctx, cs, kind=pcr.INDEX_GRID, compact_tables=1), pcr.Index(ctx, ct, kind=pcr.INDEX_GRID, compact_tables=1))
    for a, b, c in zip(dense, compact, sparse):
        assert a.info()["levels"] == b.info()["levels"] == c.info()["levels"] >= 2
        assert b.info()["device_bytes"] < a.info()["device_bytes"]
        assert c.info()["device_bytes"] < a.info()["device_bytes"]      # this cloud does have levels under 1/8 occupancy
    rng = np.random.default_rng(5)
    q = np.concatenate([src[rng.choice(len(src), 60000, replace=False)], tgt[:2000],
                        (rng.random((2000, 3)) - 0.5) * 2000.0, rng.standard_normal((2000, 3)) * [3.0, 3.0, 0.05]])
    for k in (1, 4, 9):
        ia, da = dense[1].knn(q, k, want_d2=True)
        ib, db = compact[1].knn(q, k, want_d2=True)
        assert (ia == ib).all() and (da == db).all(), k
    U = dense[1].unique_sq_dist()
    assert U == compact[1].unique_sq_dist()
    for R, T in ((synth.colmajor(synth.euler_zyx(0.3, -0.2, 1.2)), np.array([0.3, -0.15, 0.04])), (synth.colmajor(Rm), t)):
        for k, exact in ((4, True), (4, False), (1, False), (6, True)):
            a = pcr.icp_step(ctx, dense[0], dense[1], R, T, k, U * 36.0, U * 4.0, exact=exact)
            assert pcr.last_step_kernel() == "icp_step_grid_kernel<multi-level>"
            b = pcr.icp_step(ctx, compact[0], compact[1], R, T, k, U * 36.0, U * 4.0, exact=exact)
            assert pcr.last_step_kernel() == "icp_step_grid_kernel<multi-level, compact tables>"
            c = pcr.icp_step(ctx, dense[0], sparse[1], R, T, k, U * 36.0, U * 4.0, exact=exact)   # mixed pair of indices
            for key in ("fwd", "rev", "keep", "partials"):
                assert (a[key] == b[key]).all() and (a[key] == c[key]).all(), (k, exact, key)
    smp_a, smp_b = pcr.sample_grid(ctx, dense[0]), pcr.sample_grid(ctx, compact[0])
    assert (smp_a == smp_b).all() and len(smp_a) == dense[0].info()["occupied_cells"]
    step = pcr.icp_step(ctx, dense[0], dense[1], synth.colmajor(Rm), t, 4, U * 36.0, U * 4.0, exact=False)
    keep = np.flatnonzero(step["keep"])[:20000]
    assert len(keep) > 5000
    pp, pq = src[keep] @ Rm.T + t, tgt[step["fwd"][keep]]
    ra = pcr.reject_centroid_dist(ctx, dense[0], dense[1], pp, pq, msn=25, msd=8.0, cdf=1.0)
    rb = pcr.reject_centroid_dist(ctx, compact[0], compact[1], pp, pq, msn=25, msd=8.0, cdf=1.0)
    assert (ra["d"] == rb["d"]).all() and (ra["keep"] == rb["keep"]).all() and ra["mean"] == rb["mean"]
    for ix in compact + sparse:
        ix.close()


def _skewed_clouds():
    """Clouds built to stress the level split: (name, points)."""
    rng = np.random.default_rng(11)
    out = []
    # a dense blob inside a sparse box, 60 % of the points exact duplicates of a handful of positions
    bg = (rng.random((40000, 3)) - 0.5) * 100.0
    blob = rng.standard_normal((60000, 3)) * 0.05 + [10.0, -5.0, 2.0]
    dup = np.repeat(rng.standard_normal((6, 3)) * 0.02 + [10.0, -5.0, 2.0], 25000, axis=0)
    out.append(("duplicates", np.concatenate([bg, blob, dup])))
    # two far-apart dense clusters: their common box spans most of the cloud, so the build may not split at all
    a = rng.standard_normal((50000, 3)) * 0.02 + [-40.0, -40.0, 0.0]
    b = rng.standard_normal((50000, 3)) * 0.02 + [40.0, 40.0, 5.0]
    out.append(("two_clusters", np.concatenate([bg, a, b])))
    # everything in one plane (z == 0 exactly), density falling with the radius; and on one line
    r = np.exp(rng.random(150000) * np.log(500.0)) * 0.1
    th = rng.random(150000) * 2 * np.pi
    out.append(("planar", np.stack([r * np.cos(th), r * np.sin(th), np.zeros_like(r)], 1)))
    out.append(("collinear", np.stack([r * np.sign(np.cos(th)), np.zeros_like(r), np.zeros_like(r)], 1)))
    # nested blobs: three scales inside each other (forces several levels)
    out.append(("nested_blobs", np.concatenate([bg] + [rng.standard_normal((50000, 3)) * s + [1.0, 2.0, 3.0] for s in (5.0, 0.3, 0.01)])))
    return out


@pytest.mark.parametrize("case", range(5))
def test_skewed_builds_answer_like_the_oracle(pcr, ctx, oracle, case):
    name, pts = _skewed_clouds()[case]
    pts = np.ascontiguousarray(pts)
    cloud = pcr.Cloud(ctx, pts)
    nested = pcr.Index(ctx, cloud, kind=pcr.INDEX_GRID)
    two = pcr.Index(ctx, cloud, kind=pcr.INDEX_GRID, max_levels=2)
    assert 1 <= two.info()["levels"] <= 2 and nested.info()["levels"] <= 6
    rng = np.random.default_rng(case)
    q = np.concatenate([pts[rng.choice(len(pts), 6000, replace=False)] + rng.standard_normal((6000, 3)) * 1e-3,
                        pts[:500],
                        (rng.random((500, 3)) - 0.5) * 400.0])
    for k in (1, 5):
        ia, da = nested.knn(q, k, want_d2=True)
        ib, db = two.knn(q, k, want_d2=True)
        assert (da == db).all(), name
        wi, wd = oracle.KdOld(pts, k + 1).query(q, want_d2=True, threads=16)
        assert (wd[:, :k] == da).all(), name
        assert (ia == ib).all(), name
        if name != "duplicates":
            # an exact fp64 tie (it happens: far queries against a line of points) is decided by index here, by
            # visiting order in the KD-tree; everywhere else the indices must agree
            untied = (np.diff(wd, axis=1) > 0).all(1)
            assert untied.mean() > 0.99 and (ia[untied] == wi[untied, :k]).all(), name
        else:
            # thousands of exact duplicates: check the lowest-index rule itself on a few queries by brute force,
            # in the reference's operation order
            for j in rng.choice(len(q), 60, replace=False):
                d = ((q[j, 0] - pts[:, 0]) ** 2 + (q[j, 1] - pts[:, 1]) ** 2) + (q[j, 2] - pts[:, 2]) ** 2
                best = np.lexsort((np.arange(len(pts)), d))[:k]
                assert (best == ia[j]).all() and (d[best] == da[j]).all(), (name, j)
    smp = pcr.sample_grid(ctx, nested)
    assert len(smp) == nested.info()["occupied_cells"] and len(np.unique(smp)) == len(smp)


def test_centroid_rejection_nested_equals_flat_equals_oracle(pcr, ctx, oracle, scene):
    """CRejectionCentroidDist runs two K=25 searches per pair: the nested index must not change a bit of it."""
    from pointcloudregistration_amd import synth
    src, tgt, Rm, t, nested, flat = scene
    U = nested[1].unique_sq_dist()
    R, T = synth.colmajor(Rm), t
    pairs = oracle.icrp_correspondences(src, tgt, oracle.KdOld(tgt, 4), R, T, 4, U * 36.0, U * 4.0, threads=16)
    pp, pq = pairs["pair_p"][:40000], pairs["pair_q"][:40000]
    for msd in (8.0, 0.05):
        a = pcr.reject_centroid_dist(ctx, nested[0], nested[1], pp, pq, msn=25, msd=msd, cdf=1.0)
        b = pcr.reject_centroid_dist(ctx, flat[0], flat[1], pp, pq, msn=25, msd=msd, cdf=1.0)
        # (the flat index takes the one-wave-per-search kernel, the nested one the lane-per-search kernel: both add the
        # neighbours nearest first, as the reference does, so the statistics agree bit for bit)
        assert (a["d"] == b["d"]).all() and (a["keep"] == b["keep"]).all() and a["mean"] == b["mean"]
        want = oracle.reject_centroid_dist(src, tgt, pp, pq, msn=25, msd=msd, cdf=1.0)
        np.testing.assert_allclose(a["d"], want["d"], rtol=1e-12, atol=1e-15)
        assert (a["keep"] == want["keep"]).mean() > 0.9999   # a pair exactly on the tolerance may flip with the sum order


def test_wave_rejection_kernel_equals_lane_kernel(pcr, ctx, oracle):
    """On single-level grids CRejectionCentroidDist's searches run one WAVE per search (reject_stat_wave_kernel); pcr_tune
    key 7 = 1 forces the lane-per-search kernel on the same indices: statistic, mean, sigma and mask bit for bit, on pairs
    at a coarse alignment (the source-side query lies cells off the surface) and at the true one."""
    from pointcloudregistration_amd import synth
    src, tgt, Rt, tt = synth.registration_pair(80000, 70000, 17)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src)), pcr.Index(ctx, pcr.Cloud(ctx, tgt))
    U = tix.unique_sq_dist()
    for R, T in (synth.coarse_initial_transform(Rt, tt, 80000), (synth.colmajor(Rt), tt)):
        st = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, exact=True)
        kp = st["keep"].astype(bool)
        pp, pq = oracle.transform(src, R, T)[kp], tgt[st["fwd"][kp]]
        assert len(pp) > 10000
        for msn in (25, 9, 32):
            got = []
            for mode in (0, 1):
                try:
                    pcr.lib().pcr_tune(7, mode)
                    got.append(pcr.reject_centroid_dist(ctx, six, tix, pp, pq, msn=msn, msd=1.0, cdf=1.0))
                finally:
                    pcr.lib().pcr_tune(7, 0)
            a, b = got
            assert (a["d"] == b["d"]).all() and (a["keep"] == b["keep"]).all(), msn
            assert a["mean"] == b["mean"] and a["sigma"] == b["sigma"] and 0 < a["kept"] < len(pp), msn
    six.close()
    tix.close()


def test_sharded_centroid_rejection_equals_whole(pcr, oracle):
    """pcr_reject_centroid_dist_sharded (the second exchange): the pairs split over two "ranks" -- two
    threads, a context each, sharing the indices; the reduction hook sums the 17-double exchange buffers of both through
    the host -- must give every pair the statistic and the verdict the single call over all pairs gives; an empty
    shard and a one-rank communicator change nothing at all."""
    import ctypes as C
    import threading
    from pointcloudregistration_amd import synth
    src, tgt, Rt, tt = synth.registration_pair(30000, 30000, 99)
    R, T = synth.coarse_initial_transform(Rt, tt, 30000)
    ctxs = [pcr.Context(0), pcr.Context(0)]
    six, tix = pcr.Index(ctxs[0], pcr.Cloud(ctxs[0], src)), pcr.Index(ctxs[0], pcr.Cloud(ctxs[0], tgt))
    U = tix.unique_sq_dist()
    st = pcr.icp_step(ctxs[0], six, tix, R, T, 4, U * 36.0, U * 4.0, exact=True)
    kp = st["keep"].astype(bool)
    pp, pq = oracle.transform(src, R, T)[kp], tgt[st["fwd"][kp]]
    assert len(pp) > 5000
    whole = pcr.reject_centroid_dist(ctxs[0], six, tix, pp, pq, msn=25, msd=8.0, cdf=1.0)
    want = oracle.reject_centroid_dist(src, tgt, pp, pq, msn=25, msd=8.0, cdf=1.0)
    assert (whole["keep"] == want["keep"]).all() and 0 < whole["kept"] < len(pp)

    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def run(shards):
        gate = threading.Barrier(len(shards))
        slots = [np.zeros(17) for _ in shards]
        calls = [0] * len(shards)
        out = [None] * len(shards)

        def rank(r):
            def reduce(ptr):
                ctxs[r].synchronize()
                assert hip.hipMemcpy(slots[r].ctypes.data, ptr, 17 * 8, 2) == 0   # device -> host
                gate.wait()
                total = sum(slots)               # the same order on every rank
                gate.wait()                      # (nobody overwrites a slot that another rank still reads)
                assert hip.hipMemcpy(ptr, total.ctypes.data, 17 * 8, 1) == 0      # host -> device
                calls[r] += 1
            lo, hi = shards[r]
            out[r] = pcr.reject_centroid_dist(ctxs[r], six, tix, pp[lo:hi], pq[lo:hi], msn=25, msd=8.0, cdf=1.0, reduce=reduce)
        th = [threading.Thread(target=rank, args=(r,)) for r in range(len(shards))]
        [t.start() for t in th]
        [t.join(120) for t in th]
        assert calls == [2] * len(shards)
        return out

    h = len(pp) // 2 + 17
    a, b = run([(0, h), (h, len(pp))])
    assert (np.concatenate([a["d"], b["d"]]) == whole["d"]).all()
    for o in (a, b):   # every rank holds the statistics of ALL pairs (sums associate differently: last bits)
        assert abs(o["mean"] - whole["mean"]) <= 1e-12 * whole["mean"] and abs(o["sigma"] - whole["sigma"]) <= 1e-12 * whole["sigma"]
    assert (np.concatenate([a["keep"], b["keep"]]) == whole["keep"]).all()
    a, b = run([(0, len(pp)), (0, 0)])   # a rank without pairs still takes part in both reductions
    assert (a["keep"] == whole["keep"]).all() and a["mean"] == whole["mean"] and a["sigma"] == whole["sigma"]
    assert b["kept"] == 0 and b["mean"] == whole["mean"]
    # a hook that FAILS on its first call is still entered a second time (the other ranks are already on their way into
    # the second exchange and would wait there for ever), and the call reports the failure
    seen = []

    def failing(ptr):
        seen.append(ptr)
        return -2 if len(seen) == 1 else 0
    with pytest.raises(pcr.PcrError):
        pcr.reject_centroid_dist(ctxs[0], six, tix, pp[:100], pq[:100], msn=25, msd=8.0, cdf=1.0, reduce=failing)
    assert len(seen) == 2
    # no pair on ANY rank: an error, not a division by zero
    with pytest.raises(pcr.PcrError):
        pcr.reject_centroid_dist(ctxs[0], six, tix, pp[:0], pq[:0], msn=25, msd=8.0, cdf=1.0, reduce=lambda ptr: 0)
    try:
        comm = pcr.Comm(ctxs[0], 0, 1, pcr.Comm.unique_id())
    except pcr.PcrError:
        comm = None   # RCCL not loadable on this box: the hook path above has covered the logic
    if comm is not None:
        one = pcr.reject_centroid_dist(ctxs[0], six, tix, pp, pq, msn=25, msd=8.0, cdf=1.0, reduce=comm)
        assert (one["keep"] == whole["keep"]).all() and one["mean"] == whole["mean"] and one["sigma"] == whole["sigma"]
        comm.close()
    for c in ctxs:
        c.close()
