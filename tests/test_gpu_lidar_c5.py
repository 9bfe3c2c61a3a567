"""GPU parity on LiDAR-like scans (config C5 of SURVEY.md 8d): range-dependent density (a uniform grid's worst
case: cells near the sensor hold hundreds of points, far cells are empty), 30 % uniform-in-volume outliers in the
source.  2 M x 2 M is compared with the oracle sample by sample; 16 M x 16 M (the full C5 size, one GPU's share of
nothing -- clouds are replicated) through size-independent properties and an oracle spot check."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _u(oracle, tgt):
    return oracle.uniq_sq_dist(tgt)


@pytest.mark.timeout(900)
def test_c5_2m_iteration_vs_oracle(pcr, ctx, oracle):
    from pointcloudregistration_amd import synth
    n = 2_000_000
    src, tgt, Rm, t = synth.lidar_pair(n, 515)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src)), pcr.Index(ctx, pcr.Cloud(ctx, tgt))
    assert six.info()["levels"] > 1 and tix.info()["levels"] > 1   # the dense core gets grids of its own
    U = tix.unique_sq_dist()
    assert U == _u(oracle, tgt)
    R, T = synth.colmajor(synth.euler_zyx(0.3, -0.2, 1.2)), np.array([0.3, -0.15, 0.04])  # a coarse guess
    tree = oracle.KdOld(tgt, 4)
    for it in range(2):
        got = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, exact=True)
        want = oracle.icrp_correspondences(src, tgt, tree, R, T, 4, U * 36.0, U * 4.0, threads=16)
        assert (got["fwd"] == want["fwd"]).all(), (got["fwd"] != want["fwd"]).sum()
        assert (got["rev"] == want["rev"]).all(), (got["rev"] != want["rev"]).sum()
        assert (got["keep"] == want["keep"]).all()
        assert got["partials"][0] == want["P"] > 1000
        pp, pq = pcr.icp_pivots(six, tix, R, T)
        R2, T2, _, _, _ = pcr.solve_rigid(got["partials"], pp, pq, R, T)
        Ro, To, _, _ = oracle.svd2(want["pair_p"], want["pair_q"], R, T)
        np.testing.assert_allclose(R2, Ro, rtol=0, atol=1e-9)
        np.testing.assert_allclose(T2, To, rtol=0, atol=1e-7)
        R, T = R2, T2


@pytest.mark.timeout(1200)
def test_c5_16m_properties(pcr, ctx, oracle):
    from pointcloudregistration_amd import synth
    n = 16_000_000
    src, tgt, Rm, t = synth.lidar_pair(n, 516)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src)), pcr.Index(ctx, pcr.Cloud(ctx, tgt))
    U = tix.unique_sq_dist()
    assert U == _u(oracle, tgt)
    R, T = synth.colmajor(Rm), t
    # one production-mode step over all 16 M samples, twice: deterministic, and linear over a split of the samples
    a = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, exact=False, want_outputs=False)["partials"]
    b = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, exact=False, want_outputs=False)["partials"]
    assert (a == b).all() and a[0] > 100000
    # oracle spot check: 50 000 random samples, indices bit-exact
    smp = np.random.default_rng(1).choice(n, 50000, replace=False).astype(np.uint32)
    got = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, sample_idx=smp, exact=True)
    want = oracle.icrp_correspondences(src, tgt, oracle.KdOld(tgt, 4), R, T, 4, U * 36.0, U * 4.0, sample_idx=smp, threads=16)
    assert (got["fwd"] == want["fwd"]).all() and (got["rev"] == want["rev"]).all() and (got["keep"] == want["keep"]).all()
    # K-NN on the dense core and on far outliers: ascending, distances those of the reported points
    q = np.concatenate([src[:200000], (np.random.default_rng(2).random((1000, 3)) - 0.5) * 1000.0])
    idx, d2 = tix.knn(q, 4, want_d2=True)
    assert (np.diff(d2, axis=1) >= 0).all()
    p = tgt[idx]
    dd = ((q[:, None, 0] - p[:, :, 0]) ** 2 + (q[:, None, 1] - p[:, :, 1]) ** 2) + (q[:, None, 2] - p[:, :, 2]) ** 2
    assert (dd == d2).all()
    pick = np.random.default_rng(3).choice(len(q), 20000, replace=False)
    wi, wd = oracle.KdOld(tgt, 4).query(q[pick], want_d2=True, threads=16)
    assert (wi == idx[pick]).all() and (wd == d2[pick]).all()
