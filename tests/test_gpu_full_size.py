"""GPU parity at BASELINE.json's full size (C3: 1 M x 1 M points): size-independent properties of the K-NN
result, spot checks against the oracle's KD-tree, and ONE full ICRP iteration compared sample by sample with
the oracle (the oracle needs a few seconds for it on the box's host cores)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1_000_000


@pytest.fixture(scope="module")
def c3(pcr, ctx):
    from pointcloudregistration_amd import synth
    src, tgt, Rm, t = synth.registration_pair(N, N, 20250001)
    six = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID)
    tix = pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
    R0, T0 = synth.coarse_initial_transform(Rm, t, N)
    return dict(src=src, tgt=tgt, six=six, tix=tix, R0=R0, T0=T0, Rm=Rm, t=t)


def test_knn_properties_at_1m(pcr, c3, oracle):
    tgt, tix = c3["tgt"], c3["tix"]
    q = np.ascontiguousarray(c3["src"] @ c3["Rm"].T + c3["t"])
    idx4, d4 = tix.knn(q, 4, want_d2=True)
    idx1, d1 = tix.knn(q, 1, want_d2=True)
    assert idx4.max() < N
    assert (np.diff(d4, axis=1) >= 0).all()                       # ascending
    assert (idx1[:, 0] == idx4[:, 0]).all() and (d1[:, 0] == d4[:, 0]).all()  # K=1 is the head of K=4
    s = np.sort(idx4, axis=1)
    assert (np.diff(s, axis=1) > 0).all()                         # four distinct neighbours
    # reported distances are the distances of the reported points, bit for bit, in the reference's operation order
    p = tgt[idx4]
    dd = ((q[:, None, 0] - p[:, :, 0]) ** 2 + (q[:, None, 1] - p[:, :, 1]) ** 2) + (q[:, None, 2] - p[:, :, 2]) ** 2
    assert (dd == d4).all()
    # self-query: every target point is its own nearest neighbour at distance 0
    own = tix.knn(tgt[::3], 1)
    assert (own[:, 0] == np.arange(N, dtype=np.uint32)[::3]).all()
    # descending order is the mirror image
    assert (tix.knn(q[:5000], 4, order=pcr.ORDER_DESCENDING) == idx4[:5000, ::-1]).all()
    # spot check against the oracle's restatement of the reference KD-tree on 20 000 random queries
    pick = np.random.default_rng(0).choice(N, 20000, replace=False)
    want, wd = oracle.KdOld(tgt, 4).query(q[pick], want_d2=True, threads=16)
    assert (want == idx4[pick]).all() and (wd == d4[pick]).all()


@pytest.mark.timeout(600)
def test_one_full_iteration_at_1m_vs_oracle(pcr, ctx, c3, oracle):
    src, tgt, six, tix = c3["src"], c3["tgt"], c3["six"], c3["tix"]
    U = tix.unique_sq_dist()
    assert U == oracle.uniq_sq_dist(tgt)
    R, T = c3["R0"], c3["T0"]
    got = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, exact=True)
    want = oracle.icrp_correspondences(src, tgt, oracle.KdOld(tgt, 4), R, T, 4, U * 36.0, U * 4.0, threads=16)
    assert (got["fwd"] == want["fwd"]).all()
    assert (got["rev"] == want["rev"]).all()
    assert (got["keep"] == want["keep"]).all()
    assert got["partials"][0] == want["P"] and want["P"] > 100000
    pp, pq = pcr.icp_pivots(six, tix, R, T)
    R2, T2, _, _, mse = pcr.solve_rigid(got["partials"], pp, pq, R, T)
    Ro, To, _, _ = oracle.svd2(want["pair_p"], want["pair_q"], R, T)
    np.testing.assert_allclose(R2, Ro, rtol=0, atol=1e-10)
    np.testing.assert_allclose(T2, To, rtol=0, atol=1e-10)
    # production mode (gate early-out, no index outputs) gives the same sums
    fast = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, exact=False)
    assert (fast["partials"] == got["partials"]).all() and (fast["keep"] == got["keep"]).all()
    # linearity of the partial sums over a split of the samples (what the multi-GPU all-reduce relies on)
    half = N // 2
    a = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, sample_idx=np.arange(half, dtype=np.uint32))
    b = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, sample_idx=np.arange(half, N, dtype=np.uint32))
    tot = a["partials"] + b["partials"]
    assert tot[0] == got["partials"][0]
    np.testing.assert_allclose(tot, got["partials"], rtol=1e-10, atol=1e-10 * np.abs(got["partials"]).max())
