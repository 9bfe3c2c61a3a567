"""f4, opt-in: the normal-aware pair decision (pcr_icp_step_normal_fit) against the oracle's restatement of
CCorrespondenceEstimationClosestPoints::__insertFitPairs (CorrespondenceEstimationClosestPoints.cpp:128-175).
"Parity unpinned" by construction: the reference calls this function nowhere (:105 is commented out) and the function
reads a vector it never assigns (TmpTgtPoint) -- both sides here use the neighbour's position for it, as documented in
include/pcr.h and oracle/icrp.cpp.  What IS checked: the HIP path and the CPU restatement agree index for index."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sphere_pair(n, seed, off_surface):
    """Two samplings of the unit sphere with analytic normals (slightly perturbed, NOT unit length: the code
    normalises); the source is pushed off the surface along the normal by `off_surface` * N(0,1) and moved rigidly."""
    from pointcloudregistration_amd import synth
    rng = np.random.default_rng(seed)

    def cloud():
        u = rng.standard_normal((n, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        nrm = (u + rng.standard_normal((n, 3)) * 0.02) * rng.uniform(0.5, 2.0, (n, 1))
        return u, nrm
    tgt, tn = cloud()
    s, sn = cloud()
    s = s * (1.0 + rng.standard_normal((n, 1)) * off_surface)
    Rm, t = synth.euler_zyx(0.4, -0.3, 0.5), np.array([0.003, -0.002, 0.001])
    src = (s - t) @ Rm            # tgt ~ Rm src + t
    sn = sn @ Rm                  # normals rotate with the cloud
    return np.ascontiguousarray(src), np.ascontiguousarray(sn), tgt, np.ascontiguousarray(tn), synth.colmajor(Rm), t


@pytest.mark.parametrize("k,nvdt,compare", [(4, 0.0125, False), (4, 0.2, True), (1, 0.05, False), (8, 0.0125, True)])
def test_normal_fit_equals_the_oracle_restatement(pcr, ctx, oracle, k, nvdt, compare):
    n = 30000
    src, sn, tgt, tn, R, T = _sphere_pair(n, 31 + k, off_surface=0.03)
    sc, tc = pcr.Cloud(ctx, src), pcr.Cloud(ctx, tgt)
    six, tix = pcr.Index(ctx, sc, kind=pcr.INDEX_GRID), pcr.Index(ctx, tc, kind=pcr.INDEX_GRID)
    with pytest.raises(pcr.PcrError):   # no normals attached yet
        pcr.icp_step_normal_fit(ctx, six, tix, R, T, k, nvdt=nvdt)
    tc.set_normals(tn)
    if compare:
        with pytest.raises(pcr.PcrError):   # comparing normals needs the source's too
            pcr.icp_step_normal_fit(ctx, six, tix, R, T, k, nvdt=nvdt, compare_normals=True)
    sc.set_normals(sn)
    tree = oracle.KdOld(tgt, k)
    for smp in (None, np.random.default_rng(5).choice(n, 7001, replace=False).astype(np.uint32)):
        got = pcr.icp_step_normal_fit(ctx, six, tix, R, T, k, nvdt=nvdt, ndt=0.85, compare_normals=compare, sample_idx=smp)
        want = oracle.icrp_normal_fit(src, sn, tgt, tn, tree, R, T, k, nvdt=nvdt, ndt=0.85, compare_normals=compare,
                                      sample_idx=smp, threads=8)
        assert pcr.last_step_kernel() == "icp_step_normal_kernel"
        assert (got["tgt"] == want["tgt"]).all(), (got["tgt"] != want["tgt"]).sum()
        assert (got["keep"] == want["keep"]).all()
        assert got["partials"][0] == want["P"]
        m = n if smp is None else len(smp)
        assert 0.02 * m < want["P"] < 0.98 * m, "the case must accept some candidates and refuse others"
        if k > 1 and smp is None:   # later candidates do get chosen, not only the nearest
            first = oracle.KdOld(tgt, 1).query(oracle.transform(src, R, T))[:, 0]
            assert ((want["tgt"] != first) & (want["keep"] == 1)).sum() > 5
        pp, pq = pcr.icp_pivots(six, tix, R, T)
        R2, T2, _, _, _ = pcr.solve_rigid(got["partials"], pp, pq, R, T)
        Ro, To, _, _ = oracle.svd2(want["pair_p"], want["pair_q"], R, T)
        np.testing.assert_allclose(R2, Ro, rtol=0, atol=1e-9)
        np.testing.assert_allclose(T2, To, rtol=0, atol=1e-9)


def test_normal_fit_on_nested_levels(pcr, ctx, oracle):
    """The same on an index with nested levels (LiDAR-like density), normals random: only the index logic matters."""
    from pointcloudregistration_amd import synth
    n = 300000
    src, tgt, Rm, t = synth.lidar_pair(n, 99, outlier_frac=0.1)
    rng = np.random.default_rng(1)
    tn = rng.standard_normal((n, 3))
    sc, tc = pcr.Cloud(ctx, src), pcr.Cloud(ctx, tgt)
    six, tix = pcr.Index(ctx, sc), pcr.Index(ctx, tc)
    assert tix.info()["levels"] > 1
    tc.set_normals(tn)
    R, T = synth.colmajor(Rm), t
    got = pcr.icp_step_normal_fit(ctx, six, tix, R, T, 4, nvdt=0.3)
    assert pcr.last_step_kernel() == "icp_step_normal_kernel<multi-level>"
    want = oracle.icrp_normal_fit(src, None, tgt, tn, oracle.KdOld(tgt, 4), R, T, 4, nvdt=0.3, threads=8)
    assert (got["tgt"] == want["tgt"]).all() and got["partials"][0] == want["P"] > 1000


def test_plugin_opt_in_runs_the_normal_aware_pairs_stage_by_stage(oracle):
    """CorrespondenceEstimationClosestPointsGPU.NORMALFIT = 1 in the config: CIterativeFit runs stage by stage (not the
    fused device loop) with __insertFitPairs deciding the pairs; three iterations against the same loop made of the
    oracle's pieces (icrp_normal_fit -> svd2)."""
    from pointcloudregistration_amd import hostapi
    hostapi.lib()
    hostapi.reset()
    try:
        src, sn, tgt, tn, R, T = _sphere_pair(20000, 77, off_surface=0.03)
        hostapi.config_text("BaseConvergenceCriteria.MI = 3\nDefaultConvergenceCriteria.AMSE = 0\nDefaultConvergenceCriteria.RMSE = 0\n"
                            "DefaultConvergenceCriteria.RT = 0\nDefaultConvergenceCriteria.TT = 0\n"
                            "CorrespondenceEstimationClosestPoints.NN = 4\nCorrespondenceEstimationClosestPoints.NVDT = 0.05\n"
                            "CorrespondenceEstimationClosestPointsGPU.NORMALFIT = 1\n")
        out = hostapi.fit(src, tgt, R0=R, T0=T, src_normals=sn, tgt_normals=tn)
        assert out["status"] == 0 and out["iterations"] == 3, out
        Ro, To = np.array(R, np.float64).copy(), np.array(T, np.float64).copy()
        tree = oracle.KdOld(tgt, 4)
        for it in range(3):
            w = oracle.icrp_normal_fit(src, sn, tgt, tn, tree, Ro, To, 4, nvdt=0.05, threads=8)
            assert out["traj"][it, 13] == w["P"] > 500, it
            Ro, To, _, _ = oracle.svd2(w["pair_p"], w["pair_q"], Ro, To)
            np.testing.assert_allclose(out["traj"][it, :9], Ro, rtol=0, atol=1e-9)
            np.testing.assert_allclose(out["traj"][it, 9:12], To, rtol=0, atol=1e-9)
    finally:
        hostapi.reset()
