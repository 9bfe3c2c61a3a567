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
