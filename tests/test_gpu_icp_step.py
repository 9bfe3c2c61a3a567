"""GPU parity: one ICRP iteration through the C ABI vs the golden vectors (indices produced by the ORACLE's restatement
of the reference's KD-tree, tests/golden/make_golden.py: nothing of the reference builds here, "parity unpinned") and vs
the oracle run live."""
import numpy as np
import pytest

from conftest import case_inputs, golden

pytestmark = pytest.mark.gpu

CASES = ["patch_k1", "patch_k4", "cube4096_k1", "cube4096_k4", "spheres40k_k1", "spheres40k_k4"]
# relative tolerance on fp64 sums accumulated in a different order than the reference:
SUM_RTOL = 1e-11


def _indices(pcr, ctx, name, kind):
    g = golden(name + ".npz")
    src, tgt = case_inputs(name, g)
    k = pcr.INDEX_GRID if kind == "grid" else pcr.INDEX_BRUTE
    six = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=k)
    tix = pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=k)
    return g, src, tgt, six, tix


@pytest.mark.parametrize("kind", ["grid", "brute"])
@pytest.mark.parametrize("name", CASES)
def test_step_matches_golden(pcr, ctx, name, kind):
    if kind == "brute" and name.startswith("spheres40k"):
        pytest.skip("brute-force path is for small clouds")
    g, src, tgt, six, tix = _indices(pcr, ctx, name, kind)
    k = int(g["k"])
    U = float(g["uniq_tgt"])
    assert tix.unique_sq_dist() == U  # CUniqueData: bit-identical (same neighbours, same sum order)
    assert six.unique_sq_dist() == float(g["uniq_src"])
    for it in range(len(g["it_R"])):
        R, T = g["it_R"][it], g["it_T"][it]
        out = pcr.icp_step(ctx, six, tix, R, T, k, U * g["sdft"], U * g["ndft"], exact=True)
        assert (out["fwd"] == g["it_fwd"][it]).all(), f"it {it}: fwd mismatches {(out['fwd'] != g['it_fwd'][it]).sum()}"
        assert (out["rev"] == g["it_rev"][it]).all(), f"it {it}: rev mismatches {(out['rev'] != g['it_rev'][it]).sum()}"
        assert (out["keep"] == g["it_keep"][it]).all()
        want = g["it_partials"][it]
        scale = np.abs(want).max()
        assert out["partials"][0] == want[0]
        np.testing.assert_allclose(out["partials"], want, rtol=SUM_RTOL, atol=SUM_RTOL * scale)
        # production mode (gate early-out): identical keep mask and sums, indices only where kept
        fast = pcr.icp_step(ctx, six, tix, R, T, k, U * g["sdft"], U * g["ndft"], exact=False)
        assert (fast["keep"] == g["it_keep"][it]).all()
        assert (fast["partials"] == out["partials"]).all()
        m = fast["keep"].astype(bool)
        assert (fast["fwd"][m] == g["it_fwd"][it][m]).all() and (fast["rev"][m] == g["it_rev"][it][m]).all()
        # host solve: next accumulated transform = the oracle trajectory
        pp, pq = pcr.icp_pivots(six, tix, R, T)
        R2, T2, Rc, Tc, mse = pcr.solve_rigid(out["partials"], pp, pq, R, T)
        np.testing.assert_allclose(R2, g["traj"][it, :9], rtol=0, atol=1e-9)
        np.testing.assert_allclose(T2, g["traj"][it, 9:12], rtol=0, atol=1e-9 * max(1.0, np.abs(tgt).max()))
        np.testing.assert_allclose(mse, g["traj"][it, 12], rtol=1e-10)


def test_sample_subset_and_order(pcr, ctx, oracle):
    """Explicit sample lists (SamplerRandom-style subsets), in arbitrary order, with repeats."""
    g, src, tgt, six, tix = _indices(pcr, ctx, "cube4096_k4", "grid")
    rng = np.random.default_rng(3)
    smp = rng.integers(0, len(src), 1500).astype(np.uint32)
    R, T, U = g["it_R"][1], g["it_T"][1], float(g["uniq_tgt"])
    want = oracle.icrp_correspondences(src, tgt, oracle.KdOld(tgt, 4), R, T, 4, U * 36, U * 4, sample_idx=smp)
    out = pcr.icp_step(ctx, six, tix, R, T, 4, U * 36, U * 4, sample_idx=smp)
    assert (out["fwd"] == want["fwd"]).all() and (out["rev"] == want["rev"]).all() and (out["keep"] == want["keep"]).all()
    assert out["partials"][0] == want["P"]


@pytest.mark.parametrize("k", [5, 6, 8])
def test_k5_to_8_step_and_fit_vs_oracle(pcr, ctx, oracle, k):
    """K = 5 ... 8 run the filtered kernel's 8-entry instance: one ICRP iteration against the oracle sample by sample
    (forward K-th neighbour, reverse winner, flag, pair count), and a whole device-resident fit with the reference's
    criteria against the oracle's restatement of CIterativeFit::fit (iterations, state, pairs per iteration exactly,
    trajectory to 1e-9)."""
    from pointcloudregistration_amd import synth
    src, tgt, Rm, t = synth.registration_pair(60000, 60000, 100 + k)
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID), pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
    U = tix.unique_sq_dist()
    assert U == oracle.uniq_sq_dist(tgt)
    R0, T0 = synth.coarse_initial_transform(Rm, t, 60000)
    want = oracle.icrp_correspondences(src, tgt, oracle.KdOld(tgt, k), R0, T0, k, U * 36.0, U * 4.0, threads=16)
    got = pcr.icp_step(ctx, six, tix, R0, T0, k, U * 36.0, U * 4.0, exact=True)
    assert pcr.last_step_kernel() == "icp_step_fast_kernel<8>"
    assert (got["fwd"] == want["fwd"]).all() and (got["rev"] == want["rev"]).all() and (got["keep"] == want["keep"]).all()
    assert got["partials"][0] == want["P"] > 1000
    conv = pcr.ConvergenceParams.regcfg(12)
    fit = pcr.Fit(ctx, six, tix, k, U * 36.0, U * 4.0, conv=conv, unique_sq_dist=U, traj_cap=16).run(R0, T0, 12)
    ref = oracle.fit(src, tgt, k=k, sdft=36.0, ndft=4.0, conv=oracle.ConvParams.regcfg(12), R0=R0, T0=T0, threads=16)
    assert fit["iterations"] == ref["iterations"] and fit["state"] == ref["state"]
    rt = np.asarray(ref["traj"])
    assert (fit["traj"][:, 13] == rt[:, 13]).all()
    np.testing.assert_allclose(fit["traj"][:, :12], rt[:, :12], rtol=0, atol=1e-9)
    np.testing.assert_allclose(fit["R"], ref["R"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(fit["T"], ref["T"], rtol=0, atol=1e-9)
    six.close()
    tix.close()


def test_no_pair_survives(pcr, ctx):
    """Far-apart clouds: the reference divides by zero (RegUtilityFunctions.cpp:71); the ABI reports it."""
    g, src, tgt, six, tix = _indices(pcr, ctx, "patch_k1", "grid")
    R, T = np.eye(3).reshape(9), np.zeros(3)
    U = float(g["uniq_tgt"])
    out = pcr.icp_step(ctx, six, tix, R, T, 1, U * 36, U * 4)
    assert out["partials"][0] == 0 and out["keep"].sum() == 0
    pp, pq = pcr.icp_pivots(six, tix, R, T)
    with pytest.raises(pcr.PcrError) as e:
        pcr.solve_rigid(out["partials"], pp, pq, R, T)
    assert e.value.status == -5


def test_filtered_kernel_equals_the_fp64_walk_bit_for_bit(pcr, ctx):
    """The fp32-filtered step kernel (fp32 keys select, fp64 decides and verifies; fast_search.h) against the kernel in
    which every candidate goes through the fp64 walk (pcr_tune key 5 = 1): every index, flag and partial sum."""
    from pointcloudregistration_amd import synth
    for n, seed in ((150000, 41), (30000, 42)):
        src, tgt, Rm, t = synth.registration_pair(n, n, seed)
        six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID), pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
        U = tix.unique_sq_dist()
        smp = np.random.default_rng(6).choice(n, n // 7, replace=False).astype(np.uint32)
        transforms = [synth.coarse_initial_transform(Rm, t, n), (np.eye(3).reshape(9), np.zeros(3)),
                      (synth.colmajor(Rm), t)]
        for R, T in transforms:
            out = {}
            try:
                for mode in (1, 0):
                    pcr.lib().pcr_tune(5, mode)
                    pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, want_outputs=False)
                    assert pcr.last_step_kernel() == ("icp_step_grid_kernel" if mode else "icp_step_fast_kernel<4>")
                    out[mode] = [pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, exact=True),
                                 pcr.icp_step(ctx, six, tix, R, T, 4, U * 36.0, U * 4.0, exact=False),
                                 pcr.icp_step(ctx, six, tix, R, T, 1, U * 36.0, U * 4.0, exact=True),
                                 pcr.icp_step(ctx, six, tix, R, T, 2, U * 36.0, U * 4.0, sample_idx=smp, exact=True),
                                 pcr.icp_step(ctx, six, tix, R, T, 3, U * 9.0, U * 1.0, sample_idx=smp, exact=False),
                                 pcr.icp_step(ctx, six, tix, R, T, 5, U * 36.0, U * 4.0, exact=True),
                                 pcr.icp_step(ctx, six, tix, R, T, 6, U * 36.0, U * 4.0, sample_idx=smp, exact=False),
                                 pcr.icp_step(ctx, six, tix, R, T, 7, U * 9.0, U * 1.0, sample_idx=smp, exact=True),
                                 pcr.icp_step(ctx, six, tix, R, T, 8, U * 36.0, U * 4.0, exact=False)]
                    assert pcr.last_step_kernel() == ("icp_step_grid_kernel" if mode else "icp_step_fast_kernel<8>")
            finally:
                pcr.lib().pcr_tune(5, 0)
            for a, b in zip(out[1], out[0]):
                for key in ("fwd", "rev", "keep", "partials"):
                    assert (a[key] == b[key]).all(), (n, key, int((a[key] != b[key]).sum()))
        six.close()
        tix.close()


def test_filtered_kernel_on_partial_scans_equals_the_fp64_walk(pcr, ctx):
    """Two partial scans of uneven density (a tenth of the samples lie in stretches where the 3x3x3 cells do not settle
    the search): the filtered kernel hands those to the fp64 walk SEEDED with its own verified list (shells from ring 2,
    no second visit to ring 1).  Every index, flag and sum must equal the all-fp64 kernel's, for every k <= KC."""
    from pointcloudregistration_amd import synth
    scans, _ = synth.scan_sequence(2, 60000, 9000)
    src, tgt = scans[1], scans[0]
    six, tix = pcr.Index(ctx, pcr.Cloud(ctx, src), kind=pcr.INDEX_GRID), pcr.Index(ctx, pcr.Cloud(ctx, tgt), kind=pcr.INDEX_GRID)
    U = tix.unique_sq_dist()
    R, T = np.eye(3).reshape(9), np.zeros(3)
    out = {}
    try:
        for mode in (1, 0):
            pcr.lib().pcr_tune(5, mode)
            out[mode] = [pcr.icp_step(ctx, six, tix, R, T, k, U * 36.0, U * 4.0, exact=ex) for k in (1, 2, 3, 4) for ex in (True, False)]
            assert pcr.last_step_kernel() == ("icp_step_grid_kernel" if mode else "icp_step_fast_kernel<4>")
            out[mode] += [pcr.icp_step(ctx, six, tix, R, T, k, U * 36.0, U * 4.0, exact=ex) for k in (5, 8) for ex in (True, False)]
            assert pcr.last_step_kernel() == ("icp_step_grid_kernel" if mode else "icp_step_fast_kernel<8>")
    finally:
        pcr.lib().pcr_tune(5, 0)
    unpaired = int((out[0][7]["fwd"] == 0xFFFFFFFF).sum())
    assert 2000 < unpaired < 30000, unpaired  # the data does have samples outside the overlap
    for a, b in zip(out[1], out[0]):
        for key in ("fwd", "rev", "keep", "partials"):
            assert (a[key] == b[key]).all(), (key, int((a[key] != b[key]).sum()))
    six.close()
    tix.close()
